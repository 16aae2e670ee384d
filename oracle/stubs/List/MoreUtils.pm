# TEST INFRASTRUCTURE ONLY (oracle build).
#
# Minimal stand-in for the CPAN module List::MoreUtils, which is not installed
# in this image. The reference's minimath/minimath_generate.pl uses exactly one
# symbol from it: pairwise (minimath_generate.pl:6,114,148,170). This shim
# provides only that, so that the generator can be run unmodified from
# /root/reference to produce oracle/_ref/minimath_generated.h
package List::MoreUtils;
use strict;
use warnings;
use Exporter 'import';
our @EXPORT_OK = qw(pairwise);

sub pairwise(&\@\@)
{
    my ($code, $l0, $l1) = @_;
    my $pkg = caller;
    my $n = @$l0 > @$l1 ? scalar(@$l0) : scalar(@$l1);
    my @out;
    no strict 'refs';
    for my $i (0..$n-1)
    {
        local (${"${pkg}::a"}, ${"${pkg}::b"}) = ($l0->[$i], $l1->[$i]);
        push @out, $code->();
    }
    return @out;
}
1;
