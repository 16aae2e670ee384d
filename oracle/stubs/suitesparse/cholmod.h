// TEST INFRASTRUCTURE ONLY. Stand-in for <suitesparse/cholmod.h> (SuiteSparse is not
// installed here), so that the reference's uncertainty.c compiles in place into
// oracle/_ref/. uncertainty.c:9 includes it for ONE thing: the cholmod_sparse type whose
// p, i, x and nrow it reads (uncertainty.c:862-864, 966); it calls no CHOLMOD function.
// The type lives in the dogleg.h stand-in (same field order as SuiteSparse's)
#pragma once
#include "../dogleg.h"
