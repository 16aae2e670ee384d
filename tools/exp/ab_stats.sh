# on the GPU box: configuration 2's kernel table with two builds of the library (A = mrcal_amd/lib_head.so, B = the tree's)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in A B; do
  rm -rf /tmp/prof_$n
  lib=$R/mrcal_amd/libmrcal_amd.so; [ $n = A ] && lib=$R/mrcal_amd/lib_head.so
  MRCAL_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python $R/tools/probe_config2.py > $O/ab_$n.log 2>&1
  python $R/tools/kernel_stats_table.py /tmp/prof_$n "$n" > $O/ab_$n.txt
  echo "== $n"; grep "${1:-assemble_splined}" $O/ab_$n.txt | cut -c1-50,100-150
done
for n in A B; do echo "== trace $n"; python $R/tools/step_trace_dump.py /tmp/prof_$n 8 | cut -c1-100 | grep -v lchol; done
