# on the GPU box: configuration 2 with two builds of the library, alternating (A = mrcal_amd/lib_head.so, B = the tree's)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2 3; do
  echo "A $(MRCAL_AMD_LIB=$R/mrcal_amd/lib_head.so python tools/probe_config2.py 2>&1 | grep 'config2 ms' | cut -c1-40)"
  echo "B $(python tools/probe_config2.py 2>&1 | grep 'config2 ms' | cut -c1-40)"
done
