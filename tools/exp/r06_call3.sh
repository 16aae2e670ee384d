# round 6, call 3: wall-clock timelines of the board kernel's waves; kernel tables of configurations 1, 5, 2; the new bench
O=gpurun_out
for c in 1 ns; do
    MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so python tools/probe_board_ts.py $c > $O/r06c_board_ts_$c.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 1 5 2; do
    rm -rf /tmp/prof_c$c
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -- python $R/bench.py --only-config $c > $R/$O/r06c_config$c.json 2> /dev/null
    python $R/tools/kernel_stats_table.py /tmp/prof_c$c "round 6 (call 3), configuration $c: rocprofv3 --kernel-trace --stats -- python bench.py --only-config $c" > $R/$O/r06c_kernel_stats_config$c.txt
done
cd $R
python bench.py > $O/r06c_bench.json 2> $O/r06c_bench.err
