python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
bash tools/collect_r06_pmc.sh r06a "1 2 5"
