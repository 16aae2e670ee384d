# on the GPU box: the kernel table of the real splined calibration (tools/probe_real_splined.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_real
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_real -- python $R/tools/exp/probe_real_steps.py > $O/real.log 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_real "the reference documentation's splined calibration (186 frames, every observation a close-up): rocprofv3 --kernel-trace --stats -- python tools/exp/probe_real_steps.py (3 x 22 trial steps)" > $O/real_kernel_stats.txt
head -20 $O/real_kernel_stats.txt | cut -c1-60,100-170
python $R/tools/step_trace_dump.py /tmp/prof_real 30 | cut -c1-100
