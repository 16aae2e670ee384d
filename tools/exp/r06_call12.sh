O=gpurun_out
for h in "" "lchol_sweep=1" "lchol_fallback_log10=-30" "lchol_fallback_log10=-2"; do
  python tools/exp/r06_dbg_fallback.py $h >> $O/r06k_dbg_fallback.txt 2>&1
done
python tools/probe_oneshot.py > $O/r06k_oneshot.txt 2>&1
