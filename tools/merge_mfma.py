#!/usr/bin/env python3
"""gpurun_out/mfma_ns.json + mfma_config2.json (tools/collect_mfma.sh) -> profiles/r02_mfma_utilisation.json"""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
old = json.load(open(os.path.join(R, "profiles", "r02_mfma_utilisation.json")))
old["ns"]      = json.load(open(os.path.join(R, "gpurun_out", "mfma_ns.json")))
old["config2"] = json.load(open(os.path.join(R, "gpurun_out", "mfma_config2.json")))
old["how"] = old["how"].split(" v_mfma_f64_4x4x4 holds")[0] + " (config2: the launch-per-panel Cholesky, lchol_panel_kernel, and the sparse SYRK)"
json.dump(old, open(os.path.join(R, "profiles", "r02_mfma_utilisation.json"), "w"), indent=1)
print({k: {kk: vv.get("mfma_busy_frac") for kk, vv in old[k].items()} for k in ("ns", "config2")})
