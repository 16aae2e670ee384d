"""The randomized parity sweep of tools/fuzz_parity.py where the driver runs it: random small calibration
problems - 9 lens models, 1-4 cameras, 2-12 frames, boards of 3..10 x 3..10 corners, random do_optimize_*
selections, discrete points, input outliers -, structure-from-motion shapes (triangulated points + board frames)
and moving-camera problems (the extrinsics eliminated) through mrcal_amd.optimizer_callback() and through the
reference's own mrcal_optimizer_callback() (oracle/_ref): b_packed and the CSR structure bit-exact, x and J to
1e-6. Callbacks only: the solves of the sweep are the tool's (profiles/r03_fuzz_parity.txt) and, where they matter,
tests of their own (test_solver_parity.py, test_full_size.py)."""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", (4001, 501, 777))      # (the seeds of the two 108-case builder sweeps of round 5 with their solves: profiles/r05_fuzz_sweep_seed*.txt)
def test_fuzz_callbacks(amd, ref_api, seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    lines = []
    bad = fuzz_parity.run(40, seed, callbacks_only=True, out=lambda *a, **k: lines.append(" ".join(str(x) for x in a)))
    assert bad == 0, "\n".join(l for l in lines if not l.endswith("ok (callback only)"))
    # 40 board/point cases + 8 structure-from-motion + 6 moving-camera, every one compared
    assert sum(l.endswith("ok (callback only)") for l in lines) == 40 + 8 + 6
