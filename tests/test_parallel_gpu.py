"""The multi-GPU driver (mrcal_amd/parallel.py) on real hardware, as far as one
GPU allows:
  - world 1: GpuShard + the Python dog-leg loop == the C++ single-GPU solver
  - world 2 on ONE device over gloo (RCCL refuses two ranks per device): the
    frame-sharded phase kernels + the collectives == the single-GPU solve
"""
import os
import sys
import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _problem(amd_api):
    from mrcal_amd.synthetic import make_calibration_problem
    return make_calibration_problem(amd_api, Ncameras=3, Nframes=11, lensmodel="LENSMODEL_OPENCV8",
                                    object_width_n=10, object_height_n=10, seed=5)[0]


def test_world1_python_driver_matches_cpp_solver(amd):
    from mrcal_amd.resident import Problem
    from mrcal_amd.parallel import ShardedProblem
    from mrcal_amd.synthetic import copy_inputs
    oi = _problem(amd._api)
    with Problem(**copy_inputs(oi)) as p:
        s_cpp = p.solve()
        b_cpp = p.b_packed()
    sp = ShardedProblem(**copy_inputs(oi))
    s_py = sp.solve()
    b_py = sp.b_packed()
    sp.close()
    assert s_py["Noutliers_board"] == s_cpp["Noutliers_board"]
    assert abs(s_py["rms_reproj_error__pixels"] - s_cpp["rms_reproj_error__pixels"]) < 1e-9
    assert np.abs(b_py - b_cpp).max() < 2e-5


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    oi = _problem(mrcal_amd._api)
    sp = ShardedProblem(**oi)
    st = sp.solve()
    b  = sp.b_packed()
    if rank == 0:
        np.savez(out_path, b=b, rms=st["rms_reproj_error__pixels"], Noutliers=st["Noutliers_board"],
                 Ncollectives=st.get("Ncollectives", sp.comm.Ncollectives), frames=np.array(sp.frame_range))
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharded_on_one_device_matches_single(amd, tmp_path):
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    oi = _problem(amd._api)
    with Problem(**oi) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "w2.npz")
    port = 29600 + (os.getpid() % 300)
    try:
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    except Exception as e:
        if "gloo" in str(e).lower() and "cuda" in str(e).lower():
            pytest.skip(f"gloo cannot move device tensors in this build: {e}")
        raise
    r = np.load(out)
    assert int(r["Noutliers"]) == s1["Noutliers_board"]
    assert abs(float(r["rms"]) - s1["rms_reproj_error__pixels"]) < 1e-8
    assert np.abs(r["b"] - b1).max() < 2e-5
    assert int(r["Ncollectives"]) > 0
