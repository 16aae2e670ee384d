"""The multi-GPU solve (mrcal_amd/parallel.py, csrc/comm.cpp) on real hardware, as
far as one GPU allows:
  - RCCL, world 1: the product path - communicator made from a unique id, both
    collectives of every trial step issued from C++ on the problem's stream -
    gives the single-GPU solve BIT for bit, with <= 2 collectives per trial step
  - world 2 on ONE device over gloo (RCCL refuses two ranks per device): the
    frame-sharded kernels + the two sums per trial step == the single-GPU solve,
    with bit-identical replicated state on both ranks (protocol reference driver)
  - run_steps() continues where the previous call stopped
  - the C++ sharded solve at worlds 2 and 3 on ONE device, its collectives staged
    through host shared memory (mrcal_amd_comm_create_host): boards, points,
    pairs, outlier passes, a rank without boards
"""
import os
import sys
import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


SPLINED = "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120"


def _problem(amd_api, lensmodel="LENSMODEL_OPENCV8"):
    from mrcal_amd.synthetic import make_calibration_problem
    extra = {"do_optimize_intrinsics_core": False} if "SPLINED" in lensmodel else {}
    return make_calibration_problem(amd_api, Ncameras=3, Nframes=11, lensmodel=lensmodel,
                                    object_width_n=10, object_height_n=10, seed=5, **extra)[0]


def test_world1_python_driver_matches_cpp_solver(amd):
    from mrcal_amd.resident import Problem
    from mrcal_amd.parallel import ShardedProblem
    from mrcal_amd.synthetic import copy_inputs
    oi = _problem(amd._api)
    with Problem(**copy_inputs(oi)) as p:
        s_cpp = p.solve()
        b_cpp = p.b_packed()
    sp = ShardedProblem(_driver="python", **copy_inputs(oi))
    s_py = sp.solve()
    b_py = sp.b_packed()
    Ncoll, Ntrials = sp.Ncollectives, sp.dogleg.Ntrials_total
    sp.close()
    assert s_py["Noutliers_board"] == s_cpp["Noutliers_board"]
    assert s_py["Niterations"] == s_cpp["Niterations"] and s_py["Nevaluations"] == s_cpp["Nevaluations"]
    # same kernels, same sums in the same order: the same bits
    assert s_py["norm2_x"] == s_cpp["norm2_x"]
    assert np.array_equal(b_py, b_cpp)
    # two sums per trial step (+ a handful per outlier pass and the final gather)
    assert Ncoll <= 2*Ntrials + 4*(s_py["Noutlier_passes"] + 1) + 1


def _worker(rank, world, port, out_path, lensmodel):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    oi = _problem(mrcal_amd._api, lensmodel)
    sp = ShardedProblem(_driver="python", **oi)
    st = sp.solve()
    b  = sp.b_packed()
    # the replicated control state and the (gathered) state vector must be BIT-identical on all ranks
    tb = torch.from_numpy(np.stack((b, -b)))
    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
    assert np.array_equal(tb[0].numpy(), b) and np.array_equal(-tb[1].numpy(), b), "ranks disagree on the solution"
    ti = torch.tensor([st["Niterations"], -st["Niterations"], st["Nevaluations"], -st["Nevaluations"]])
    dist.all_reduce(ti, op=dist.ReduceOp.MAX)
    assert ti[0] == -ti[1] and ti[2] == -ti[3], "ranks disagree on the iteration counts"
    if rank == 0:
        np.savez(out_path, b=b, rms=st["rms_reproj_error__pixels"], Noutliers=st["Noutliers_board"],
                 Ncollectives=sp.Ncollectives, Ntrials=sp.dogleg.Ntrials_total, Npasses=st["Noutlier_passes"],
                 frames=np.array(sp.frame_range))
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


# (the splined models: the staged assembly and its gather, a shard's frames only)
@pytest.mark.parametrize("lensmodel", ("LENSMODEL_OPENCV8", SPLINED))
def test_world2_sharded_on_one_device_matches_single(amd, tmp_path, lensmodel):
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    oi = _problem(amd._api, lensmodel)
    with Problem(**oi) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "w2.npz")
    port = 29600 + (os.getpid() % 300)
    try:
        mp.spawn(_worker, args=(2, port, out, lensmodel), nprocs=2, join=True)
    except Exception as e:
        if "gloo" in str(e).lower() and "cuda" in str(e).lower():
            pytest.skip(f"gloo cannot move device tensors in this build: {e}")
        raise
    r = np.load(out)
    assert 0 < r["frames"][1] < 11            # the frames really were split
    assert int(r["Noutliers"]) == s1["Noutliers_board"]
    assert abs(float(r["rms"]) - s1["rms_reproj_error__pixels"]) < 1e-8
    # (a splined surface has knots the boards barely reach: the optimum is flat along them and the two runs, whose
    #  sums are ordered differently, stop 1e-4 apart there in packed units; the cost and the outliers agree)
    assert np.abs(r["b"] - b1).max() < (1e-3 if "SPLINED" in lensmodel else 2e-5)
    assert 0 < int(r["Ncollectives"]) <= 2*int(r["Ntrials"]) + 4*(int(r["Npasses"]) + 1) + 1


def test_run_steps_continues(amd):
    from mrcal_amd.parallel import ShardedProblem
    from mrcal_amd.resident import Problem
    from mrcal_amd.synthetic import copy_inputs
    oi = _problem(amd._api)
    sp = ShardedProblem(_driver="python", **copy_inputs(oi))
    n, tr = sp.run_steps(3, None)
    assert n == 3 and tr > 0
    n, tr = sp.run_steps(4, tr)
    st = sp.solver_stats()
    sp.close()
    with Problem(**copy_inputs(oi)) as p:
        _, tr1 = p.run_steps(7, None)
        s1 = p.solver_stats()
    assert n == 4
    assert st["Nevaluations"] == s1["Nevaluations"] == 8          # the seed + 7 trial points
    assert st["norm2_x"] == s1["norm2_x"]
    assert tr == tr1


def _rccl_worker(out_path, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    oi = _problem(mrcal_amd._api)
    sp = ShardedProblem(**oi)                 # the product path: RCCL from C++
    st = sp.solve()
    b  = sp.b_packed()
    Ncoll_solve = sp.Ncollectives
    n, tr = sp.run_steps(5, None)
    np.savez(out_path, b=b, rms=st["rms_reproj_error__pixels"], Noutliers=st["Noutliers_board"],
             Niterations=st["Niterations"], Nevaluations=st["Nevaluations"], Npasses=st["Noutlier_passes"],
             norm2_x=st["norm2_x"], Ncollectives=Ncoll_solve, Ncollectives_5steps=sp.Ncollectives - Ncoll_solve)
    sp.close()
    dist.destroy_process_group()


def test_rccl_world1_is_the_single_gpu_solve(amd, tmp_path):
    """RCCL cannot put two ranks on one device, so the real backend is exercised
    with a world of ONE: ncclCommInitRank from a unique id, every all-reduce of
    the sharded step issued by libmrcal_amd.so on the problem's HIP stream - the
    plumbing the 8-GPU run depends on. The sums of one rank are the single-GPU
    numbers: bit-identical results, and at most two collectives per trial step"""
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    oi = _problem(amd._api)
    with Problem(**oi) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "rccl1.npz")
    port = 29900 + (os.getpid() % 90)
    ctx = mp.get_context("spawn")
    proc = ctx.Process(target=_rccl_worker, args=(out, port))
    proc.start()
    proc.join(300)
    if proc.is_alive():
        proc.terminate()
        pytest.fail("the RCCL world-1 run hung")
    assert proc.exitcode == 0
    r = np.load(out)
    assert int(r["Noutliers"]) == s1["Noutliers_board"]
    assert int(r["Niterations"]) == s1["Niterations"] and int(r["Nevaluations"]) == s1["Nevaluations"]
    assert float(r["norm2_x"]) == s1["norm2_x"]
    assert np.array_equal(r["b"], b1)
    # every evaluation (trial or starting point of a pass) costs two collectives; a pass a few more
    Ntrials_max = int(r["Nevaluations"]) + 8*(int(r["Npasses"]) + 1)
    assert 20 < int(r["Ncollectives"]) <= 2*Ntrials_max + 5*(int(r["Npasses"]) + 1) + 1
    assert int(r["Ncollectives_5steps"]) == 2*(5 + 1)


def _points_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    from test_callback_parity import points_only_problem
    oi = points_only_problem(mrcal_amd._api)
    sp = ShardedProblem(_driver="python", **oi)
    st = sp.solve()
    if rank == 0:
        np.savez(out_path, b=sp.b_packed(), rms=st["rms_reproj_error__pixels"], norm2_x=st["norm2_x"])
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


def test_world2_points_only(amd, tmp_path):
    """no frames at all: every rank gets the empty frame range, and only the
    leader may own the points and the regularization rows - or the sums over
    the ranks count them twice (|x|^2 would come out doubled)"""
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    from test_callback_parity import points_only_problem
    oi = points_only_problem(amd._api)
    with Problem(**oi) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "w2p.npz")
    port = 29300 + (os.getpid() % 250)
    mp.spawn(_points_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    assert abs(float(r["norm2_x"]) - s1["norm2_x"]) < 1e-9*s1["norm2_x"]
    assert abs(float(r["rms"]) - s1["rms_reproj_error__pixels"]) < 1e-9
    assert np.abs(r["b"] - b1).max() < 1e-6


def _rccl_world2_worker(rank, world, port, out_path):
    """one rank of a REAL multi-GPU run: its own device, RCCL all-reduces issued from libmrcal_amd.so"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    oi = _problem(mrcal_amd._api)
    sp = ShardedProblem(_driver="rccl", **oi)
    st = sp.solve()
    b = sp.b_packed()
    # every rank ends with the same replicated state
    t = torch.from_numpy(b.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    # the first box with two devices leaves MEASURED numbers behind where DESIGN.md section 7 has
    # assumptions (25 / 35 us for the first collective, 12 us for the second): the solve's wall clock on every rank, a
    # trial step, and the two all-reduces at the sizes of the metric's problem, configuration 3 and configuration 2,
    # from event pairs around the library's own ncclAllReduce on a stream (one a time: latency, not throughput)
    import time, json
    measured = dict(rank=rank, device=torch.cuda.current_device(), solve_seconds=st.get("seconds"),
                    evaluations=st["Nevaluations"], collectives=sp.Ncollectives)
    try:
        sp2 = ShardedProblem(_driver="rccl", **_problem(mrcal_amd._api))
        _, tr = sp2.run_steps(5, None); sp2.synchronize(); dist.barrier()
        t0 = time.perf_counter(); n, tr = sp2.run_steps(30, tr); sp2.synchronize(); dt = time.perf_counter() - t0
        measured["trial_step_us_small_problem"] = 1e6*dt/n
        lib, comm = sp2._lib, sp2._comm_handle
        stream = torch.cuda.current_stream()
        f = lib.mrcal_amd_comm_allreduce_sum
        import ctypes as C
        f.restype, f.argtypes = C.c_bool, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        coll = {}
        for name, ndoubles in (("second: 4 scalars", 4), ("first, metric's problem: Nc = 140", 140*140 + 2*140 + 2),
                               ("first, configuration 3: Nc = 284", 284*284 + 2*284 + 2), ("first, configuration 2: Nc = 1206", 1206*1206 + 2*1206 + 2)):
            buf = torch.zeros(ndoubles, dtype=torch.float64, device="cuda")
            for _ in range(5): f(comm, buf.data_ptr(), ndoubles, stream.cuda_stream)
            torch.cuda.synchronize(); dist.barrier()
            us = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream); ok = f(comm, buf.data_ptr(), ndoubles, stream.cuda_stream); e1.record(stream)
                e1.synchronize(); us.append(1e3*e0.elapsed_time(e1))
            coll[name] = dict(doubles=ndoubles, us_median=float(np.median(us)), us_min=float(min(us)), us_max=float(max(us)))
        measured["allreduce"] = coll
        sp2.close()
    except Exception as e:          # (the numbers are a by-product: the test's verdict does not hang on them)
        measured["error"] = f"{type(e).__name__}: {e}"
    gathered = [None]*world
    dist.all_gather_object(gathered, measured)
    if rank == 0:
        np.savez(out_path, b=b, rms=st["rms_reproj_error__pixels"], norm2_x=st["norm2_x"],
                 Noutliers=st["Noutliers_board"], replicated=bool(torch.equal(lo, hi)),
                 Ncollectives=sp.Ncollectives, Nevaluations=st["Nevaluations"])
        rec = dict(what=f"tests/test_parallel_gpu.py _rccl_world2_worker at world {world}: ShardedProblem(_driver='rccl'), one process per rank and device; "
                        "the all-reduces are the library's own (csrc/comm.cpp) timed alone on a stream with event pairs",
                   world=world, devices=torch.cuda.device_count(), device_name=torch.cuda.get_device_name(0), ranks=gathered)
        for d in ("gpurun_out", "profiles"):
            try:
                with open(os.path.join(ROOT, d, f"r06_rccl_world{world}_measured.json"), "w") as fjson: json.dump(rec, fjson, indent=1)
            except OSError:
                pass
        print(json.dumps(rec, indent=1))
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world2_on_two_devices(amd, tmp_path):
    """The product path with MORE THAN ONE RANK: two processes, two GPUs, ShardedProblem(_driver="rccl") - the
    collectives are ncclAllReduce calls issued by libmrcal_amd.so between its own kernel launches. Skips itself
    on a one-GPU box (the boxes of this pool); runs the first time it finds two devices. Same optimum, outliers
    and cost as the single-GPU solve (the sums over two shards differ from the unsharded ones in the last bits,
    so the trajectories may differ by rounding: compared to the solve's own tolerance, not bit for bit)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, this box has {torch.cuda.device_count()}")
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    oi = _problem(amd._api)
    with Problem(**oi) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "rccl2.npz")
    port = 29500 + (os.getpid() % 250)
    mp.spawn(_rccl_world2_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    assert bool(r["replicated"])
    assert int(r["Noutliers"]) == s1["Noutliers_board"]
    assert abs(float(r["norm2_x"]) - s1["norm2_x"]) < 1e-9*s1["norm2_x"]
    assert np.abs(r["b"] - b1).max() < 2e-5
    assert int(r["Ncollectives"]) >= 2*int(r["Nevaluations"])


def test_rccl_worker_measures_at_world1(amd, tmp_path):
    """The worker of the two-device test above, at world 1 on the one device this pool's boxes have: the same code - the
    RCCL communicator, the sharded solve, the timed steps, the event-timed all-reduces - so that the measuring part is known
    to run before a box with two devices ever sees it; its record (gpurun_out/ and profiles/r06_rccl_world1_measured.json)
    is what one-rank collectives cost. The solve must be the single-GPU solve's bits (test_rccl_world1... says the same)"""
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    oi = _problem(amd._api)
    with Problem(**oi) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "rccl1.npz")
    port = 29800 + (os.getpid() % 150)
    mp.spawn(_rccl_world2_worker, args=(1, port, out), nprocs=1, join=True)
    r = np.load(out)
    assert bool(r["replicated"]) and int(r["Noutliers"]) == s1["Noutliers_board"]
    assert np.array_equal(r["b"], b1)
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_rccl_world1_measured.json")))
    assert rec["world"] == 1 and "allreduce" in rec["ranks"][0], rec["ranks"][0]


@pytest.mark.timeout(900)
def test_sharded_metric_size_outliers_against_the_references_record(amd, tmp_path):
    """since round 5's arithmetic the 8-rank solve of the metric's problem and the single-GPU
    solve mark 7843 outliers each but not the same 7843 (one corner at the k-sigma line falls the other way) - and nothing
    said which of the two the REFERENCE's mask equals. Here both are solved on the inputs the record of the reference's own
    mrcal_optimize() was made from (tests/golden/reference_solve_ns_seed0.npz: 80 s of one core, in the build container)
    and both masks are compared with the recorded one: the counts must agree, whatever differs must be a handful of corners
    AT the line, and the rms must be the reference's to 1e-5 - and no higher - where the marks are identical. Which of the two (or both)
    reproduces the reference's marks corner by corner is printed and recorded, not imposed: the sums of eight shards and of
    one GPU are both legitimate roundings of the same numbers"""
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    from mrcal_amd.synthetic import copy_inputs
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_reference_solves import inputs_hash
    rec = np.load(os.path.join(ROOT, "tests", "golden", "reference_solve_ns_seed0.npz"))
    oi = _host_case(amd._api, "metric_size_recorded")
    assert inputs_hash(oi) == str(rec["inputs_sha256"]), "the synthesized inputs are not the record's"
    o1 = copy_inputs(oi)
    s1 = amd.optimize(**o1)
    mask_1 = (o1["observations_board"][..., 2] < 0).ravel()
    mask_r = np.unpackbits(rec["outlier_mask_packed"])[:mask_1.size].astype(bool)
    out = str(tmp_path / "host8.npz")
    port = 29700 + (os.getpid() % 250)
    mp.spawn(_host_worker, args=(8, port, out, "metric_size_recorded"), nprocs=8, join=True)
    r = np.load(out)
    mask_8 = r["outlier_mask"]
    d1, d8 = int((mask_1 != mask_r).sum()), int((mask_8 != mask_r).sum())
    rms_r = float(rec["rms_reproj_error__pixels"])
    print(f"outliers: reference {int(mask_r.sum())}, single GPU {int(mask_1.sum())} ({d1} marks differ), 8 ranks {int(mask_8.sum())} ({d8} marks differ); "
          f"rms reference {rms_r!r}, single GPU {s1['rms_reproj_error__pixels']!r}, 8 ranks {float(r['rms'])!r}")
    import json
    recd = dict(problem="8 cameras x 1000 frames OPENCV8, seed 0 (bench.py's), inputs made with the reference's library",
                outliers=dict(reference=int(mask_r.sum()), single_gpu=int(mask_1.sum()), ranks8_host_transport=int(mask_8.sum())),
                marks_differing_from_the_references=dict(single_gpu=d1, ranks8_host_transport=d8),
                rms=dict(reference=rms_r, single_gpu=float(s1["rms_reproj_error__pixels"]), ranks8_host_transport=float(r["rms"])))
    for d in ("gpurun_out", "profiles"):
        try:
            with open(os.path.join(ROOT, d, "r06_ns_sharded_outliers_vs_reference.json"), "w") as f: json.dump(recd, f, indent=1)
        except OSError: pass
    assert bool(r["replicated"])
    assert abs(int(mask_1.sum()) - int(mask_r.sum())) <= 2 and abs(int(mask_8.sum()) - int(mask_r.sum())) <= 2
    assert d1 <= 4 and d8 <= 4, (d1, d8)
    assert min(d1, d8) == 0, "neither the single-GPU nor the 8-rank solve reproduces the reference's marks corner by corner"
    # (observed: 0 and 0 marks differ - both solves reproduce the reference's 7843 corner by corner -, and the rms of either is
    #  1.5e-6 BELOW the recorded one: at this seed the restated libdogleg stops a little earlier in OPENCV8's flat valley)
    for d, rms in ((d1, float(s1["rms_reproj_error__pixels"])), (d8, float(r["rms"]))):
        assert abs(rms - rms_r) < (1e-5 if d == 0 else 1e-4)*rms_r
        assert rms <= rms_r*(1. + 1e-9)


# ---- discrete points sharded by point, triangulated points by point set --------------------
def _sfm_with_everything(api, seed=9):
    """boards AND triangulated points AND discrete points (some fixed, listed out of point order) in one problem,
    intrinsics locked; the discrete points sit where the truth geometry puts them, so that the optimum is well
    determined"""
    from test_triangulated import sfm_problem, R_from_r
    oi, truth = sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=120, seed=seed, noise=0.3, Nboard_frames=9)
    rng = np.random.RandomState(seed + 5)
    Np, Nfixed = 11, 3
    pts = np.column_stack((rng.uniform(-2, 1, Np), rng.uniform(-1, 1, Np), rng.uniform(5, 9, Np)))
    # every point's first observation in point order (the wrapper wants the indices to extend the set one at a
    # time, mrcal-pywrap.c:1159-1204), the others after them in random order: NOT sorted by point
    cams = [np.sort(rng.choice(4, size=3, replace=False)) for _ in range(Np)]
    pairs = [(ip, cams[ip][0]) for ip in range(Np)]
    rest  = [(ip, c) for ip in range(Np) for c in cams[ip][1:]]
    pairs += [rest[i] for i in rng.permutation(len(rest))]
    idx, obs = [], []
    for ip, ic in pairs:
        rt = truth["rt_cam_ref"][ic-1] if ic > 0 else None
        pc = pts[ip] if rt is None else R_from_r(rt[:3]) @ pts[ip] + rt[3:]
        q = api.project(pc[None], oi["lensmodel"], oi["intrinsics"][ic])[0] + rng.normal(0, 0.3, 2)
        idx.append((ip, ic, ic-1)); obs.append((q[0], q[1], rng.uniform(0.5, 1.5)))
    idx = np.array(idx, dtype=np.int32); obs = np.array(obs)
    obs[2,2] = -1.
    oi.update(points = np.ascontiguousarray(pts + np.r_[rng.normal(0, 0.05, (Np-Nfixed,3)), np.zeros((Nfixed,3))]),
              observations_point = np.ascontiguousarray(obs), indices_point_camintrinsics_camextrinsics = idx,
              Npoints_fixed = Nfixed)
    return oi


@pytest.mark.parametrize("world", (2, 3))
def test_shards_of_points_and_pairs_add_up(amd, world):
    """what the all-reduce relies on, for every kind of row at once: the normal equations of the shards - boards
    by frame, discrete points by point, triangulated points by point set, regularization with the leader - sum
    to the unsharded ones, every measurement row is in exactly one shard, every eliminated block in exactly one"""
    from mrcal_amd.resident import Problem
    from mrcal_amd.parallel import partition_frames, partition_points, partition_triangulated
    from mrcal_amd.synthetic import copy_inputs
    oi = _sfm_with_everything(amd._api)
    p = amd._api._ingest(dict(oi), callback=False)
    fr = partition_frames(p.c_board["iframe"].reshape(-1,1), p.Nframes, world)
    pr = partition_points(np.column_stack((p.c_point["i_point"],)*3), p.Npoints, world)
    tr = partition_triangulated(p.c_tri["flags"] & 1, world)
    with Problem(**copy_inputs(oi)) as p0:
        ne = p0.normal_equations()
        Nmeas, Nnz = p0.Nmeas, p0.Nnz
    acc, Nm, Nz = None, 0, 0
    for r in range(world):
        with Problem(_shard=fr[r], _leader=(r == 0), _shard_points=pr[r], _shard_tripoints=tr[r], **copy_inputs(oi)) as ps:
            n = ps.normal_equations()
            Nm += ps.Nmeas; Nz += ps.Nnz
        if acc is None: acc = {k: np.array(n[k], dtype=float) for k in ("A", "Bt", "D", "g")}; acc["norm2_x"] = n["norm2_x"]
        else:
            for k in ("A", "Bt", "D", "g"): acc[k] += n[k]
            acc["norm2_x"] += n["norm2_x"]
    assert Nm == Nmeas and Nz == Nnz
    for k in ("A", "Bt", "D", "g"):
        assert np.abs(acc[k] - ne[k]).max() < 1e-10*np.abs(ne[k]).max(), k
    assert abs(acc["norm2_x"] - ne["norm2_x"]) < 1e-10*ne["norm2_x"]


def test_rccl_world1_solves_points_and_pairs(amd):
    """ShardedProblem no longer refuses triangulated points: the product path (collectives from C++, outlier
    logic of the pairs per shard with the variance summed over the ranks) with a world of one == the single-GPU
    solve, outliers of both kinds included"""
    from mrcal_amd.resident import Problem
    from mrcal_amd.parallel import ShardedProblem
    from mrcal_amd.synthetic import copy_inputs
    oi = _sfm_with_everything(amd._api)
    oi["do_apply_outlier_rejection"] = True
    oi["observations_board"][3,2,4,:2] += 40.       # something to throw out
    with Problem(**copy_inputs(oi)) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    sp = ShardedProblem(_driver="rccl", **copy_inputs(oi))
    s2 = sp.solve()
    b2 = sp.b_packed()
    sp.close()
    assert s2["Noutliers_board"] == s1["Noutliers_board"] > 0
    assert abs(s2["norm2_x"] - s1["norm2_x"]) < 1e-9*s1["norm2_x"]
    assert np.abs(b1 - b2).max() < 1e-6


def _pairs_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    oi = _sfm_with_everything(mrcal_amd._api)
    oi["do_apply_outlier_rejection"] = False
    sp = ShardedProblem(_driver="python", **oi)
    st = sp.solve()
    if rank == 0:
        np.savez(out_path, b=sp.b_packed(), rms=st["rms_reproj_error__pixels"], norm2_x=st["norm2_x"],
                 ranges=np.array((sp.frame_range, sp.point_range, sp.tripoint_range)))
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


def test_world2_points_and_pairs(amd, tmp_path):
    """two ranks (one device, protocol driver over gloo): boards, discrete points and triangulated points each
    split between the ranks; the same optimum as the single-GPU solve"""
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    from mrcal_amd.synthetic import copy_inputs
    oi = _sfm_with_everything(amd._api)
    oi["do_apply_outlier_rejection"] = False
    with Problem(**copy_inputs(oi)) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "w2sfm.npz")
    port = 29600 + (os.getpid() % 250)
    mp.spawn(_pairs_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    # every kind really was split
    assert all(0 < hi - lo for lo, hi in r["ranges"])
    assert abs(float(r["norm2_x"]) - s1["norm2_x"]) < 1e-9*s1["norm2_x"]
    assert np.abs(r["b"] - b1).max() < 1e-6


# ---- the C++ sharded solve at world > 1 on one device: collectives staged through the host (csrc/comm.cpp) ----
HOST_CASES = ("boards", "boards_splined", "everything", "fewer_frames_than_ranks", "metric_size")
def _host_case(api, which):
    from mrcal_amd.synthetic import make_calibration_problem
    if which == "metric_size_recorded":
        # the same problem with its perfect corners projected by the REFERENCE's library, as the record of the
        # reference's own solve of it was made (tests/golden/make_reference_solves.py ns_seed0)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from make_reference_solves import recorded_inputs
        from mrcal_amd._cabi import MrcalLib
        from mrcal_amd._api import Api
        return recorded_inputs("ns_seed0", Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so"))))
    if which == "metric_size":
        # the benchmark's own problem: 8 cameras x 1000 frames OPENCV8 (bench.py)
        return make_calibration_problem(api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8",
                                        object_width_n=10, object_height_n=10, seed=0)[0]
    if which == "boards":         oi = _problem(api)
    if which == "boards_splined": oi = _problem(api, SPLINED)
    if which == "everything":
        oi = _sfm_with_everything(api)
        oi["do_apply_outlier_rejection"] = True
        oi["observations_board"][3,2,4,:2] += 40.
    if which == "fewer_frames_than_ranks":
        # two frames for three ranks: one rank owns no board at all, and still takes part in every collective of
        # the outlier pass
        oi = make_calibration_problem(api, Ncameras=2, Nframes=2, lensmodel="LENSMODEL_OPENCV4",
                                      object_width_n=10, object_height_n=10, seed=3)[0]
        oi["do_optimize_intrinsics_distortions"] = False
        oi["observations_board"][1,2,4,:2] += 40.
    return oi


def _host_worker(rank, world, port, out_path, which):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MRCAL_AMD_HOST_COMM_TIMEOUT="60")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    oi_case = _host_case(mrcal_amd._api, which)
    sp = ShardedProblem(_driver="host", **oi_case)
    st = sp.solve()
    b = sp.b_packed()
    t = torch.from_numpy(b.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    counts = torch.tensor([st["Niterations"], st["Nevaluations"], st["Noutliers_board"], st["Noutlier_passes"]])
    cl, ch = counts.clone(), counts.clone()
    dist.all_reduce(cl, op=dist.ReduceOp.MIN); dist.all_reduce(ch, op=dist.ReduceOp.MAX)
    ranges = torch.zeros(world, 6, dtype=torch.int64)
    ranges[rank] = torch.tensor(sp.frame_range + sp.point_range + sp.tripoint_range)
    dist.all_reduce(ranges)
    # the outlier marks of the whole problem: every rank holds those of its own frames' observations (a contiguous run:
    # the observations are sorted by frame)
    mask = None
    if which == "metric_size_recorded":
        import ctypes as C
        iframe = oi_case["indices_frame_camintrinsics_camextrinsics"][:, 0]
        mine = np.nonzero((iframe >= sp.frame_range[0]) & (iframe < sp.frame_range[1]))[0]
        H, W = oi_case["observations_board"].shape[1:3]
        pool = np.zeros((len(mine), H, W, 3))
        assert sp._lib.mrcal_amd_problem_get_board_pool(sp.problem.handle, pool.ctypes.data_as(C.c_void_p))
        full = torch.zeros(len(iframe)*H*W, dtype=torch.int64)
        if len(mine): full[mine[0]*H*W:(mine[-1] + 1)*H*W] = torch.from_numpy((pool[..., 2] < 0).astype(np.int64).ravel())
        dist.all_reduce(full)
        mask = full.numpy().astype(bool)
    if rank == 0:
        np.savez(out_path, b=b, rms=st["rms_reproj_error__pixels"], norm2_x=st["norm2_x"],
                 Noutliers=st["Noutliers_board"],
                 replicated=bool(torch.equal(lo, hi) and torch.equal(cl, ch)),
                 Ncollectives=sp.Ncollectives, Nevaluations=st["Nevaluations"], ranges=ranges.numpy(),
                 **({} if mask is None else {"outlier_mask": mask}))
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which,world", (("boards", 2), ("boards_splined", 2), ("everything", 2), ("everything", 3),
                                         ("fewer_frames_than_ranks", 3), ("metric_size", 8)))
def test_cpp_sharded_solve_over_the_host_transport(amd, tmp_path, which, world):
    """The PRODUCT solver with more than one rank on a one-GPU box: ShardedProblem(_driver="host") is the C++
    sharded solve - the device-controlled step with its two all-reduces, mark_outliers() with its three, the
    final gather - with mrcal_amd_comm_allreduce_sum() staged through a shared-memory segment instead of RCCL
    (which refuses two ranks on one device). Same outliers, cost and optimum as the single-GPU solve; every rank
    ends with the same state and the same counters. The last case leaves a rank without a single board: its
    collectives must still pair up with the others' (a mismatch is a timeout here, reported, not a hang)"""
    import torch.multiprocessing as mp
    from mrcal_amd.resident import Problem
    from mrcal_amd.synthetic import copy_inputs
    oi = _host_case(amd._api, which)
    with Problem(**copy_inputs(oi)) as p:
        s1 = p.solve()
        b1 = p.b_packed()
    out = str(tmp_path / "host.npz")
    port = 29700 + (os.getpid() % 250)
    mp.spawn(_host_worker, args=(world, port, out, which), nprocs=world, join=True)
    r = np.load(out)
    assert bool(r["replicated"])
    fr = r["ranges"][:, 0:2]
    if which == "fewer_frames_than_ranks":
        assert (fr[:,1] - fr[:,0] == 0).any() and s1["Noutliers_board"] > 0
    else:
        assert (fr[:,1] - fr[:,0] > 0).all()
    if which == "everything":
        assert (r["ranges"][:,3] > r["ranges"][:,2]).all() and (r["ranges"][:,5] > r["ranges"][:,4]).all()
        assert s1["Noutliers_board"] > 0
    assert int(r["Noutliers"]) == s1["Noutliers_board"]
    # (the metric's problem: 7843 outliers out of 800 000 corners in both - but since round 5 not the same 7843: the
    #  eight shards' sums and the one GPU's differ in their last bits, and with this round's arithmetic
    #  (-ffp-contract=on) ONE corner at the k-sigma line is an outlier in the one solve and its neighbour in the other.
    #  Both solves converge (expected improvement 1e-10 at the end of either: tools/exp/dbg_sharded8.py); their costs
    #  differ by that corner's 12 units of 3.5 million. The other cases mark the same corners: 1e-9)
    assert abs(float(r["norm2_x"]) - s1["norm2_x"]) < (1e-5 if which == "metric_size" else 1e-9)*s1["norm2_x"]
    db = np.abs(r["b"] - b1)
    if which == "metric_size":
        # (with another corner thrown out OPENCV8's weakly determined directions - the high-order distortions - end 0.3
        #  packed units elsewhere, the rest of the 6140 variables within 1e-4)
        assert np.median(db) < 1e-4 and np.percentile(db, 99) < 1e-2, (np.median(db), np.percentile(db, 99), db.max())
    else:
        assert db.max() < (1e-3 if which == "boards_splined" else 2e-5)
    assert int(r["Ncollectives"]) >= 2*int(r["Nevaluations"])


@pytest.mark.parametrize("world", (2, 4))
def test_bench_multi_rank_path_on_one_device(world):
    """bench.py --gpus N end to end at N > 1 on a one-GPU box: it starts its own ranks (torch.distributed.run on
    127.0.0.1), every rank builds its shard, the timed region is bracketed by the barriers, the wall clock is the
    max over the ranks and rank 0 prints the one line - with MRCAL_AMD_BENCH_ONE_DEVICE=1, which puts the ranks on
    device 0 and the solve's collectives on the host transport (the line says so; it is not a measurement)"""
    import json, subprocess
    env = dict(os.environ, MRCAL_AMD_BENCH_ONE_DEVICE="1", MRCAL_AMD_HOST_COMM_TIMEOUT="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"): env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
                        "--cameras", "3", "--frames", "40"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and abs(d["value"] - 1e3/d["ms_per_step"]) < 1e-6*d["value"]
    assert d["scaling"] == "strong" and d["unit"] == "iterations/s" and "NOT a measurement" in d["transport"]
    assert d["solver"]["collectives"] >= 2*d["solver"]["evaluations"]
    assert d["roofline"]["launches_timed"] > 0 and d["cpu_baseline"] is None
    # the line checks itself: every rank reported, the same collectives on every rank, the transport's own count of
    # the ranks, each rank's shard of the Jacobian build against the roofline, max and min of the ranks' clocks
    assert len(d["ranks"]) == world and [r["rank"] for r in d["ranks"]] == list(range(world))
    assert d["consistent"] and d["comm_world_observed"] == [world]
    assert all(r["collectives"] == d["solver"]["collectives"] and r["collective_bytes"] > 0 for r in d["ranks"])
    assert sum(r["frames"][1] - r["frames"][0] for r in d["ranks"]) == 40
    assert all(r["roofline_frac"] > 0 for r in d["ranks"])
    assert d["rank_seconds_min"] <= d["rank_seconds_max"] and abs(d["rank_seconds_max"]*1e3/6 - d["ms_per_step"]) < 1e-6
