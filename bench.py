#!/usr/bin/env python3
"""Benchmark of the optimize() hot path on MI355X.

Metric (BASELINE.json): dog-leg ("LM") iterations per second on the 8-camera x
1000-frame OPENCV8 chessboard calibration, plus the wall-clock of one full solve.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
           --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one dog-leg step of the solver on the whole problem: evaluation of
all residuals x and of the CSR Jacobian J at the trial state (written to HBM in
the reference's layout), Jt x and the block normal equations, the
Schur-complement Cholesky factorization when the trust region asks for the
Gauss-Newton step, step selection and the accept/reject test. That is what one
iteration of libdogleg costs the reference (one optimizer_callback() + one
CHOLMOD factorize/solve). The timed region starts with everything resident in
HBM; steps continue the solve from the seed (no artificial repetition of a
converged state).

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix = FP64 vector peak (vendor figure); measured ceiling
                               # of v_mfma_f64_4x4x4 on this chip: 26 flop/clk/SIMD = 63.9 TFLOP/s (tools/exp/launch_floor.hip)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus",   type=int, default=1)
    ap.add_argument("--steps",  type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cameras", type=int, default=8)
    ap.add_argument("--frames",  type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-iterations", type=int, default=4)
    ap.add_argument("--no-full-solve", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other single-GPU BASELINE.json configurations (the line's configs[])")
    ap.add_argument("--only-config", default=None,
                    help="dev: time just this entry of configs[] (1, 2, 3, 4 or 5) and print it instead of the line")
    ap.add_argument("--sharded", action="store_true",
                    help="diagnostic: take the multi-GPU code path (frame shards + RCCL all-reduces from C++) even with one rank")
    return ap.parse_args()


def cpu_baseline(oi, Niterations):
    """The reference on the host cores of this box, one thread (the reference is single-threaded), on the SAME
    problem, bounded to ~30 s. The reference's own code is its callback (oracle/_ref/libmrcal_ref.so: mrcal.c compiled
    in place); libdogleg + CHOLMOD are not installed, so the solve beside it is timed three ways:

      restated-libdogleg   the reference's mrcal_optimize() over oracle/dogleg_restated.c (simplicial up-looking
                           sparse Cholesky), capped at a few accepted iterations
      schur-numpy-lapack   the product's own algorithm on the host (oracle/schur_numpy.py): per-observation Grams,
                           batched 6x6 dpotrf of the frames, one dense dpotrf of the camera block
      scipy-superlu        the same blocks assembled into a sparse JtJ for scipy.sparse.linalg.splu (an LU, not a Cholesky)

    value = the fastest EXACT-CHOLESKY variant's trial steps per second; value_upper_bound_callback_only = the
    reference's callback with a solve of zero cost"""
    path = os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")
    if not os.path.exists(path):
        return None
    from mrcal_amd._cabi import MrcalLib
    from mrcal_amd._api  import Api
    from mrcal_amd.synthetic import copy_inputs
    ref = Api(MrcalLib(path))
    variants = []

    o = copy_inputs(oi)
    o["do_apply_outlier_rejection"] = False
    ref.clib.dogleg_restated_set_max_iterations(int(Niterations))
    t0 = time.perf_counter()
    ref.optimize(**o)
    dt = time.perf_counter() - t0
    ref.clib.dogleg_restated_set_max_iterations(0)
    n = [C.c_int(0) for _ in range(3)]
    ref.clib.dogleg_restated_last_counts(*[C.byref(v) for v in n])
    tc, tf = C.c_double(0), C.c_double(0)
    ref.clib.dogleg_restated_last_timing(C.byref(tc), C.byref(tf))
    Nsteps, Ncallbacks, Nfact = [v.value for v in n]
    # a step = one evaluation at a trial point (accepted or not), like ours
    Ntrials = max(Ncallbacks - 1, 1)
    callback_ms = 1e3*tc.value/max(Ncallbacks,1)
    variants.append(dict(name = "restated-libdogleg", exact_cholesky = True, value = Ntrials/dt, trial_steps = Ntrials, seconds = dt,
                         callback_ms_per_evaluation = callback_ms, solve_ms_each = 1e3*tf.value/max(Nfact,1),
                         what = f"the reference's mrcal_optimize(), first {Nsteps} accepted iterations, over the restated libdogleg "
                                "(oracle/dogleg_restated.c: simplicial up-looking sparse Cholesky standing in for CHOLMOD)"))

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import schur_numpy
    from threadpoolctl import threadpool_limits
    Nhost = max(10, Ntrials)
    with threadpool_limits(1):
        for name, solver, exact, what in (
                ("schur-numpy-lapack", schur_numpy.gauss_newton_step_schur, True,
                 "the reference's optimizer_callback() + the product's own algorithm on the host (oracle/schur_numpy.py): "
                 "per-observation Grams, batched 6x6 dpotrf of the frames, one dense dpotrf of the camera block"),
                ("scipy-superlu", schur_numpy.gauss_newton_step_superlu, False,
                 "the reference's optimizer_callback() + the same blocks assembled into a sparse JtJ for scipy.sparse.linalg.splu "
                 "(SuperLU: an LU with partial pivoting, NOT a Cholesky)")):
            r = schur_numpy.timed_trial_steps(ref, oi, Nhost, solver, copy_inputs)
            variants.append(dict(name = name, exact_cholesky = exact, value = r["Ntrials"]/r["seconds"], trial_steps = r["Ntrials"],
                                 seconds = r["seconds"], callback_ms_per_evaluation = 1e3*r["seconds_callback"]/r["Ntrials"],
                                 solve_ms_each = 1e3*r["seconds_solve"]/max(r["Nsolves"], 1), solves = r["Nsolves"],
                                 cost_ratio = r["cost1"]/r["cost0"], what = what))
    best = max((v for v in variants if v["exact_cholesky"]), key=lambda v: v["value"])
    seconds = sum(v["seconds"] for v in variants)
    return dict(value  = best["value"],
                unit   = "iterations/s",
                cores  = 1,
                host_cores = os.cpu_count(),
                host_cores_usable = (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()),
                kind   = "reference",
                sample = f"same problem, first trial steps from the seed: {', '.join(str(v['trial_steps']) + ' (' + v['name'] + ')' for v in variants)}; "
                         f"value = {best['name']}; {seconds:.1f} s of one core in all",
                seconds = seconds,
                variants = variants,
                callback_ms_per_evaluation = callback_ms,
                # what is the reference's OWN code in these figures is the callback; the reference cannot be faster than
                # its callback alone:
                value_upper_bound_callback_only = 1e3/callback_ms if callback_ms > 0 else None,
                cores_note = "cores = the threads used: the reference's optimize() path is single-threaded (mrcal.c has no threads; "
                             "BLAS pinned to 1), host_cores = what this box has",
                note = "the reference's libdogleg + CHOLMOD are not installed here: value = the fastest exact-Cholesky stand-in "
                       "beside the reference's own callback; value_upper_bound_callback_only = evaluations/s of the reference's "
                       "optimizer_callback() alone (a solve of zero cost). The reference with the real CHOLMOD lies in "
                       "[value, value_upper_bound_callback_only]")


# The other configurations BASELINE.json lists (the metric's own is the line's value): every one of them fits ONE
# MI355X, so every one of them is timed here, on rank 0 of a single-GPU run, AFTER the line's timed region and
# outside it - a witnessed number per configuration instead of a table made by hand.
# configs[0] (1 camera x 40 frames OPENCV4) is the reference's own CPU-runnable plumbing case: a parity test, not a
# bench line. Per entry: the problem's sizes, the seed, ms per trial step over STEPS trial steps from the seed (after
# WARMUP untimed ones), the full solve on a fresh copy, and - where there are boards - the Jacobian kernel's roofline
# from HIP event pairs around its launches during EXTRA further steps (not during the timed ones: a pair costs the
# stream 11 us)
def other_configurations(only=None):
    import numpy as np
    import torch
    import mrcal_amd
    from mrcal_amd.resident  import Problem
    from mrcal_amd.synthetic import make_calibration_problem, make_sfm_problem, copy_inputs, CONFIG2_LENSMODEL
    SEED_BOARDS, SEED_SFM, STEPS, WARMUP, EXTRA = 0, 6, 20, 3, 12
    traffic_by_config = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r06_jacobian_kernel_hbm_traffic.json")))
        traffic_by_config = {k: v for k, v in tj.get("configs", {}).items()}
    except Exception:
        pass
    def boards(**kw):
        return make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, seed=SEED_BOARDS, **kw)[0]
    table = (
        ("1", "BASELINE.json configs[1]: 4 cameras x 400 frames x 10x10 corners, LENSMODEL_OPENCV8, all variables optimized, warp + regularization",
         SEED_BOARDS, lambda: boards(Ncameras=4, Nframes=400, lensmodel="LENSMODEL_OPENCV8")),
        ("2", f"BASELINE.json configs[2]: 1 camera x 800 frames x 10x10 corners, {CONFIG2_LENSMODEL} (30x20 control points over 150 degrees), "
              "core locked (mrcal-calibrate-cameras' recipe for the splined models), frames + warp + regularization",
         SEED_BOARDS, lambda: boards(Ncameras=1, Nframes=800, lensmodel=CONFIG2_LENSMODEL, do_optimize_intrinsics_core=False)),
        ("3", "BASELINE.json configs[3]: 16 cameras x 2000 frames x 10x10 corners, LENSMODEL_OPENCV8, unsharded on ONE MI355X (it fits)",
         SEED_BOARDS, lambda: boards(Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8")),
        ("4", "BASELINE.json configs[4] without boards: SfM, 4 cameras LENSMODEL_OPENCV4 (intrinsics locked), 20000 triangulated points, "
              "extrinsics optimized, unity_cam01 regularization, unsharded on ONE MI355X",
         SEED_SFM, lambda: make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=SEED_SFM, noise=0.3)[0]),
        ("5", "BASELINE.json configs[4]: the same + 400 board frames (1600 board observations), frames optimized too",
         SEED_SFM, lambda: make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=SEED_SFM, noise=0.3, Nboard_frames=400)[0]))
    out = []
    for key, workload, seed, make in table:
        if only is not None and key != str(only):
            continue
        entry = dict(config = key, workload = workload, seed = seed, data = "synthetic", n_gpus = 1)
        try:
            oi = make()
            with Problem(**copy_inputs(oi)) as p:
                _, tr = p.run_steps(WARMUP, None)
                p.synchronize(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                n, tr = p.run_steps(STEPS, tr)
                p.synchronize(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                st = p.solver_stats()
                entry.update(Nstate = p.Nstate, Nmeasurements = p.Nmeas, Nnz_J = p.Nnz, steps = n, warmup = WARMUP,
                             ms_per_step = 1e3*dt/n, iterations_per_s = n/dt,
                             solver = dict(evaluations = st["Nevaluations"], factorizations = st["Nfactorizations"]))
                alg = p.jacobian_algorithmic_bytes()
                if alg > 0:
                    p.jacobian_timing_begin(EXTRA + 2, 1)
                    p.run_steps(EXTRA, tr)
                    p.synchronize()
                    nl, ktot, kmin, kmax = p.jacobian_timing_end()
                    if nl > 0 and ktot > 0:
                        kms = ktot/nl
                        ach = alg/1e9/(kms*1e-3)
                        entry["roofline"] = dict(bound = "hbm", kernel = "the board Jacobian build of this configuration "
                                                 "(board_kernel / board_splined_rows_kernel: residuals x, CSR Jacobian values" +
                                                 ("" if "SPLINED" in workload else ", per-observation Gram") + ")" +
                                                 (" with the triangulated pairs in the same launch (board_tri_kernel: their bytes are in algorithmic_bytes_per_launch)"
                                                  if key == "5" else ""),
                                                 achieved = ach, peak = HBM_PEAK_GBS, unit = "GB/s", frac = ach/HBM_PEAK_GBS,
                                                 algorithmic_bytes_per_launch = alg, kernel_ms_avg = kms, kernel_ms_min = kmin,
                                                 kernel_ms_max = kmax, launches_timed = nl, timed_every = 1,
                                                 timed_in = f"{EXTRA} further trial steps behind the {n} timed ones",
                                                 traffic = traffic_by_config.get(key, {}).get("hbm_bytes_per_launch"),
                                                 traffic_source = None if key not in traffic_by_config else
                                                     "committed constant: profiles/r06_jacobian_kernel_hbm_traffic.json (rocprofv3 --pmc "
                                                     "WRITE_SIZE / FETCH_SIZE, a pass each; not re-measured in this run)")
            entry.update(full_solves(oi, Problem, copy_inputs, torch))
            # (the product mode of round 6, never the metric: the same steps without the Jacobian stream, where the
            #  problem has one to leave out)
            with Problem(**copy_inputs(oi)) as p3:
                if p3.jacobian_stream_is_optional():
                    p3.set_jacobian_stream(False)
                    _, tr = p3.run_steps(WARMUP, None)
                    p3.synchronize(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n3, tr = p3.run_steps(STEPS, tr)
                    p3.synchronize(); torch.cuda.synchronize()
                    entry["ms_per_step_no_jacobian_stream"] = 1e3*(time.perf_counter() - t0)/n3
        except Exception as e:      # a configuration that fails says so in its entry; the line is still printed
            entry["error"] = f"{type(e).__name__}: {e}"
        out.append(entry)
    return out


def full_solves(oi, Problem, copy_inputs, torch):
    """The second half of the metric: "one mrcal.optimize() with outlier rejection, from seed to
    return" - the DROP-IN call, host arrays in, host arrays out, through the C ABI's mrcal_optimize()
    (/root/reference/mrcal-pywrap.c:2149 -> mrcal.c:6179): problem creation, the observations across PCIe, the CSR
    structure, the solve with its outlier passes, the results back, teardown. With the Jacobian stream ON in every
    step (the metric's definition of a step). Beside it, separately named and never the metric:
      full_solve_resident                      the solve alone on a problem already resident in HBM (what rounds 1-5
                                               reported as full_solve), stream on
      full_solve_no_jacobian_stream            the drop-in call as the product runs it by default (round 6: nothing
                                               reads the CSR values a solve's steps would write; same bits out)
      full_solve_resident_no_jacobian_stream   the resident solve in that mode"""
    import mrcal_amd
    out = {}
    def stats(s, dt):
        d = dict(seconds = dt, rms_reproj_error__pixels = s["rms_reproj_error__pixels"], Noutliers_board = s["Noutliers_board"])
        for k_out, k_in in (("iterations", "Niterations"), ("evaluations", "Nevaluations"), ("outlier_passes", "Noutlier_passes")):
            if k_in in s: d[k_out] = s[k_in]
        return d
    for key, stream in (("full_solve", True), ("full_solve_no_jacobian_stream", False)):
        o = copy_inputs(oi)
        prev = mrcal_amd.set_optimize_jacobian_stream(stream)
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s = mrcal_amd.optimize(**o)
            dt = time.perf_counter() - t0
        finally:
            mrcal_amd.set_optimize_jacobian_stream(prev)
        out[key] = stats(s, dt)
        out[key]["what"] = "mrcal_amd.optimize(**optimization_inputs), seed to return, through the C ABI's mrcal_optimize(); " + \
                           ("every step streams the CSR Jacobian to HBM (the metric's step)" if stream else
                            "the product's default: the steps do not stream the CSR Jacobian nothing reads (NOT the metric)")
    for key, stream in (("full_solve_resident", True), ("full_solve_resident_no_jacobian_stream", False)):
        with Problem(**copy_inputs(oi)) as p2:
            if not stream and not p2.jacobian_stream_is_optional():
                continue
            p2.set_jacobian_stream(stream)
            p2.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            s2 = p2.solve()
            p2.synchronize()
            out[key] = stats(s2, time.perf_counter() - t0)
    return out


def callback_times(oi, copy_inputs, torch, N=3):
    """mrcal_amd.optimizer_callback(**optimization_inputs): the drop-in single evaluation, host arrays in and out
    (/root/reference/mrcal-pywrap.c:1890-2010), ms per call (the best of N after one warm-up call): without the
    Jacobian (x only), with it (rowptr, colidx and values across PCIe into fresh numpy arrays), and with the
    factorization it returns by default"""
    import mrcal_amd
    out = {}
    for key, kw in (("no_jacobian", dict(no_jacobian=True, no_factorization=True)),
                    ("with_jacobian", dict(no_factorization=True)),
                    ("with_jacobian_and_factorization", dict())):
        best = None
        for i in range(N + 1):
            o = copy_inputs(oi)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = mrcal_amd.optimizer_callback(**o, **kw)
            dt = time.perf_counter() - t0
            del r
            if i > 0: best = dt if best is None else min(best, dt)
        out[key] = 1e3*best
    return out


# MRCAL_AMD_BENCH_ONE_DEVICE=1: every rank on device 0, the solve's collectives staged through host shared memory
# (mrcal_amd_comm_create_host) instead of RCCL, which refuses two ranks per device. For checking THIS SCRIPT's
# multi-rank path - the launcher, the barriers, the max over ranks, rank 0's line - on a one-GPU box
# (tests/test_parallel_gpu.py); the line it prints says so and is not a measurement
ONE_DEVICE = os.environ.get("MRCAL_AMD_BENCH_ONE_DEVICE") == "1"


def spawn_ranks(ngpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, the way
    the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
    does, and pass rank 0's JSON line through. Returns the exit code"""
    import socket
    import subprocess
    import torch
    if torch.cuda.device_count() < ngpus and not ONE_DEVICE:
        print(f"bench.py: --gpus {ngpus} but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank       = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world      = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: taking the launcher's", file=sys.stderr)
        args.gpus = world

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU")
    torch.cuda.set_device(0 if ONE_DEVICE else local_rank)
    sharded = world > 1 or args.sharded
    if sharded:
        # torch.distributed is only the side channel of the start-up (the 128-byte
        # RCCL id, the barriers, the max over ranks of the wall clock): gloo. The
        # collectives of the solve are RCCL all-reduces issued by libmrcal_amd.so
        # itself on the problem's HIP stream (mrcal_amd/csrc/comm.cpp)
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29871", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("gloo")

    import mrcal_amd
    from mrcal_amd.synthetic import make_calibration_problem

    if args.only_config is not None:
        print(json.dumps(other_configurations(args.only_config)), flush=True)
        return

    oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=args.cameras, Nframes=args.frames,
                                     lensmodel="LENSMODEL_OPENCV8",
                                     object_width_n=10, object_height_n=10, seed=0)
    workload = f"{args.cameras} cameras x {args.frames} frames x 10x10 corners, LENSMODEL_OPENCV8, " \
               "all variables optimized, warp + regularization"

    if sharded:
        from mrcal_amd.parallel import ShardedProblem
        problem = ShardedProblem(_driver="host" if ONE_DEVICE else "rccl", **oi)
        barrier = lambda: (dist.barrier(), torch.cuda.synchronize())
    else:
        from mrcal_amd.resident import Problem
        problem = Problem(**oi)
        barrier = lambda: torch.cuda.synchronize()

    def sync_all():
        barrier()
        problem.synchronize()

    if not sharded:
        # clocks, code objects and allocator warm before anything is measured: a
        # scratch copy of the problem takes a few dozen steps and is thrown away.
        # (Not the --warmup steps: those belong to the measured solve's trajectory)
        from mrcal_amd.synthetic import copy_inputs
        scratch = Problem(**copy_inputs(oi))
        scratch.run_steps(30, None)
        scratch.synchronize()
        scratch.close()

    # untimed warmup, then EXACTLY --steps timed steps
    tr = None
    if args.warmup > 0:
        _, tr = problem.run_steps(args.warmup, tr)
    sync_all()
    # THE TIMED REGION: exactly --steps trial steps, nothing else on the stream (round 6: no event pairs in here at any
    # --steps - a pair costs the stream ~11 us; until round 5 every 4th or 8th board launch carried one). Every step
    # evaluates x AND streams the CSR values of J (the resident problem's default; said explicitly where it can be)
    if not sharded: problem.set_jacobian_stream(True)
    t0 = time.perf_counter()
    n, tr = problem.run_steps(args.steps, tr)
    sync_all()
    dt = time.perf_counter() - t0
    assert n == args.steps
    # the dominant kernel's own duration: HIP event pairs around EVERY board launch of EXTRA further trial steps of
    # the same solve, behind the timed ones (as configs[] does)
    EXTRA = 16
    problem.jacobian_timing_begin(EXTRA + 4, 1)
    problem.run_steps(EXTRA, tr)
    sync_all()
    nlaunch, ktot_ms, kmin_ms, kmax_ms = problem.jacobian_timing_end()
    TIMED_EVERY = 1

    dt_rank = dt
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = problem.solver_stats()
    per_rank = None
    if sharded:
        # what lets a scaling run be checked from its own line: every rank's wall clock over the
        # timed steps, its shard of the Jacobian build against the HBM roofline (shard bytes / shard kernel time), the
        # collectives it queued and how many ranks its transport says the communicator has
        ci = problem.comm_info()
        mine = dict(rank = rank, device = torch.cuda.current_device(), seconds = dt_rank,
                    frames = list(problem.frame_range),
                    board_kernel_ms_avg = (ktot_ms/nlaunch) if nlaunch else None,
                    algorithmic_bytes_per_launch = problem.jacobian_algorithmic_bytes(),
                    collectives = ci["Ncollectives"], collective_bytes = ci["bytes"],
                    comm_world_observed = ci["world_observed"])
        if mine["board_kernel_ms_avg"]:
            mine["roofline_achieved_GBs"] = mine["algorithmic_bytes_per_launch"]/1e9/(mine["board_kernel_ms_avg"]*1e-3)
            mine["roofline_frac"] = mine["roofline_achieved_GBs"]/HBM_PEAK_GBS
        per_rank = [None]*world
        dist.all_gather_object(per_rank, mine)

    # the dominant kernel: the board Jacobian build. Algorithmic bytes per
    # launch: per observation of P corners and k nonzeros per
    # row, 24 P (read qx,qy,w) + 16 P (write x) + 16 P k (write J values)
    alg_bytes  = problem.jacobian_algorithmic_bytes()
    kernel_ms  = ktot_ms/max(nlaunch,1)
    achieved   = alg_bytes/1e9/(kernel_ms*1e-3) if kernel_ms > 0 else 0.0
    traffic    = None
    tpath = os.path.join(ROOT, "profiles", "board_kernel_hbm_traffic.json")
    if os.path.exists(tpath) and not sharded:
        try:
            tj = json.load(open(tpath))
            if tj.get("workload_cameras") == args.cameras and tj.get("workload_frames") == args.frames:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # MFMA utilisation of the dense blocks (north_star): the Gram of the board kernel live (its
    # v_mfma_f64_4x4x4 flops over the same event-timed launches), the matrix-pipe busy share of the
    # board kernel and of the Schur SYRK from the committed PMC pass (profiles/)
    mfma = None
    try:
        NM = 8                                   # accumulators of the OPENCV8 tile (7 column blocks: problem.hpp)
        mfma_per_obs = NM*((2*10*10 + 3)//4)
        flops = 512.0*mfma_per_obs*args.cameras*args.frames/max(world, 1)
        mfma = dict(kernel = "board_kernel Gram, v_mfma_f64_4x4x4 (4 blocks x 4x4x4 x 2 flop = 512 flop each, "
                             f"{mfma_per_obs} per observation)",
                    flops_per_launch = flops,
                    achieved = flops/1e12/(kernel_ms*1e-3) if kernel_ms > 0 else 0.0,
                    peak = FP64_MFMA_PEAK_TFLOPS, unit = "TFLOP/s")
        mfma["frac"] = mfma["achieved"]/FP64_MFMA_PEAK_TFLOPS
        for mname in ("r06_mfma_utilisation.json", "r05_mfma_utilisation.json", "r04_mfma_utilisation.json", "r03_mfma_utilisation.json", "r02_mfma_utilisation.json"):
            mpath = os.path.join(ROOT, "profiles", mname)
            if not os.path.exists(mpath):
                continue
            mj = json.load(open(mpath))
            if mj.get("workload_cameras") == args.cameras and mj.get("workload_frames") == args.frames:
                mfma["pmc_matrix_pipe_busy_frac"] = {k: v["mfma_busy_frac"] for k, v in mj["ns"].items() if "mfma_busy_frac" in v}
                mfma["pmc_matrix_pipe_busy_frac_config2_splined"] = {k: v["mfma_busy_frac"] for k, v in mj["config2"].items() if "mfma_busy_frac" in v}
                mfma["pmc_source"] = f"committed constant: profiles/{mname} (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs), " \
                                     "a rocprofv3 --pmc pass of its own; not re-measured in this run)"
            break
    except Exception:
        mfma = None

    result = dict(
        metric  = "LM (dog-leg) iterations/sec, 8-cam x 1000-frame OPENCV8 calibration",
        value   = args.steps/dt,
        unit    = "iterations/s",
        n_gpus  = world,
        steps   = args.steps,
        warmup  = args.warmup,
        ms_per_step = 1e3*dt/args.steps,
        higher_is_better = True,
        scaling = "strong",
        vs_baseline = None,
        dtype   = "f64",
        data    = "synthetic",
        config  = dict(workload = workload,
                       Nstate = problem.Nstate_global, Nmeasurements = problem.Nmeas_global,
                       Nnz_J = problem.Nnz_global,
                       parallelism = "single GPU" if not sharded else f"frames sharded over {world} GPU(s), one process per GPU, 2 {'host-staged (validation)' if ONE_DEVICE else 'RCCL'} all-reduces per trial step: [S|r|g_S||x|^2] ({problem.problem.Nstate} state variables: Nc^2+2Nc+2 doubles) and 4 scalars"),
        roofline = dict(bound = "hbm",
                        kernel = "board_kernel<OPENCV,8,J,Gram> (residuals x, CSR Jacobian values, per-observation Gram on the FP64 matrix cores)",
                        achieved = achieved, peak = HBM_PEAK_GBS, unit = "GB/s",
                        frac = achieved/HBM_PEAK_GBS,
                        traffic = traffic,
                        traffic_source = None if traffic is None else
                            "committed constant: profiles/board_kernel_hbm_traffic.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes, "
                            "not re-measured in this run)",
                        algorithmic_bytes_per_launch = alg_bytes,
                        kernel_ms_avg = kernel_ms, kernel_ms_min = kmin_ms, kernel_ms_max = kmax_ms,
                        launches_timed = nlaunch, timed_every = TIMED_EVERY,
                        timed_in = f"{EXTRA} further trial steps of the same solve behind the {args.steps} timed ones (no event pair inside the timed region)",
                        mfma = mfma),
        solver = dict(evaluations = st["Nevaluations"], factorizations = st["Nfactorizations"],
                      **({"collectives": st["Ncollectives"]} if sharded else {})),
    )
    if per_rank is not None:
        secs = [r["seconds"] for r in per_rank]
        result["ranks"] = per_rank
        result["rank_seconds_max"] = max(secs)
        result["rank_seconds_min"] = min(secs)
        result["comm_world_observed"] = sorted(set(r["comm_world_observed"] for r in per_rank))
        # (every rank queues the same collectives - the control block is replicated -; a scaling line whose ranks
        #  disagree, or whose transport counts another number of ranks than the launcher's, measured something else)
        result["consistent"] = (len(set(r["collectives"] for r in per_rank)) == 1 and
                                result["comm_world_observed"] in ([world], [-1]))

    if ONE_DEVICE:
        result["transport"] = "host shared memory, all ranks on ONE device (MRCAL_AMD_BENCH_ONE_DEVICE=1): a check of the multi-rank path, NOT a measurement"
    if rank == 0 and not args.no_full_solve and not sharded:
        # the second half of the metric: one full solve, seed to return, outlier rejection included - the drop-in
        # optimize() (round 6), the resident solve beside it, and both again in the product's no-stream mode
        from mrcal_amd.resident import Problem
        from mrcal_amd.synthetic import copy_inputs
        result.update(full_solves(oi, Problem, copy_inputs, torch))
        result["optimizer_callback_ms"] = callback_times(oi, copy_inputs, torch)
        # the metric's steps again, in the product mode that leaves the Jacobian stream out: separately named, NOT
        # the metric (a step of the metric evaluates x AND J)
        with Problem(**copy_inputs(oi)) as p3:
            p3.set_jacobian_stream(False)
            tr3 = None
            if args.warmup > 0: _, tr3 = p3.run_steps(args.warmup, tr3)
            p3.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            n3, tr3 = p3.run_steps(args.steps, tr3)
            p3.synchronize(); torch.cuda.synchronize()
            result["ms_per_step_no_jacobian_stream"] = 1e3*(time.perf_counter() - t0)/n3
            result["no_jacobian_stream_note"] = "product mode (round 6): the solve's steps do not stream the CSR values of J, which nothing " \
                                                "in optimize() reads; identical bits in b_packed / x / outliers. NOT the metric: value, " \
                                                "ms_per_step, roofline and full_solve are measured with the stream on"

    if rank == 0 and not sharded and not args.no_configs:
        result["configs"] = other_configurations()
    if rank == 0:
        cb = None
        if not sharded and not args.no_cpu_baseline:
            cb = cpu_baseline(oi, args.cpu_baseline_iterations)
        result["cpu_baseline"] = cb

    problem.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE line, and the last thing on stdout (RCCL prints its banner at teardown)
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    if sharded:
        # RCCL writes a version banner to stdout when the process exits: keep the JSON line the last one
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)


if __name__ == "__main__":
    main()
