"""Multi-GPU solve: one process per GPU, board observations sharded by FRAME, discrete points by POINT,
triangulated points by point SET.

Why frames: every measurement row of a chessboard calibration touches exactly
one frame pose, so a contiguous range of frames owns its rows of x and J, its
6x6 diagonal blocks of JtJ and its couplings to the camera block outright. What
all shards share is the small dense camera block (all intrinsics + extrinsics +
the board warp: Nc = 140 variables at 8 cameras).

The sharded trial step IS the single-GPU device-controlled step
(csrc/solver.cpp enqueue_trial_step) with TWO sums over the ranks in it:

  [choose the step | evaluate x, J, Grams at the trial point for MY frames |
   block normal equations, my frames eliminated on the spot | my summand of the
   Schur complement]
        -> all-reduce  [ S | r | g_S | |x|^2 | status ]     Nc^2 + 2 Nc + 2 doubles (158 KB at NS)
  [accept/reject, trust region (replicated: every rank decides the same from the
   same sums) | Cholesky of S (replicated) | back-substitution of MY frames |
   my part of g^T JtJ g]
        -> all-reduce  [ g^T JtJ g | |g_E|^2 | |gn_E|^2 | gn_E . g_E ]              4 doubles

Rank-local: the frame poses of the shard's frames (state, gradient, steps) and
its rows of x and J. Replicated: the camera block of the state and the
trust-region control block. Both collectives are latency-bound (xGMI bandwidth
is irrelevant at these sizes); the reference has no counterpart (it is
single-threaded).

Product path (ShardedProblem): the collectives are RCCL all-reduces issued from
C++ on the problem's HIP stream (csrc/comm.cpp); Python only carries the
128-byte communicator id between the ranks at start-up and calls
solve()/run_steps() like a single-GPU caller.

Protocol reference (ShardedDogleg + a "shard" object): the same two-collective
protocol driven from Python through torch.distributed, so that the whole N>1
flow (partition, what is summed, termination, outlier rejection, the final
gather) runs on CPU under gloo with a numpy shard (tests/test_parallel_cpu.py)
and on one GPU with two ranks sharing the device (tests/test_parallel_gpu.py:
RCCL refuses two ranks per device, gloo does not).
"""
import ctypes as C
import math
import numpy as np

RING = 8       # control-block snapshots in flight
LAG  = 3       # how many trial steps the host may run ahead of what it has seen


def partition_frames(indices_frame_camintrinsics_camextrinsics, Nframes, world):
    """Contiguous frame ranges [(f0,f1)]*world with ~equal board-observation
    counts. Every frame lands in exactly one range; ranges may be empty if
    world > Nframes"""
    counts = np.bincount(np.asarray(indices_frame_camintrinsics_camextrinsics)[:,0],
                         minlength=Nframes).astype(np.int64) if Nframes > 0 else np.zeros((0,), np.int64)
    total  = int(counts.sum())
    csum   = np.concatenate(((0,), np.cumsum(counts)))
    bounds = [0]
    for r in range(1, world):
        target = total*r/world
        f = int(np.searchsorted(csum, target, side="left"))
        f = min(max(f, bounds[-1]), Nframes)
        bounds.append(f)
    bounds.append(Nframes)
    return [(bounds[r], bounds[r+1]) for r in range(world)]


def partition_counts(counts, world):
    """Contiguous index ranges [(i0,i1)]*world over len(counts) items with ~equal sums of counts: every item in
    exactly one range; ranges may be empty"""
    counts = np.asarray(counts, dtype=np.int64)
    N      = len(counts)
    total  = int(counts.sum())
    csum   = np.concatenate(((0,), np.cumsum(counts)))
    bounds = [0]
    for r in range(1, world):
        i = int(np.searchsorted(csum, total*r/world, side="left"))
        bounds.append(min(max(i, bounds[-1]), N))
    bounds.append(N)
    return [(bounds[r], bounds[r+1]) for r in range(world)]


def partition_points(indices_point_camintrinsics_camextrinsics, Npoints, world):
    """Discrete points sharded BY POINT: contiguous ranges of i_point with ~equal observation
    counts. A point's 3x3 block of JtJ and all of its observations (in whatever order the caller listed them) go
    to one rank"""
    idx = np.asarray(indices_point_camintrinsics_camextrinsics).reshape(-1, 3)
    counts = np.bincount(idx[:,0], minlength=Npoints) if Npoints > 0 else np.zeros((0,), np.int64)
    return partition_counts(counts, world)


def partition_triangulated(last_in_set, world):
    """Triangulated points sharded by point SET (the consecutive observations of one point, closed by
    last_in_set): contiguous ranges of set indices with ~equal numbers of PAIRS (a set of n observations is
    n(n-1)/2 measurement rows)"""
    last = np.asarray(last_in_set, dtype=bool)
    if last.size == 0:
        return [(0, 0)]*world
    ends = np.nonzero(last)[0]
    if ends.size == 0 or ends[-1] != last.size - 1:
        ends = np.concatenate((ends, (last.size - 1,)))
    n = np.diff(np.concatenate(((-1,), ends)))
    return partition_counts(n*(n-1)//2, world)


# ---------------------------------------------------------------------------
# the protocol reference

class Communicator:
    """sum all-reduce over torch.distributed; a no-op for a single process"""
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist  = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank  = self.dist.get_rank(group)       if self.dist else 0
        self.Ncollectives = 0
    def sum(self, t):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.Ncollectives += 1


class DoglegParameters:
    """mrcal's settings of libdogleg (mrcal.c:6296-6299) over its defaults"""
    max_iterations                 = 300
    trustregion0                   = 1.0e3
    trustregion_decrease_factor    = 0.1
    trustregion_decrease_threshold = 0.25
    trustregion_increase_factor    = 2.0
    trustregion_increase_threshold = 0.75
    update_threshold               = 1e-7
    trustregion_threshold          = 0.0


class ShardedDogleg:
    """The dog-leg loop over a shard + a communicator, collectives from Python.

    The shard provides (see GpuShard):
      Nmeas_global, Ncorners_global, do_outlier_rejection
      reset(check_termination, max_iterations, trustregion0)   restart the device-side dog-leg
      enqueue(initial, segment)      queue segment 0 / 1 of a trial step (initial: of the
                                     evaluation of the starting point)
      comm_buffer(which)             tensor to sum over the shards after segment 0 / 1
      snapshot(slot) / wait(slot)    control-block snapshot; wait -> dict(done, error, ...)
      finish()                       drain; -> dict(Nsteps_accepted, Nevaluations,
                                     Nfactorizations, Ntrials, error, trustregion, norm2_x, lambda_)
      outlier_stats(thresh_sq) -> tensor [n_outliers, n_beyond, sum_x2] (local, current point)
      mark_outliers(thresh_sq) -> tensor [n_marked] (local)
      masked_state() -> tensor: the state with what this shard does not own zeroed;
      set_state(tensor)
      context()  -> context manager making the shard's stream current
    """
    def __init__(self, shard, comm, parameters=None):
        self.s     = shard
        self.comm  = comm
        self.prm   = parameters or DoglegParameters()
        self.started = False
        self.stats = dict(Niterations=0, Nevaluations=0, Nfactorizations=0, Noutlier_passes=0)
        self.Ntrials_total = 0

    def _queue(self, initial):
        s = self.s
        for seg in (0, 1):
            s.enqueue(initial, seg)
            self.comm.sum(s.comm_buffer(seg))

    def run(self, max_steps=None, check_termination=True, trustregion=None):
        """Queues trial steps until the device declares the solve finished
        (check_termination) or max_steps of them have been queued. Continues
        from where the previous call stopped if a trust region is passed in.
        Returns (trial steps queued, trust region)"""
        s = self.s
        with s.context():
            if not (self.started and trustregion is not None and trustregion > 0):
                s.reset(check_termination, self.prm.max_iterations,
                        trustregion if (trustregion is not None and trustregion > 0) else self.prm.trustregion0)
                self._queue(True)
                self.started = True
            n, done = 0, False
            guard = 100*self.prm.max_iterations + 1000
            while not done and n < guard:
                if max_steps is not None and n >= max_steps:
                    break
                self._queue(False)
                s.snapshot(n % RING)
                n += 1
                if check_termination and n >= LAG:
                    # every rank waits for the SAME snapshot: same decision everywhere
                    done = bool(s.wait((n - LAG) % RING)["done"])
            c = s.finish()
        self.Ntrials_total += n + 1
        if c["error"]:
            raise RuntimeError("could not make JtJ positive definite")
        if check_termination:
            self.stats["Niterations"]     += c["Nsteps_accepted"]
            self.stats["Nevaluations"]    += c["Nevaluations"]
            self.stats["Nfactorizations"] += c["Nfactorizations"]
        else:
            self.stats["Niterations"]     = c["Nsteps_accepted"]
            self.stats["Nevaluations"]    = c["Nevaluations"]
            self.stats["Nfactorizations"] = c["Nfactorizations"]
        self.stats["norm2_x"] = c["norm2_x"]
        self.stats["lambda_"] = c["lambda_"]
        return n, c["trustregion"]

    def mark_outliers(self):
        """mrcal.c:3978-4402, boards only, with global statistics. Returns
        (found_new, Noutliers_total)"""
        s, k0, k1 = self.s, 4.0, 5.0
        with s.context():
            st = s.outlier_stats(-1.0)
            self.comm.sum(st)
            nout, _, sumx2 = st.tolist()
            ninl = s.Ncorners_global - int(nout)
            if ninl <= 0:
                return False, int(nout)
            var = sumx2/(2.0*ninl)
            st = s.outlier_stats(k1*k1*var)
            self.comm.sum(st)
            if int(st[1].item()) == 0:
                return False, int(nout)
            nm = s.mark_outliers(k0*k0*var)
            self.comm.sum(nm)
            return True, int(nout) + int(nm[0].item())

    def gather_state(self):
        with self.s.context():
            b = self.s.masked_state()
            self.comm.sum(b)
            self.s.set_state(b)

    def solve(self):
        Noutliers = 0
        while True:
            self.started = False
            self.run()
            if not self.s.do_outlier_rejection:
                # the outliers given on input are still reported (mrcal.c:6418-6421)
                with self.s.context():
                    st = self.s.outlier_stats(-1.0)
                    self.comm.sum(st)
                Noutliers = int(st[0].item())
                break
            found, Noutliers = self.mark_outliers()
            if not found:
                break
            self.stats["Noutlier_passes"] += 1
        self.started = False
        self.gather_state()
        rms = math.sqrt(self.stats["norm2_x"]/self.s.Nmeas_global)
        return dict(self.stats, rms_reproj_error__pixels=rms, Noutliers_board=Noutliers)


class _DeviceArray:
    """exposes raw device memory to torch (CUDA array interface)"""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr,
                                             data=(int(ptr), False), version=2)


def _declare_sharded(L):
    if getattr(L, "_mrcal_amd_sharded_declared", False):
        return
    vp = C.c_void_p
    L.mrcal_amd_problem_shard_info.restype  = C.c_int
    L.mrcal_amd_problem_shard_info.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
    L.mrcal_amd_problem_sharded_reset.restype  = C.c_bool
    L.mrcal_amd_problem_sharded_reset.argtypes = [vp, C.c_int, C.c_int, C.c_double]
    L.mrcal_amd_problem_sharded_enqueue.restype  = C.c_bool
    L.mrcal_amd_problem_sharded_enqueue.argtypes = [vp, C.c_int, C.c_int]
    L.mrcal_amd_problem_sharded_comm_buffer.restype  = vp
    L.mrcal_amd_problem_sharded_comm_buffer.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.mrcal_amd_problem_sharded_snapshot.restype  = C.c_bool
    L.mrcal_amd_problem_sharded_snapshot.argtypes = [vp, C.c_int]
    L.mrcal_amd_problem_sharded_wait.restype  = C.c_bool
    L.mrcal_amd_problem_sharded_wait.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.mrcal_amd_problem_sharded_finish.restype  = C.c_bool
    L.mrcal_amd_problem_sharded_finish.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mrcal_amd_problem_current.restype, L.mrcal_amd_problem_current.argtypes = C.c_int, [vp]
    for name, args in (("outlier_stats", [vp, C.c_int, C.c_double, vp, vp]),
                       ("mark_outliers", [vp, C.c_int, C.c_double, vp])):
        f = getattr(L, f"mrcal_amd_problem_phase_{name}")
        f.restype, f.argtypes = C.c_bool, args
    L.mrcal_amd_comm_unique_id.restype,  L.mrcal_amd_comm_unique_id.argtypes  = C.c_bool, [vp]
    L.mrcal_amd_comm_create.restype,     L.mrcal_amd_comm_create.argtypes     = vp, [vp, C.c_int, C.c_int]
    L.mrcal_amd_comm_create_host.restype, L.mrcal_amd_comm_create_host.argtypes = vp, [C.c_char_p, C.c_int, C.c_int]
    L.mrcal_amd_comm_destroy.restype,    L.mrcal_amd_comm_destroy.argtypes    = None, [vp]
    L.mrcal_amd_comm_Ncollectives.restype, L.mrcal_amd_comm_Ncollectives.argtypes = C.c_long, [vp]
    L.mrcal_amd_comm_Ndoubles.restype, L.mrcal_amd_comm_Ndoubles.argtypes = C.c_longlong, [vp]
    L.mrcal_amd_comm_world_observed.restype, L.mrcal_amd_comm_world_observed.argtypes = C.c_int, [vp]
    L.mrcal_amd_problem_attach_comm.restype,  L.mrcal_amd_problem_attach_comm.argtypes  = C.c_bool, [vp, vp]
    L.mrcal_amd_problem_gather_state.restype, L.mrcal_amd_problem_gather_state.argtypes = C.c_bool, [vp]
    L._mrcal_amd_sharded_declared = True


class GpuShard:
    """The shard interface over libmrcal_amd.so's sharded-step API, for a driver
    that does the collectives itself (ShardedDogleg)"""
    def __init__(self, problem, Nmeas_global, Ncorners_global, do_outlier_rejection):
        import torch
        self.torch = torch
        self.p     = problem
        self.L = L = problem._lib
        _declare_sharded(L)
        info = (C.c_int*12)()
        # (called unconditionally: under `python -O` an assert - and the call inside it - is not there)
        nfilled = L.mrcal_amd_problem_shard_info(problem.handle, info, len(info))
        if nfilled < len(info):
            raise RuntimeError(f"mrcal_amd_problem_shard_info() filled {nfilled} of {len(info)} fields: "
                               "libmrcal_amd.so and mrcal_amd/parallel.py are not of the same build")
        # (Nie: S_split, the end of the leading intrinsics + extrinsics variables of a sharded problem's state)
        self.Nstate, self.Nie, self.NE, self.Nc = info[0], info[1], info[2], info[3]
        self.frame_lo, self.frame_hi = info[4], info[5]
        self.Nfb = info[8]
        self.point_lo, self.point_hi = info[10] - info[8], info[11] - info[8]      # variable points owned
        self.is_leader = bool(info[6])
        self.Nmeas_global = Nmeas_global
        self.Ncorners_global = Ncorners_global
        self.do_outlier_rejection = do_outlier_rejection
        torch.cuda.init()
        self.stream = torch.cuda.ExternalStream(problem.stream(),
                                                device=torch.device("cuda", torch.cuda.current_device()))
        self._comm   = {}
        self._counts = torch.zeros(4, dtype=torch.int32, device="cuda")
        self._sums   = torch.zeros(1, dtype=torch.float64, device="cuda")

    def context(self):
        return self.torch.cuda.stream(self.stream)

    def _ok(self, ok, what):
        if not ok:
            raise RuntimeError(f"{what} failed:" + self.p._api._last_error())

    def reset(self, check_termination, max_iterations, trustregion0):
        self._ok(self.L.mrcal_amd_problem_sharded_reset(self.p.handle, int(bool(check_termination)),
                                                        int(max_iterations), float(trustregion0)), "sharded_reset")
    def enqueue(self, initial, segment):
        self._ok(self.L.mrcal_amd_problem_sharded_enqueue(self.p.handle, int(bool(initial)), segment), "sharded_enqueue")
    def comm_buffer(self, which):
        if which not in self._comm:
            n = C.c_int64(0)
            ptr = self.L.mrcal_amd_problem_sharded_comm_buffer(self.p.handle, which, C.byref(n))
            if not ptr:
                raise RuntimeError("mrcal_amd_problem_sharded_comm_buffer() failed" + self.p._api._last_error())
            self._comm[which] = self.torch.as_tensor(_DeviceArray(ptr, n.value, "<f8"), device="cuda")
        return self._comm[which]
    def snapshot(self, slot):
        self._ok(self.L.mrcal_amd_problem_sharded_snapshot(self.p.handle, slot), "sharded_snapshot")
    def wait(self, slot):
        out = (C.c_int*4)()
        self._ok(self.L.mrcal_amd_problem_sharded_wait(self.p.handle, slot, out), "sharded_wait")
        return dict(done=out[0], error=out[1], Nsteps_accepted=out[2], Ntrials=out[3])
    def finish(self):
        oi, od = (C.c_int*5)(), (C.c_double*3)()
        self._ok(self.L.mrcal_amd_problem_sharded_finish(self.p.handle, oi, od), "sharded_finish")
        return dict(Nsteps_accepted=oi[0], Nevaluations=oi[1], Nfactorizations=oi[2], Ntrials=oi[3],
                    error=oi[4], trustregion=od[0], norm2_x=od[1], lambda_=od[2])

    def outlier_stats(self, thresh_sq):
        iop = self.L.mrcal_amd_problem_current(self.p.handle)
        self._counts.zero_(); self._sums.zero_()
        self._ok(self.L.mrcal_amd_problem_phase_outlier_stats(self.p.handle, iop, float(thresh_sq),
                                                               self._counts.data_ptr(), self._sums.data_ptr()),
                 "phase_outlier_stats")
        return self.torch.cat((self._counts[:2].to(self.torch.float64), self._sums))
    def mark_outliers(self, thresh_sq):
        iop = self.L.mrcal_amd_problem_current(self.p.handle)
        self._counts.zero_()
        self._ok(self.L.mrcal_amd_problem_phase_mark_outliers(self.p.handle, iop, float(thresh_sq),
                                                               self._counts.data_ptr()), "phase_mark_outliers")
        return self._counts[:1].to(self.torch.float64)

    def masked_state(self):
        b = self.p.b_packed()
        mine = np.zeros(self.Nstate, dtype=bool)
        if self.is_leader:
            mine[:self.Nie] = True
            mine[self.Nie + self.NE:] = True             # the warp
        mine[self.Nie + 6*self.frame_lo : self.Nie + 6*self.frame_hi] = True
        mine[self.Nie + 6*self.Nfb + 3*self.point_lo : self.Nie + 6*self.Nfb + 3*self.point_hi] = True
        b[~mine] = 0.0
        return self.torch.from_numpy(b).to("cuda")
    def set_state(self, t):
        self.p.set_b_packed(t.cpu().numpy())


class ShardedProblem:
    """A calibration problem sharded by frame over the ranks of the default
    torch.distributed process group. Every rank passes the SAME
    optimization_inputs. API-compatible with resident.Problem where the
    benchmark needs it.

    The product path: the two collectives per trial step are RCCL all-reduces
    issued from C++ on the problem's stream; torch.distributed (any backend:
    gloo is enough) only carries the 128-byte communicator id at start-up.
    _driver="host": the same C++ solve with its collectives staged through a
    shared-memory segment of the host (csrc/comm.cpp): several ranks on ONE
    device, which RCCL refuses - what the tests of the C++ path at world > 1 run.
    _driver="python": the protocol reference (collectives through
    torch.distributed from Python)"""
    def __init__(self, group=None, _driver="rccl", **optimization_inputs):
        import torch
        import torch.distributed as dist
        from . import _api
        from .resident import Problem
        have_dist = dist.is_available() and dist.is_initialized()
        self.rank  = dist.get_rank(group)       if have_dist else 0
        self.world = dist.get_world_size(group) if have_dist else 1
        p = _api._ingest(optimization_inputs, callback=False)
        self.do_outlier_rejection = bool(p.sel.as_dict()["do_apply_outlier_rejection"])
        if len(p.c_tri) > 0 and _driver == "python" and self.do_outlier_rejection:
            # (the protocol reference's mark_outliers() knows boards only; the product path - the C++ one - does
            #  the pairs' divergence and k-sigma logic per shard, with the variance summed over the ranks)
            raise NotImplementedError("the Python protocol driver rejects board outliers only; use the RCCL driver")
        # boards by frame, discrete points by point, triangulated points by point set: the rows of x and J and the
        # eliminated blocks of each are then local to one rank
        self.frame_range = partition_frames(p.c_board["iframe"].reshape(-1,1), p.Nframes, self.world)[self.rank]
        self.point_range = partition_points(np.column_stack((p.c_point["i_point"],)*3), p.Npoints, self.world)[self.rank] \
                           if len(p.c_point) > 0 else (0, 0)
        self.tripoint_range = partition_triangulated(p.c_tri["flags"] & 1, self.world)[self.rank]
        self.problem = Problem(_shard=self.frame_range, _leader=(self.rank == 0),
                               _shard_points=self.point_range, _shard_tripoints=self.tripoint_range,
                               **optimization_inputs)
        self.Nstate_global, self.Nmeas_global = _api._sizes(p)
        self.Nstate = self.Nstate_global
        self.Nnz_global = self.problem.Nnz
        if have_dist and self.world > 1:
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.tensor([float(self.problem.Nnz)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=group)
            self.Nnz_global = int(t.item())
        self._lib = L = self.problem._lib
        _declare_sharded(L)
        self._comm_handle = None
        self.dogleg = None
        self.Ncorners = p.Nobservations_board*max(p.width_n,0)*max(p.height_n,0)
        if _driver == "python":
            self.comm  = Communicator(group)
            self.shard = GpuShard(self.problem, self.Nmeas_global, self.Ncorners, self.do_outlier_rejection)
            self.dogleg = ShardedDogleg(self.shard, self.comm)
            return
        if _driver == "host":
            import os, secrets
            name = [("/mrcal_amd_%d_%s" % (os.getpid(), secrets.token_hex(4))) if self.rank == 0 else None]
            if have_dist and self.world > 1:
                dist.broadcast_object_list(name, src=0, group=group)
            self._comm_handle = L.mrcal_amd_comm_create_host(name[0].encode(), self.rank, self.world)
            if not self._comm_handle:
                raise RuntimeError("mrcal_amd_comm_create_host() failed:" + _api._last_error())
            if not L.mrcal_amd_problem_attach_comm(self.problem.handle, self._comm_handle):
                raise RuntimeError("mrcal_amd_problem_attach_comm() failed:" + _api._last_error())
            return
        if _driver != "rccl":
            raise ValueError("_driver: 'rccl', 'host' or 'python'")
        # the communicator id, made by rank 0, to everybody
        idbuf = (C.c_char*128)()
        if self.rank == 0:
            if not L.mrcal_amd_comm_unique_id(idbuf):
                raise RuntimeError("mrcal_amd_comm_unique_id() failed:" + _api._last_error())
        if have_dist and self.world > 1:
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.tensor(list(idbuf.raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0, group=group)
            idbuf = (C.c_char*128).from_buffer_copy(bytes(t.cpu().tolist()))
        self._comm_handle = L.mrcal_amd_comm_create(idbuf, self.rank, self.world)
        if not self._comm_handle:
            raise RuntimeError("mrcal_amd_comm_create() failed:" + _api._last_error())
        if not L.mrcal_amd_problem_attach_comm(self.problem.handle, self._comm_handle):
            raise RuntimeError("mrcal_amd_problem_attach_comm() failed:" + _api._last_error())

    @property
    def Ncollectives(self):
        if self.dogleg is not None:
            return self.comm.Ncollectives
        return int(self._lib.mrcal_amd_comm_Ncollectives(self._comm_handle))

    def comm_info(self):
        """dict(world_observed, Ncollectives, bytes) of this rank's communicator: the transport's own rank count
        (ncclCommCount()), how many all-reduces it has queued and how many bytes they summed"""
        if self.dogleg is not None:
            return dict(world_observed=self.comm.world, Ncollectives=self.comm.Ncollectives, bytes=None)
        return dict(world_observed=int(self._lib.mrcal_amd_comm_world_observed(self._comm_handle)),
                    Ncollectives=int(self._lib.mrcal_amd_comm_Ncollectives(self._comm_handle)),
                    bytes=8*int(self._lib.mrcal_amd_comm_Ndoubles(self._comm_handle)))

    def run_steps(self, Nsteps, trustregion=None):
        if self.dogleg is not None:
            return self.dogleg.run(max_steps=Nsteps, check_termination=False, trustregion=trustregion)
        return self.problem.run_steps(Nsteps, trustregion)
    def solve(self):
        if self.dogleg is not None:
            return self.dogleg.solve()
        st = self.problem.solve()
        # (rms and counters are those of the whole problem: they come out of the replicated control block)
        st["rms_reproj_error__pixels"] = math.sqrt(st["norm2_x"]/self.Nmeas_global)
        return st
    def solver_stats(self):
        if self.dogleg is not None:
            return dict(self.dogleg.stats, Ncollectives=self.Ncollectives)
        return dict(self.problem.solver_stats(), Ncollectives=self.Ncollectives)
    def b_packed(self):
        return self.problem.b_packed()
    def synchronize(self):
        self.problem.synchronize()
    def jacobian_timing_begin(self, capacity, stride=1):
        self.problem.jacobian_timing_begin(capacity, stride)
    def jacobian_timing_end(self):
        return self.problem.jacobian_timing_end()
    def jacobian_algorithmic_bytes(self):
        return self.problem.jacobian_algorithmic_bytes()
    def close(self):
        self.problem.close()
        if self._comm_handle:
            self._lib.mrcal_amd_comm_destroy(self._comm_handle)
            self._comm_handle = None
