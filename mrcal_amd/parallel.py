"""Multi-GPU solve: one process per GPU, board observations sharded by FRAME.

Why frames: every measurement row of a chessboard calibration touches exactly
one frame pose, so a contiguous range of frames owns its rows of x and J, its
6x6 diagonal blocks of JtJ and its couplings to the camera block outright. What
all shards share is the small dense camera block (all intrinsics + extrinsics +
the board warp: Nc = 140 variables at 8 cameras). Per dog-leg step each rank

  seg 0  eliminates its frames locally (when the trust region asks for the
         Gauss-Newton step) -> all-reduce of its summand of the Schur
         complement [S | r]                   (Nc^2+Nc doubles: 158 KB at NS)
  seg 1  factors the same Nc x Nc system redundantly, back-substitutes its own
         frames -> all-reduce of the frame steps           (NE doubles: 48 KB)
  seg 2  chooses the dog-leg step (the expected improvement comes from dot
         products of replicated vectors: nothing to sum), evaluates x, J and its
         blocks of JtJ at the trial point for ITS frames
         -> all-reduce of [Jt x | |x|^2]       (Nstate+2 doubles: 49 KB at NS)
  seg 3  -> all-reduce of g^T JtJ g                                 (1 double)
  seg 4  Cauchy step of the new point, rho test, accept/reject

i.e. four small collectives, all latency-bound (xGMI bandwidth is irrelevant
at these sizes); the reference has no counterpart (it is single-threaded).

NOTHING is read back between the segments: the trust-region state of
libdogleg is a control block in device memory (csrc/solver_kernels.hip,
"dog-leg control"), REPLICATED on every rank; every rank runs the same control
kernels on the same all-reduced sums and so takes the same decisions. The
collectives are therefore unconditional: a trial that needs no factorization
(or a voided one) sums zeros. The host queues trial steps, and looks at a
pinned snapshot of the control block a few steps behind to learn that the
device has declared the solve finished; all ranks look at the same snapshot
index, so they queue the same number of steps and the collectives match up.

The driver is written against a small "shard" interface so that the whole N>1
flow (partition, segment/collective sequence, termination, outlier rejection)
also runs on CPU under gloo with a numpy shard (tests/test_parallel_cpu.py).
On the GPU the shard is GpuShard: the sharded-step API of libmrcal_amd.so
(include/mrcal_amd.h), with torch only aliasing its HBM buffers for RCCL.
"""
import ctypes as C
import math
import numpy as np

NSEGMENTS = 5       # segments of a trial step; a collective follows each of the first 4
RING      = 8       # control-block snapshots in flight
LAG       = 3       # how many trial steps the host may run ahead of what it has seen


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


class Communicator:
    """sum/max all-reduce over torch.distributed; no-ops for a single process
    (unless always=True: a world of one still goes through the backend, which is
    how the RCCL plumbing is exercised on a one-GPU box)"""
    def __init__(self, group=None, always=False):
        import torch.distributed as dist
        self.dist  = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank  = self.dist.get_rank(group)       if self.dist else 0
        self.active = self.dist is not None and (self.world > 1 or always)
        self.Ncollectives = 0
    def sum(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.Ncollectives += 1
    def max(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
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
    """The dog-leg loop over a shard + a communicator.

    The shard provides (see GpuShard):
      Nmeas_global, Ncorners_global, do_outlier_rejection
      reset(check_termination, max_iterations, trustregion0)   restart the device-side dog-leg
      enqueue(initial, segment)      queue segment 0..4 of a trial step (initial: of the
                                     evaluation of the starting point, segments 2..4)
      comm_buffer(segment)           tensor to sum over the shards after segment 0..3 (or None)
      snapshot(slot) / wait(slot)    control-block snapshot; wait -> dict(done, error, ...)
      finish()                       drain; -> dict(Nsteps_accepted, Nevaluations,
                                     Nfactorizations, Ntrials, error, trustregion, norm2_x, lambda_)
      outlier_stats(thresh_sq) -> tensor [n_outliers, n_beyond, sum_x2] (local, current point)
      mark_outliers(thresh_sq) -> tensor [n_marked] (local)
      context()  -> context manager making the shard's stream current
    """
    def __init__(self, shard, comm, parameters=None):
        self.s     = shard
        self.comm  = comm
        self.prm   = parameters or DoglegParameters()
        self.started = False
        self.stats = dict(Niterations=0, Nevaluations=0, Nfactorizations=0, Noutlier_passes=0)

    def _queue(self, initial):
        s = self.s
        for seg in range(2 if initial else 0, NSEGMENTS):
            s.enqueue(initial, seg)
            if seg < NSEGMENTS-1:
                buf = s.comm_buffer(seg)
                if buf is not None and buf.numel() > 0:
                    self.comm.sum(buf)

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
        rms = math.sqrt(self.stats["norm2_x"]/self.s.Nmeas_global)
        return dict(self.stats, rms_reproj_error__pixels=rms, Noutliers_board=Noutliers)


class _DeviceArray:
    """exposes raw device memory to torch (CUDA array interface)"""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr,
                                             data=(int(ptr), False), version=2)


class GpuShard:
    """The shard interface over libmrcal_amd.so's sharded-step API"""
    def __init__(self, problem, Nmeas_global, Ncorners_global, do_outlier_rejection):
        import torch
        self.torch = torch
        self.p     = problem
        L = problem._lib
        vp = C.c_void_p
        if not getattr(L, "_mrcal_amd_sharded_declared", False):
            L.mrcal_amd_problem_shard_info.restype  = None
            L.mrcal_amd_problem_shard_info.argtypes = [vp, C.POINTER(C.c_int)]
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
            L._mrcal_amd_sharded_declared = True
        self.L = L
        info = (C.c_int*8)()
        L.mrcal_amd_problem_shard_info(problem.handle, info)
        self.Nstate, self.Nie, self.NE, self.Nc = info[0], info[1], info[2], info[3]
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
    def comm_buffer(self, segment):
        if segment not in self._comm:
            n = C.c_int64(0)
            ptr = self.L.mrcal_amd_problem_sharded_comm_buffer(self.p.handle, segment, C.byref(n))
            if not ptr:
                raise RuntimeError("mrcal_amd_problem_sharded_comm_buffer() failed" + self.p._api._last_error())
            self._comm[segment] = (self.torch.as_tensor(_DeviceArray(ptr, n.value, "<f8"), device="cuda")
                                   if n.value > 0 else None)
        return self._comm[segment]
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


class ShardedProblem:
    """A calibration problem sharded by frame over the ranks of the default
    torch.distributed process group (backend nccl = RCCL). Every rank passes
    the SAME optimization_inputs. API-compatible with resident.Problem where
    the benchmark needs it"""
    def __init__(self, group=None, _always_communicate=False, **optimization_inputs):
        from . import _api
        from .resident import Problem
        self.comm = Communicator(group, always=_always_communicate)
        p = _api._ingest(optimization_inputs, callback=False)
        if len(p.c_tri) > 0:
            # the outlier logic of triangulated pairs is sequential over the pairs
            # (mrcal.c:3978-4402) and lives with the single-GPU solve
            raise NotImplementedError("ShardedProblem: triangulated points are solved on one GPU (mrcal_amd.optimize())")
        ranges = partition_frames(p.c_board["iframe"].reshape(-1,1), p.Nframes, self.comm.world)
        self.frame_range = ranges[self.comm.rank]
        self.problem = Problem(_shard=self.frame_range, _leader=(self.comm.rank == 0),
                               **optimization_inputs)
        self.Nstate_global, self.Nmeas_global = _api._sizes(p)
        self.Nstate = self.Nstate_global
        self.Nnz_global = self.problem.Nnz
        if self.comm.active:
            import torch
            t = torch.tensor([float(self.problem.Nnz)], dtype=torch.float64, device="cuda")
            self.comm.sum(t)
            self.Nnz_global = int(t.item())
        Ncorners = p.Nobservations_board*max(p.width_n,0)*max(p.height_n,0)
        self.shard = GpuShard(self.problem, self.Nmeas_global, Ncorners,
                              bool(p.sel.as_dict()["do_apply_outlier_rejection"]))
        self.dogleg = ShardedDogleg(self.shard, self.comm)

    def run_steps(self, Nsteps, trustregion=None):
        return self.dogleg.run(max_steps=Nsteps, check_termination=False, trustregion=trustregion)
    def solve(self):
        return self.dogleg.solve()
    def solver_stats(self):
        return dict(self.dogleg.stats, Ncollectives=self.comm.Ncollectives)
    def b_packed(self):
        return self.problem.b_packed()
    def synchronize(self):
        self.problem.synchronize()
    def jacobian_timing_begin(self, capacity):
        self.problem.jacobian_timing_begin(capacity)
    def jacobian_timing_end(self):
        return self.problem.jacobian_timing_end()
    def jacobian_algorithmic_bytes(self):
        return self.problem.jacobian_algorithmic_bytes()
    def close(self):
        self.problem.close()
