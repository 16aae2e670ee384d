"""Multi-GPU solve: one process per GPU, board observations sharded by FRAME.

Why frames: every measurement row of a chessboard calibration touches exactly
one frame pose, so a contiguous range of frames owns its rows of x and J, its
6x6 diagonal blocks of JtJ and its couplings to the camera block outright. What
all shards share is the small dense camera block (all intrinsics + extrinsics +
the board warp: Nc = 140 variables at 8 cameras). Per dog-leg step each rank

  1. evaluates x, J and its blocks of JtJ for ITS frames           (HIP kernels)
  2. all-reduces [Jt x | |x|^2]                  (Nstate+1 doubles: 49 KB at NS)
  3. all-reduces g^T JtJ g                                          (1 double)
  4. when the Gauss-Newton step is needed: eliminates its frames locally and
     all-reduces its summand of the Schur complement  [S | r]
                                           (Nc^2+Nc doubles: 158 KB at NS)
     then every rank factors the same Nc x Nc system redundantly and
     back-substitutes its own frames; the frame steps are all-reduced
                                                            (NE doubles: 48 KB)
  5. all-reduces s^T JtJ s for the accept/reject test                (1 double)

i.e. five small collectives, all latency-bound (xGMI bandwidth is irrelevant
at these sizes); the reference has no counterpart (it is single-threaded).

The trust-region logic here is the same libdogleg algorithm as in
csrc/solver.cpp (the single-GPU path), written against a small "shard"
interface so that the whole N>1 control flow also runs on CPU under gloo with a
numpy shard (tests/test_parallel_cpu.py). On the GPU the shard is GpuShard: the
phase API of libmrcal_amd.so, with torch only aliasing its HBM buffers for RCCL.
"""
import ctypes as C
import math
import numpy as np

# indices into the per-operating-point scalars buffer (csrc/solver_kernels.hpp)
SC_NORM2_X, SC_NORM2_G, SC_GNG, SC_TMP0, SC_TMP1, SC_TMP2, SC_TMP3 = range(7)


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
    """sum/max all-reduce over torch.distributed; no-ops for a single process"""
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist  = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank  = self.dist.get_rank(group)       if self.dist else 0
        self.Ncollectives = 0
    def sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.Ncollectives += 1
    def max(self, t):
        if self.world > 1:
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
      Nstate, Nie, NE, Nc, Nmeas_global, do_outlier_rejection
      vec(name, iop) -> torch tensor aliasing a solver buffer of operating point iop
            names: b, g, step_cauchy, step_gn, scalars ; 'step','schur','status' (iop ignored)
      evaluate(iop); quadform(iop, v, out); factor_local(iop, lam); solve_backsub(iop)
      outlier_stats(iop, thresh_sq) -> tensor [n_outliers, n_beyond, sum_x2] (local)
      mark_outliers(iop, thresh_sq) -> tensor [n_marked] (local)
      context()  -> context manager making the shard's stream current
    """
    def __init__(self, shard, comm, parameters=None):
        import torch
        self.torch = torch
        self.s     = shard
        self.comm  = comm
        self.prm   = parameters or DoglegParameters()
        self.lam   = 0.0
        self.ib    = 0
        self.host  = [dict(), dict()]    # host mirrors per operating point
        self.stage = shard.new_buffer(shard.Nstate + 8)
        self.stats = dict(Niterations=0, Nevaluations=0, Nfactorizations=0, Noutlier_passes=0)

    # -- one operating point ------------------------------------------------
    def evaluate(self, i):
        s, t = self.s, self.torch
        with s.context():
            s.evaluate(i)
            g, sc = s.vec("g", i), s.vec("scalars", i)
            n = s.Nstate
            st = self.stage
            st[:n].copy_(g); st[n:n+1].copy_(sc[SC_NORM2_X:SC_NORM2_X+1])
            self.comm.sum(st[:n+1])
            g.copy_(st[:n]); sc[SC_NORM2_X:SC_NORM2_X+1].copy_(st[n:n+1])
            sc[SC_GNG] = 0.0
            s.quadform(i, g, sc[SC_GNG:SC_GNG+1])
            self.comm.sum(sc[SC_GNG:SC_GNG+1])
            sc[SC_NORM2_G] = t.dot(g, g)
            norm2_x, norm2_g, gNg = sc[:3].tolist()       # the one host sync
            k = -norm2_g/gNg if gNg > 0.0 else 0.0
            t.mul(g, k, out=s.vec("step_cauchy", i))
        self.host[i] = dict(norm2_x=norm2_x, cauchy_lensq=k*k*norm2_g, gn_valid=False,
                            did_step_to_edge=False)
        self.stats["Nevaluations"] += 1

    def gauss_newton(self, i):
        s, t, h = self.s, self.torch, self.host[i]
        if h["gn_valid"]:
            return
        while True:
            with s.context():
                s.factor_local(i, self.lam)
                self.comm.sum(s.vec("schur", i))
                gn = s.vec("step_gn", i)
                s.solve_backsub(i)
                self.comm.sum(gn[s.Nie:s.Nie+s.NE])
                status = s.vec("status", i)
                self.comm.max(status)
                lensq = float(t.dot(gn, gn).item())
                bad   = int(status.item()) != 0 or not math.isfinite(lensq)
            self.stats["Nfactorizations"] += 1
            if not bad:
                h["gn_lensq"] = lensq
                h["gn_valid"] = True
                return
            self.lam = 1e-10 if self.lam == 0.0 else self.lam*10.0
            if not self.lam < 1e30:
                raise RuntimeError("could not make JtJ positive definite")

    def take_step(self, ib, ia, trustregion):
        s, t, h = self.s, self.torch, self.host[ib]
        with s.context():
            step   = s.vec("step", 0)
            cauchy = s.vec("step_cauchy", ib)
            if h["cauchy_lensq"] >= trustregion*trustregion:
                t.mul(cauchy, trustregion/math.sqrt(h["cauchy_lensq"]), out=step)
                step_len_sq = trustregion*trustregion
                h["did_step_to_edge"] = True
            else:
                self.gauss_newton(ib)
                gn = s.vec("step_gn", ib)
                if h["gn_lensq"] <= trustregion*trustregion:
                    step.copy_(gn)
                    step_len_sq = h["gn_lensq"]
                    h["did_step_to_edge"] = False
                else:
                    ab     = float(t.dot(cauchy, gn).item())
                    dsq    = trustregion*trustregion
                    norm2a, norm2b = h["cauchy_lensq"], h["gn_lensq"]
                    l2     = norm2a - 2.0*ab + norm2b
                    neg_c  = norm2a - ab
                    disc   = max(neg_c*neg_c - l2*(norm2a - dsq), 0.0)
                    k      = (neg_c + math.sqrt(disc))/l2
                    t.add(cauchy*(1.0-k), gn, alpha=k, out=step)
                    step_len_sq = (1.0-k)*(1.0-k)*norm2a + 2.0*k*(1.0-k)*ab + k*k*norm2b
                    h["did_step_to_edge"] = True
            t.add(s.vec("b", ib), step, out=s.vec("b", ia))
            # expected improvement -2 g.s - s^T JtJ s
            sc = s.vec("scalars", ib)
            sc[SC_TMP2] = 0.0
            s.quadform(ib, step, sc[SC_TMP2:SC_TMP2+1])
            self.comm.sum(sc[SC_TMP2:SC_TMP2+1])
            gs  = float(t.dot(s.vec("g", ib), step).item())
            sNs = float(sc[SC_TMP2].item())
        return step_len_sq, -2.0*gs - sNs

    # -- libdogleg's loop ----------------------------------------------------
    def run(self, max_steps=None, check_termination=True, trustregion=None):
        """max_steps: stop after this many trial steps (accepted or not).
        Returns (trial steps taken, trust region)"""
        prm = self.prm
        ib, ia = self.ib, 1 - self.ib
        if trustregion is None or trustregion <= 0:
            trustregion = prm.trustregion0
        if not self.host[ib]:
            self.evaluate(ib)
        Ntrials, Naccepted = 0, 0
        done = False
        while not done:
            if max_steps is not None and Ntrials >= max_steps:
                break
            if check_termination and Naccepted >= prm.max_iterations:
                break
            step_len_sq, expected = self.take_step(ib, ia, trustregion)
            if check_termination and step_len_sq < prm.update_threshold**2:
                break
            self.evaluate(ia)
            Ntrials += 1
            rho = (self.host[ib]["norm2_x"] - self.host[ia]["norm2_x"])/expected
            if rho < prm.trustregion_decrease_threshold:
                trustregion *= prm.trustregion_decrease_factor
            elif rho > prm.trustregion_increase_threshold and self.host[ib]["did_step_to_edge"]:
                trustregion *= prm.trustregion_increase_factor
            if rho > 0.0:
                ib, ia = ia, ib
                Naccepted += 1
            elif check_termination and (trustregion < prm.trustregion_threshold or
                                        trustregion == 0.0 or trustregion != trustregion):
                done = True
        self.ib = ib
        self.s.set_current(ib)
        self.stats["Niterations"] += Naccepted
        self.stats["norm2_x"] = self.host[ib]["norm2_x"]
        return Ntrials, trustregion

    def mark_outliers(self):
        """mrcal.c:3978-4402, boards only, with global statistics. Returns
        (found_new, Noutliers_total)"""
        s, k0, k1 = self.s, 4.0, 5.0
        with s.context():
            st = s.outlier_stats(self.ib, -1.0)
            self.comm.sum(st)
            nout, _, sumx2 = st.tolist()
            ninl = s.Ncorners_global - int(nout)
            if ninl <= 0:
                return False, int(nout)
            var = sumx2/(2.0*ninl)
            st = s.outlier_stats(self.ib, k1*k1*var)
            self.comm.sum(st)
            if int(st[1].item()) == 0:
                return False, int(nout)
            nm = s.mark_outliers(self.ib, k0*k0*var)
            self.comm.sum(nm)
            return True, int(nout) + int(nm[0].item())

    def solve(self):
        Noutliers = 0
        while True:
            self.host = [dict(), dict()]
            self.run()
            if not self.s.do_outlier_rejection:
                break
            found, Noutliers = self.mark_outliers()
            if not found:
                break
            self.stats["Noutlier_passes"] += 1
        rms = math.sqrt(self.stats["norm2_x"]/self.s.Nmeas_global)
        return dict(self.stats, rms_reproj_error__pixels=rms, Noutliers_board=Noutliers,
                    lambda_=self.lam)


class _DeviceArray:
    """exposes raw device memory to torch (CUDA array interface)"""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr,
                                             data=(int(ptr), False), version=2)


class GpuShard:
    """The shard interface over libmrcal_amd.so's phase API"""
    NAMES = dict(b=0, x=1, g=2, step_cauchy=3, step_gn=4, scalars=5, step=6, schur=7, status=8)

    def __init__(self, problem, Nmeas_global, Ncorners_global, do_outlier_rejection):
        import torch
        self.torch = torch
        self.p     = problem
        L = problem._lib
        vp = C.c_void_p
        if not getattr(L, "_mrcal_amd_phase_declared", False):
            L.mrcal_amd_problem_buffer.restype  = vp
            L.mrcal_amd_problem_buffer.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]
            L.mrcal_amd_problem_shard_info.restype  = None
            L.mrcal_amd_problem_shard_info.argtypes = [vp, C.POINTER(C.c_int)]
            for name, args in (("evaluate", [vp, C.c_int]), ("quadform", [vp, C.c_int, vp, vp]),
                               ("factor_local", [vp, C.c_int, C.c_double]), ("solve_backsub", [vp, C.c_int]),
                               ("outlier_stats", [vp, C.c_int, C.c_double, vp, vp]),
                               ("mark_outliers", [vp, C.c_int, C.c_double, vp])):
                f = getattr(L, f"mrcal_amd_problem_phase_{name}")
                f.restype, f.argtypes = C.c_bool, args
            L.mrcal_amd_problem_set_current.restype, L.mrcal_amd_problem_set_current.argtypes = None, [vp, C.c_int]
            L._mrcal_amd_phase_declared = True
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
        self._vec = {}
        self._counts = torch.zeros(4, dtype=torch.int32, device="cuda")
        self._sums   = torch.zeros(1, dtype=torch.float64, device="cuda")

    def context(self):
        return self.torch.cuda.stream(self.stream)

    def new_buffer(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device="cuda")

    def vec(self, name, iop):
        key = (name, iop if name not in ("step", "schur", "status") else 0)
        t = self._vec.get(key)
        if t is None:
            n = C.c_int64(0)
            ptr = self.L.mrcal_amd_problem_buffer(self.p.handle, self.NAMES[name], key[1], C.byref(n))
            if not ptr:
                raise RuntimeError("mrcal_amd_problem_buffer() failed" + self.p._api._last_error())
            typestr = "<i4" if name == "status" else "<f8"
            t = self.torch.as_tensor(_DeviceArray(ptr, n.value, typestr), device="cuda")
            self._vec[key] = t
        return t

    def _ok(self, ok, what):
        if not ok:
            raise RuntimeError(f"{what} failed:" + self.p._api._last_error())

    def evaluate(self, iop):
        self._ok(self.L.mrcal_amd_problem_phase_evaluate(self.p.handle, iop), "phase_evaluate")
    def quadform(self, iop, v, out):
        self._ok(self.L.mrcal_amd_problem_phase_quadform(self.p.handle, iop, v.data_ptr(), out.data_ptr()), "phase_quadform")
    def factor_local(self, iop, lam):
        self._ok(self.L.mrcal_amd_problem_phase_factor_local(self.p.handle, iop, float(lam)), "phase_factor_local")
    def solve_backsub(self, iop):
        self._ok(self.L.mrcal_amd_problem_phase_solve_backsub(self.p.handle, iop), "phase_solve_backsub")
    def set_current(self, iop):
        self.L.mrcal_amd_problem_set_current(self.p.handle, iop)

    def outlier_stats(self, iop, thresh_sq):
        self._counts.zero_(); self._sums.zero_()
        self._ok(self.L.mrcal_amd_problem_phase_outlier_stats(self.p.handle, iop, float(thresh_sq),
                                                               self._counts.data_ptr(), self._sums.data_ptr()),
                 "phase_outlier_stats")
        return self.torch.cat((self._counts[:2].to(self.torch.float64), self._sums))
    def mark_outliers(self, iop, thresh_sq):
        self._counts.zero_()
        self._ok(self.L.mrcal_amd_problem_phase_mark_outliers(self.p.handle, iop, float(thresh_sq),
                                                               self._counts.data_ptr()), "phase_mark_outliers")
        return self._counts[:1].to(self.torch.float64)


class ShardedProblem:
    """A calibration problem sharded by frame over the ranks of the default
    torch.distributed process group (backend nccl = RCCL). Every rank passes
    the SAME optimization_inputs. API-compatible with resident.Problem where
    the benchmark needs it"""
    def __init__(self, group=None, **optimization_inputs):
        from . import _api
        from .resident import Problem
        self.comm = Communicator(group)
        p = _api._ingest(optimization_inputs, callback=False)
        ranges = partition_frames(p.c_board["iframe"].reshape(-1,1), p.Nframes, self.comm.world)
        self.frame_range = ranges[self.comm.rank]
        self.problem = Problem(_shard=self.frame_range, _leader=(self.comm.rank == 0),
                               **optimization_inputs)
        self.Nstate_global, self.Nmeas_global = _api._sizes(p)
        self.Nstate = self.Nstate_global
        self.Nnz_global = self.problem.Nnz
        if self.comm.world > 1:
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
