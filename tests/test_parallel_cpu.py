"""The N>1 flow (mrcal_amd/parallel.py, the protocol reference driver) on CPU:
world_size 2 and 3 over gloo, with a numpy shard standing in for the GPU's
sharded-step kernels: TWO sums over the shards per trial step. The numpy shard
gets its x and J from the CPU checker (the reference's own optimizer_callback,
oracle/_ref), keeps only the rows of ITS frames, and does the block algebra
densely. Checked:

  - partition_frames(): contiguous, complete, balanced
  - the sharded solve lands on the same optimum as the same driver run
    unsharded, and as the reference's mrcal_optimize() on the same problem
  - same outliers, same trial-step count as the unsharded run (the arithmetic
    is identical up to summation order)
"""
import contextlib
import math
import os
import sys
import numpy as np
import pytest

from conftest import ROOT, REFLIB_PATH


def test_partition_frames():
    from mrcal_amd.parallel import partition_frames
    rng = np.random.RandomState(0)
    for Nframes, world in ((10,1), (10,2), (1000,8), (7,8), (3,2), (0,2)):
        ncam = rng.randint(1, 5, size=Nframes)
        idx = np.zeros((int(ncam.sum()),3), dtype=np.int32)
        idx[:,0] = np.repeat(np.arange(Nframes), ncam)
        r = partition_frames(idx, Nframes, world)
        assert len(r) == world
        assert r[0][0] == 0 and r[-1][1] == Nframes
        for a, b in zip(r[:-1], r[1:]):
            assert a[1] == b[0] and a[0] <= a[1]
        if Nframes >= 4*world:
            counts = [ncam[f0:f1].sum() for f0,f1 in r]
            assert max(counts) - min(counts) <= 2*ncam.max()


def test_partitions_of_points_and_pairs():
    from mrcal_amd.parallel import partition_points, partition_triangulated, partition_counts
    idx = np.array(((0,0,-1),(2,1,0),(1,0,-1),(2,0,-1),(3,1,0),(3,0,-1),(3,2,1)), dtype=np.int32)     # unsorted by point
    for world in (1, 2, 3, 8):
        r = partition_points(idx, 5, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == 5
        assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
    last = np.array((0,0,1, 0,1, 0,0,0,1, 0,1), dtype=bool)      # sets of 3, 2, 4, 2 observations: 3, 1, 6, 1 pairs
    assert partition_triangulated(last, 1) == [(0,4)]
    r = partition_triangulated(last, 2)
    assert r[0][0] == 0 and r[-1][1] == 4 and r[0][1] == r[1][0]
    assert partition_triangulated(np.zeros((0,), dtype=bool), 3) == [(0,0)]*3
    assert partition_counts([5,5,5,5], 2) == [(0,2),(2,4)]


class NumpyShard:
    """CPU stand-in for GpuShard (same interface), for ONE rank: the two
    segments of the sharded trial step in numpy, the device-side control logic
    (csrc/step.hip, "the fused step": step2_choose_kernel,
    step2_finish, step2_chol_done) restated in Python. x and J come from the
    CPU checker; the block algebra is dense"""
    def __init__(self, ref_api, oi, frame_range, is_leader, tripoint_range=None):
        import torch
        self.torch, self.api, self.oi = torch, ref_api, oi
        self.f0, self.f1 = frame_range
        self.is_leader = is_leader
        self.Nstate = ref_api.num_states(**oi)
        self.Nmeas_global = ref_api.num_measurements(**oi)
        self.Ni  = ref_api.num_states_intrinsics(**oi)
        self.Nce = 0 if oi.get("rt_cam_ref") is None else oi["rt_cam_ref"].shape[0]
        self.Nf  = 0 if oi.get("rt_ref_frame") is None else oi["rt_ref_frame"].shape[0]
        self.Nie = self.Ni + ref_api.num_states_extrinsics(**oi)
        self.NE  = ref_api.num_states_frames(**oi)
        self.iwarp = ref_api.state_index_calobject_warp(**oi)
        Nwarp = 0 if self.iwarp is None else 2
        self.Nc  = self.Nie + Nwarp
        self.sidx = np.concatenate((np.arange(self.Nie), (self.iwarp + np.arange(2)) if Nwarp else np.zeros((0,), int))).astype(int)
        self.eidx_local = self.Nie + np.arange(6*self.f0, 6*self.f1) if self.NE else np.zeros((0,), int)
        if self.NE == 0: self.f0 = self.f1 = 0
        ob = oi.get("observations_board")
        Nobs_board = 0 if ob is None else ob.shape[0]
        H, W = (ob.shape[1:3] if Nobs_board else (0, 0))
        self.HW = H*W
        self.Ncorners_global = Nobs_board*H*W
        self.do_outlier_rejection = bool(oi.get("do_apply_outlier_rejection", True))
        if Nobs_board:
            frames = oi["indices_frame_camintrinsics_camextrinsics"][:,0]
            self.local_obs = np.nonzero((frames >= self.f0) & (frames < self.f1))[0]
        else:
            self.local_obs = np.zeros((0,), int)
        rows = (self.local_obs[:,None]*2*H*W + np.arange(2*H*W)[None,:]).ravel()
        Nboard_rows = Nobs_board*2*H*W
        # triangulated points: the rows of the pairs of MY point sets (parallel.partition_triangulated)
        Ntri = ref_api.num_measurements_points_triangulated(**oi)
        if Ntri:
            from test_triangulated import enumerate_pairs
            flags = ref_api._ingest(dict(oi), callback=True).c_tri["flags"]
            set_of_obs = np.concatenate(((0,), np.cumsum(flags & 1)[:-1]))
            pairs = enumerate_pairs(flags)
            assert len(pairs) == Ntri
            m0 = ref_api.measurement_index_points_triangulated(**oi)
            t0, t1 = tripoint_range
            mine = [m0 + k for k, (i0, i1) in enumerate(pairs) if t0 <= set_of_obs[i0] < t1]
            rows = np.concatenate((rows, np.array(mine, dtype=int)))
        if is_leader:
            # what belongs to nobody's frames or points: the regularization rows (and discrete points, which this
            # stand-in leaves whole with the leader)
            first = Nboard_rows + ref_api.num_measurements_points(**oi) + Ntri
            rows = np.concatenate((rows, np.arange(Nboard_rows, Nboard_rows + ref_api.num_measurements_points(**oi)),
                                   np.arange(first, self.Nmeas_global)))
        self.rows = rows.astype(int)
        z = lambda n: np.zeros(n)
        self.op = [dict(b=z(self.Nstate), g=z(self.Nstate), N=None, x=None, norm2_x=0.0, gNg=0.0, gg=0.0,
                        step_cauchy=z(self.Nstate), step_gn=z(self.Nstate), sNs=0.0, gs=0.0) for _ in range(2)]
        Nc = self.Nc
        self.comm = [torch.zeros(Nc*Nc + 2*Nc + 2, dtype=torch.float64),       # [S | r | g_S | |x|^2 | status]
                     torch.zeros(4, dtype=torch.float64)]                     # [gNg | ggE | gnE2 | gnE.gE]
        self.step = z(self.Nstate)
        self.current = 0
        self.lam = 0.0
        self.ring = [None]*8
        # the seed
        self.op[0]["b"][:] = ref_api.optimizer_callback(no_jacobian=True, no_factorization=True, **oi)[0]

    def context(self):
        return contextlib.nullcontext()
    def b_current(self):
        return self.op[self.current]["b"].copy()

    # ---- the interface ---------------------------------------------------
    def reset(self, check_termination, max_iterations, trustregion0):
        self.lam = 0.0                # a new run starts unregularized
        self.ctl = dict(done=0, error=0, check=int(bool(check_termination)), maxit=max_iterations,
                        tr=float(trustregion0), lam=self.lam, ib=self.current, ia=1-self.current,
                        Nsteps_accepted=0, Ntrials=0, Nfactorizations=0, Nevaluations=0,
                        norm2_x=[0.0,0.0], cauchy_lensq=[0.0,0.0], gn_valid=[0,0], gn_lensq=[0.0,0.0],
                        gn_dot_g=[0.0,0.0], gn_lambda=[0.0,0.0], edge=[0,0],
                        refactor=0, gn_fresh=0, derive=0, step_len_sq=0.0)
        # the first thing that happens is the assembly + elimination of the starting point
        self.fl = dict(skip_eval=0, elim_mode=1, elim_sel=self.current, skip_backsub=1)
        self.status = 0
    def comm_buffer(self, which):
        return self.comm[which]
    def snapshot(self, slot):
        self.ring[slot] = dict(self.ctl)
    def wait(self, slot):
        return self.ring[slot]
    def finish(self):
        c = self.ctl
        self.current = c["ib"]
        self.lam     = c["lam"]
        return dict(Nsteps_accepted=c["Nsteps_accepted"], Nevaluations=c["Nevaluations"],
                    Nfactorizations=c["Nfactorizations"], Ntrials=c["Ntrials"], error=c["error"],
                    trustregion=c["tr"], norm2_x=c["norm2_x"][c["ib"]], lambda_=c["lam"])
    def enqueue(self, initial, seg):
        getattr(self, f"_seg{seg}")(bool(initial))
    def masked_state(self):
        b = self.op[self.current]["b"].copy()
        mine = np.zeros(self.Nstate, dtype=bool)
        if self.is_leader:
            mine[self.sidx] = True
        mine[self.eidx_local] = True
        b[~mine] = 0.0
        return self.torch.from_numpy(b)
    def set_state(self, t):
        self.op[self.current]["b"][:] = t.numpy()

    # ---- segment 0: choose | evaluate | assemble + eliminate | my summand of [S | r | g_S | |x|^2 | status]
    def _seg0(self, initial):
        c, fl = self.ctl, self.fl
        if not initial:
            self._choose()
        out = self.comm[0].numpy()
        mode = fl["elim_mode"]
        if mode == 0:
            return                              # (the kernels leave the buffer alone; nobody reads it)
        O = self.op[fl["elim_sel"]]
        if mode == 1:
            self._evaluate(fl["elim_sel"])
        Nc = self.Nc
        self.status = 0
        S, r = self._eliminate(O, c["lam"], add_g=(mode == 1) or self.is_leader)
        out[:Nc*Nc]             = S.ravel()
        out[Nc*Nc:Nc*Nc+Nc]     = r
        out[Nc*Nc+Nc:Nc*Nc+2*Nc] = O["g"][self.sidx]
        out[Nc*Nc+2*Nc]         = O["norm2_x"]
        out[Nc*Nc+2*Nc+1]       = float(self.status)

    # ---- segment 1: accept/reject | Cholesky | back-substitution | my summand of the four scalars
    def _seg1(self, initial):
        c, fl = self.ctl, self.fl
        Nc = self.Nc
        v = self.comm[0].numpy()
        mode = fl["elim_mode"]
        ip = c["ib"] if initial else c["ia"]
        norm2_x = float(v[Nc*Nc+2*Nc]); eblock_failed = v[Nc*Nc+2*Nc+1] != 0.0
        # step2_finish
        if mode == 1:
            P = self.op[ip]
            P["norm2_x_global"] = norm2_x
            c["norm2_x"][ip]  = norm2_x
            c["gn_valid"][ip] = 0
            c["edge"][ip]     = 0
            c["Nevaluations"] += 1
            if not initial: self._accept()
        go = unpack = 0
        c["gn_fresh"] = 0; c["derive"] = 0
        if not c["done"] and c["check"] and c["Nsteps_accepted"] >= c["maxit"]:
            c["done"] = 1
        ib = c["ib"]
        if mode == 1 and ib == ip:
            unpack = 1; c["derive"] = 1
        if not c["done"]:
            if mode == 2:   go = 1
            elif mode == 1: go = int(ib == ip)
        if go and eblock_failed:
            self._raise_lambda(); c["refactor"] = 1; go = 0
        if go: c["Nfactorizations"] += 1
        fl["skip_backsub"] = 1
        if unpack:
            self.op[ip]["g"][self.sidx] = v[Nc*Nc+Nc:Nc*Nc+2*Nc]
        O = self.op[c["ib"]]
        out = self.comm[1].numpy()
        out[:] = 0.0
        if go:
            S = v[:Nc*Nc].reshape(Nc, Nc)
            S = np.tril(S) + np.tril(S, -1).T          # (only the lower triangle is meaningful)
            r = v[Nc*Nc:Nc*Nc+Nc]
            ok = np.all(np.isfinite(S))
            if ok:
                try:    np.linalg.cholesky(S)
                except np.linalg.LinAlgError: ok = False
            if not ok:
                self._raise_lambda(); c["refactor"] = 1
            else:
                c["refactor"] = 0
                c["gn_valid"][ib]  = 1
                c["gn_lambda"][ib] = c["lam"]
                c["gn_fresh"]      = 1
                fl["skip_backsub"] = 0
                ds = -np.linalg.solve(S, r)
                O["step_gn"][self.sidx] = ds
                for e, B, Dinv in self.fact:
                    O["step_gn"][e] = -Dinv @ (O["g"][e] + B.T @ ds)
                d = O["step_gn"][self.eidx_local]
                out[2] = float(d @ d)
                out[3] = float(d @ O["g"][self.eidx_local])
        if c["derive"]:
            g = O["g"]
            out[0] = float(g @ O["N"] @ g)                              # my rows' part of g^T JtJ g
            out[1] = float(g[self.eidx_local] @ g[self.eidx_local])

    # ---- the control logic ------------------------------------------------
    def _raise_lambda(self):
        c = self.ctl
        c["lam"] = 1e-10 if c["lam"] == 0.0 else c["lam"]*10.0
        if not c["lam"] < 1e30:
            c["error"] = c["done"] = 1

    def _accept(self):
        # ctl_accept (solver_device.hpp)
        c = self.ctl
        ib, ia = c["ib"], c["ia"]
        expected = -2.0*self.op[ib]["gs"] - self.op[ib]["sNs"]
        rho = (c["norm2_x"][ib] - c["norm2_x"][ia])/expected
        if not (rho == rho) or not (c["norm2_x"][ia] == c["norm2_x"][ia]): rho = -1.0
        if rho < 0.25:                     c["tr"] *= 0.1
        elif rho > 0.75 and c["edge"][ib]: c["tr"] *= 2.0
        if rho > 0.0:
            c["ib"], c["ia"] = ia, ib
            c["Nsteps_accepted"] += 1
        elif c["check"] and (c["tr"] < 0.0 or c["tr"] == 0.0 or c["tr"] != c["tr"]):
            c["done"] = 1

    def _choose(self):
        # step2_choose_kernel
        c, fl = self.ctl, self.fl
        if c["done"]:
            fl.update(skip_eval=1, elim_mode=0)
            return
        ib, ia = c["ib"], c["ia"]
        O = self.op[ib]
        v2 = self.comm[1].numpy()
        if c["derive"]:
            ggS = float(O["g"][self.sidx] @ O["g"][self.sidx])
            gNg, gg = float(v2[0]), ggS + float(v2[1])
            kcau = -gg/gNg if gNg > 0.0 else 0.0
            norm2a = kcau*kcau*gg
            O["step_cauchy"][:] = kcau*O["g"]
            O["gNg"], O["gg"] = gNg, gg
            c["cauchy_lensq"][ib] = norm2a
        else:
            gNg, gg = O["gNg"], O["gg"]
            kcau = -gg/gNg if gNg > 0.0 else 0.0
            norm2a = c["cauchy_lensq"][ib]
        tr, dsq = c["tr"], c["tr"]**2
        cauchy_only = norm2a >= dsq
        if c["refactor"] or (not cauchy_only and not c["gn_valid"][ib]):
            c["refactor"] = 1
            self.status = 0
            fl.update(skip_eval=1, elim_mode=2, elim_sel=ib)
            return
        fresh = (not cauchy_only) and c["gn_fresh"]
        gn_lensq, gn_dot_g = c["gn_lensq"][ib], c["gn_dot_g"][ib]
        if fresh:
            gnS = O["step_gn"][self.sidx]
            gn_lensq = float(gnS @ gnS) + float(v2[2])
            gn_dot_g = float(gnS @ O["g"][self.sidx]) + float(v2[3])
            if not gn_lensq == gn_lensq:
                self._raise_lambda()
                c["refactor"] = 1; c["gn_valid"][ib] = 0
                fl.update(skip_eval=1, elim_mode=(0 if c["done"] else 2), elim_sel=ib)
                return
        norm2b = ab = 0.0
        if cauchy_only:
            kc, kg, len_sq, edge = tr/math.sqrt(norm2a), 0.0, dsq, 1
        else:
            norm2b, ab = gn_lensq, kcau*gn_dot_g
            if norm2b <= dsq:
                kc, kg, len_sq, edge = 0.0, 1.0, norm2b, 0
            else:
                l2    = norm2a - 2.0*ab + norm2b
                neg_c = norm2a - ab
                disc  = max(neg_c*neg_c - l2*(norm2a - dsq), 0.0)
                k     = (neg_c + math.sqrt(disc))/l2
                kc, kg = 1.0-k, k
                len_sq = kc*kc*norm2a + 2.0*kg*kc*ab + kg*kg*norm2b
                edge = 1
        self.step[:] = kc*O["step_cauchy"] + (kg*O["step_gn"] if kg != 0.0 else 0.0)
        self.op[ia]["b"][:] = O["b"] + self.step
        if fresh:
            c["gn_lensq"][ib], c["gn_dot_g"][ib] = gn_lensq, gn_dot_g
        # the expected improvement from dot products (no pass over JtJ)
        sNs, gs = kc*kc*kcau*kcau*gNg, kc*kcau*gg
        if kg != 0.0:
            a, lam = gn_dot_g, c["gn_lambda"][ib]
            sNs += 2.0*kc*kg*(-kcau*gg - lam*ab) + kg*kg*(-a - lam*norm2b)
            gs  += kg*a
        O["sNs"], O["gs"] = sNs, gs
        c["step_len_sq"] = len_sq
        c["edge"][ib] = edge
        c["Ntrials"] += 1
        self.status = 0
        if c["check"] and len_sq < 1e-7**2:
            c["done"] = 1
            fl.update(skip_eval=1, elim_mode=0)
        else:
            fl.update(skip_eval=0, elim_mode=1, elim_sel=ia)

    # ---- pieces -----------------------------------------------------------
    def inputs_at(self, b):
        """optimization_inputs with the state b: whatever blocks are in the state go back into their arrays"""
        oi = dict(self.oi)
        u = b.copy()
        self.api.unpack_state(u, **self.oi)
        i = self.api.state_index_intrinsics(0, **self.oi)
        if i is not None and self.Ni:
            Ncam = oi["intrinsics"].shape[0]
            per  = self.Ni//Ncam
            core = bool(self.oi.get("do_optimize_intrinsics_core", True))
            intr = oi["intrinsics"].copy()
            blk  = u[i:i+self.Ni].reshape(Ncam, per)
            if per == intr.shape[1]: intr[:] = blk
            elif core:               intr[:,:4] = blk          # the core alone
            else:                    intr[:,4:] = blk          # the distortions alone
            oi["intrinsics"] = np.ascontiguousarray(intr)
        i = self.api.state_index_extrinsics(0, **self.oi)
        if i is not None: oi["rt_cam_ref"] = np.ascontiguousarray(u[i:i+6*self.Nce].reshape(self.Nce, 6))
        i = self.api.state_index_frames(0, **self.oi)
        if i is not None: oi["rt_ref_frame"] = np.ascontiguousarray(u[i:i+6*self.Nf].reshape(self.Nf, 6))
        if self.iwarp is not None: oi["calobject_warp"] = np.ascontiguousarray(u[self.iwarp:self.iwarp+2])
        return oi

    def _evaluate(self, iop):
        O = self.op[iop]
        # (the frame poses of the other shards' frames in MY state are stale: my rows do not depend on them)
        _, x, J, _ = self.api.optimizer_callback(no_factorization=True, **self.inputs_at(O["b"]))
        Jl = J[self.rows].toarray()
        xl = x[self.rows]
        O["x"] = x
        O["N"] = Jl.T @ Jl
        O["g"][:] = Jl.T @ xl
        O["norm2_x"] = float(xl @ xl)

    def _eliminate(self, O, lam, add_g):
        N, g = O["N"], O["g"]
        S = N[np.ix_(self.sidx, self.sidx)].copy()
        r = np.zeros(self.Nc)
        if self.is_leader: S += lam*np.eye(self.Nc)
        if add_g:          r += g[self.sidx]
        self.fact = []
        for f in range(self.f0, self.f1):
            e = self.Nie + 6*f + np.arange(6)
            D = N[np.ix_(e,e)] + lam*np.eye(6)
            B = N[np.ix_(self.sidx, e)]
            try:
                np.linalg.cholesky(D)
            except np.linalg.LinAlgError:
                self.status = 1
                D = np.eye(6)
            Dinv = np.linalg.inv(D)
            S -= B @ Dinv @ B.T
            r -= B @ Dinv @ g[e]
            self.fact.append((e, B, Dinv))
        return S, r

    def _local_corners(self):
        x = self.op[self.current]["x"]
        if self.Ncorners_global == 0:
            e = np.zeros((0,))
            return np.zeros((0,), int), e, e, np.zeros((0,3))
        pool = self.oi["observations_board"].reshape(-1,3)
        idx = (self.local_obs[:,None]*self.HW + np.arange(self.HW)[None,:]).ravel()
        return idx, x[2*idx], x[2*idx+1], pool
    def outlier_stats(self, thresh_sq):
        idx, dx, dy, pool = self._local_corners()
        w = pool[idx,2]
        inl = w > 0
        nbig = 0
        if thresh_sq >= 0:
            nbig = int(np.count_nonzero(inl & ((dx*dx > thresh_sq) | (dy*dy > thresh_sq))))
        return self.torch.tensor([float(np.count_nonzero(~inl)), float(nbig),
                                  float(np.sum((dx*dx + dy*dy)[inl]))], dtype=self.torch.float64)
    def mark_outliers(self, thresh_sq):
        idx, dx, dy, pool = self._local_corners()
        m = (pool[idx,2] > 0) & ((dx*dx > thresh_sq) | (dy*dy > thresh_sq))
        pool[idx[m],2] *= -1.   # in the caller's array, like the reference
        return self.torch.tensor([float(np.count_nonzero(m))], dtype=self.torch.float64)


def _worker(rank, world, port, seed, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mrcal_amd._cabi import MrcalLib
    from mrcal_amd._api  import Api
    from mrcal_amd.synthetic import make_calibration_problem
    from mrcal_amd.parallel import ShardedDogleg, Communicator, partition_frames
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = Api(MrcalLib(REFLIB_PATH))
    oi, _ = make_calibration_problem(ref, Ncameras=2, Nframes=9, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=6, object_height_n=5, seed=seed)
    ranges = partition_frames(oi["indices_frame_camintrinsics_camextrinsics"], 9, world)
    shard  = NumpyShard(ref, oi, ranges[rank], rank == 0)
    dl     = ShardedDogleg(shard, Communicator())
    stats  = dl.solve()
    b      = shard.b_current()
    # every rank must hold the same state
    if world > 1:
        tb = torch.from_numpy(b.copy())
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        assert np.array_equal(tb.numpy(), b), "ranks disagree on the solution"
        # gather the outlier marks (each rank marked its own frames)
        w = torch.from_numpy(oi["observations_board"][...,2].copy())
        dist.all_reduce(w, op=dist.ReduceOp.MIN)
        weights = w.numpy()
    else:
        weights = oi["observations_board"][...,2].copy()
    if rank == 0:
        np.savez(out_path, b=b, weights=weights, rms=stats["rms_reproj_error__pixels"],
                 Noutliers=stats["Noutliers_board"], Nevaluations=stats["Nevaluations"],
                 Ncollectives=dl.comm.Ncollectives, Ntrials=dl.Ntrials_total, Npasses=stats["Noutlier_passes"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, tmp_path, seed=3):
    import torch.multiprocessing as mp
    out = str(tmp_path / f"world{world}.npz")
    port = 29500 + (os.getpid() % 500) + 7*world
    if world == 1:
        _worker(0, 1, port, seed, out)
    else:
        mp.spawn(_worker, args=(world, port, seed, out), nprocs=world, join=True)
    return np.load(out)


@pytest.mark.timeout(600)
def test_sharded_solve_matches_unsharded_and_reference(tmp_path, ref_api):
    from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
    r1 = _run(1, tmp_path)
    r2 = _run(2, tmp_path)
    r3 = _run(3, tmp_path)
    for r in (r2, r3):
        # both stop within a 1e-7 (packed units) step of the optimum; summation
        # order differs between the shardings
        assert np.abs(r["b"] - r1["b"]).max() < 2e-5
        assert abs(r["rms"] - r1["rms"]) < 1e-8
        assert int(r["Noutliers"]) == int(r1["Noutliers"])
        assert np.array_equal(r["weights"] < 0, r1["weights"] < 0)
        # TWO sums over the shards per trial step (the starting point of a pass counts as
        # one), a few per outlier pass, one for the final gather of the state
        assert 0 < int(r["Ncollectives"]) <= 2*int(r["Ntrials"]) + 4*(int(r["Npasses"]) + 1) + 1

    # and against the reference's own mrcal_optimize() (+ restated libdogleg)
    oi, _ = make_calibration_problem(ref_api, Ncameras=2, Nframes=9, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=6, object_height_n=5, seed=3)
    oi = copy_inputs(oi)
    s = ref_api.optimize(**oi)
    assert abs(s["rms_reproj_error__pixels"] - float(r2["rms"])) < 1e-6*s["rms_reproj_error__pixels"]
    assert np.abs(s["b_packed"] - r2["b"]).max() < 2e-5
    assert int(s["Noutliers_board"]) == int(r2["Noutliers"])
    assert np.array_equal(oi["observations_board"][...,2] < 0, r2["weights"] < 0)


def _sfm_worker(rank, world, port, Nboard_frames, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from mrcal_amd._cabi import MrcalLib
    from mrcal_amd._api  import Api
    from mrcal_amd.parallel import ShardedDogleg, Communicator, partition_frames, partition_triangulated
    from test_triangulated import sfm_problem
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = Api(MrcalLib(REFLIB_PATH))
    oi, _ = sfm_problem("LENSMODEL_OPENCV4", Ncam=3, Npoints=50, seed=11, noise=0.3, Nboard_frames=Nboard_frames, board_wh=(5,4))
    flags = ref._ingest(dict(oi), callback=True).c_tri["flags"]
    tr = partition_triangulated(flags & 1, world)
    fr = partition_frames(oi["indices_frame_camintrinsics_camextrinsics"], Nboard_frames, world) if Nboard_frames else [(0,0)]*world
    shard = NumpyShard(ref, oi, fr[rank], rank == 0, tripoint_range=tr[rank])
    dl    = ShardedDogleg(shard, Communicator())
    stats = dl.solve()
    b     = shard.b_current()
    Nrows = torch.tensor([float(len(shard.rows))], dtype=torch.float64)
    if world > 1:
        tb = torch.from_numpy(b.copy())
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        assert np.array_equal(tb.numpy(), b), "ranks disagree on the solution"
        dist.all_reduce(Nrows)
    if rank == 0:
        np.savez(out_path, b=b, rms=stats["rms_reproj_error__pixels"], Nrows=Nrows.numpy(), Nmeas=shard.Nmeas_global,
                 my_rows=len(shard.rows), Ncollectives=dl.comm.Ncollectives, Ntrials=dl.Ntrials_total)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("Nboard_frames", (0, 6))
def test_sharded_triangulated_solve(tmp_path, ref_api, Nboard_frames):
    """world 2 over gloo on a problem with TRIANGULATED points (sharded by point, the Jacobian touches
    the extrinsics only, the camera block is what is summed), alone and with board frames beside them: every
    measurement row in exactly one shard, the same optimum as the unsharded run of the same driver and as the
    reference's mrcal_optimize()"""
    import torch.multiprocessing as mp
    from mrcal_amd.synthetic import copy_inputs
    from test_triangulated import sfm_problem
    res = {}
    for world in (1, 2):
        out  = str(tmp_path / f"sfm{world}.npz")
        port = 29100 + (os.getpid() % 300) + 11*world + Nboard_frames
        if world == 1: _sfm_worker(0, 1, port, Nboard_frames, out)
        else:          mp.spawn(_sfm_worker, args=(world, port, Nboard_frames, out), nprocs=world, join=True)
        res[world] = np.load(out)
    r1, r2 = res[1], res[2]
    assert int(r2["Nrows"][0]) == int(r2["Nmeas"]) and 0 < int(r2["my_rows"]) < int(r2["Nmeas"])
    oi, _ = sfm_problem("LENSMODEL_OPENCV4", Ncam=3, Npoints=50, seed=11, noise=0.3, Nboard_frames=Nboard_frames, board_wh=(5,4))
    def unpacked(b):
        u = np.array(b, dtype=float); ref_api.unpack_state(u, **oi); return u
    # in UNPACKED units (radians, metres): a camera rotation is packed in units of 0.1 degree (scales.h), so the
    # 1e-7-packed-units termination of the dog-leg leaves the poses equal to ~1e-6 rad, not to 2e-5 packed units
    # (observed 1.3e-5 m in the 1.6 m translation of the camera unity_cam01 does not pin, 9e-7 rad in the rotations)
    assert np.abs(unpacked(r2["b"]) - unpacked(r1["b"])).max() < 5e-5
    assert abs(float(r2["rms"]) - float(r1["rms"])) < 1e-7
    assert 0 < int(r2["Ncollectives"]) <= 2*int(r2["Ntrials"]) + 5
    oi2 = copy_inputs(oi)
    s = ref_api.optimize(**oi2)
    assert abs(s["rms_reproj_error__pixels"] - float(r2["rms"])) < 1e-6*s["rms_reproj_error__pixels"]
    assert np.abs(unpacked(s["b_packed"]) - unpacked(r2["b"])).max() < 5e-5


def test_one_collective_takes_three_camera_blocks():
    """DESIGN.md section 7, "Why two collectives": the four scalars the dog-leg's next choice needs from all ranks -
    g^T N g, |g_E|^2, |gn_E|^2, gn_E . g_E - can ride in the FIRST all-reduce only as coefficients of forms in vectors
    that exist after it (the summed camera gradient g_S, the solution d_s of the summed system). This test builds three
    shards of a small block problem with numpy, sums ONE buffer
        [ S | r | g_S | M | m | A | Bt^T g_E | |x|^2, status, c, |y|^2, g_E^T D g_E, |g_E|^2 ]     3 Nc^2 + 4 Nc + 6
    and recovers from it, on every rank alike, what the two-collective step computes with the second all-reduce. It
    also shows what the review's 2 Nc^2 + 4 Nc + 6 misses: without the summed A, g^T N g cannot be formed (each rank's
    A_r meets the SUMMED g_S on both sides)"""
    rng = np.random.RandomState(11)
    Nc, Nf, world = 9, 12, 3
    def spd(n, scale=1.0):
        a = rng.normal(size=(n + 3, n))
        return scale*(a.T @ a)
    # a consistent problem: J = [Jc | Jf blocks], rows assigned to frames, frames to ranks
    rows_per_frame = 14
    Jc = [rng.normal(size=(rows_per_frame, Nc)) for _ in range(Nf)]
    Jf = [rng.normal(size=(rows_per_frame, 6))  for _ in range(Nf)]
    x  = [rng.normal(size=rows_per_frame)       for _ in range(Nf)]
    owner = [f % world for f in range(Nf)]
    # what every rank has from its own frames
    local = []
    for rk in range(world):
        fr = [f for f in range(Nf) if owner[f] == rk]
        A   = sum(Jc[f].T @ Jc[f] for f in fr)
        gS  = sum(Jc[f].T @ x[f]  for f in fr)
        Bt  = {f: Jf[f].T @ Jc[f] for f in fr}
        D   = {f: Jf[f].T @ Jf[f] for f in fr}
        gE  = {f: Jf[f].T @ x[f]  for f in fr}
        local.append(dict(fr=fr, A=A, gS=gS, Bt=Bt, D=D, gE=gE, n2=sum(x[f] @ x[f] for f in fr)))
    # the ONE buffer
    def summand(L):
        S = L["A"].copy(); r = L["gS"].copy()
        M = np.zeros((Nc, Nc)); m = np.zeros(Nc); Btg = np.zeros(Nc)
        c = y2 = gDg = gE2 = 0.0
        for f in L["fr"]:
            Lf = np.linalg.cholesky(L["D"][f])
            Wt = np.linalg.solve(Lf, L["Bt"][f]); y = np.linalg.solve(Lf, L["gE"][f])
            S -= Wt.T @ Wt; r -= Wt.T @ y
            Z  = np.linalg.solve(Lf.T, Wt); z = np.linalg.solve(Lf.T, y)
            M += Z.T @ Z; m += Z.T @ z; c += z @ z; y2 += y @ y
            Btg += L["Bt"][f].T @ L["gE"][f]
            gDg += L["gE"][f] @ L["D"][f] @ L["gE"][f]; gE2 += L["gE"][f] @ L["gE"][f]
        return np.concatenate([S.ravel(), r, L["gS"], M.ravel(), m, L["A"].ravel(), Btg, [L["n2"], 0.0, c, y2, gDg, gE2]])
    buf = sum(summand(L) for L in local)
    assert buf.size == 3*Nc*Nc + 4*Nc + 6
    o = 0
    def take(n):
        nonlocal o
        v = buf[o:o+n]; o += n
        return v
    S = take(Nc*Nc).reshape(Nc, Nc); r = take(Nc); gS = take(Nc); M = take(Nc*Nc).reshape(Nc, Nc); m = take(Nc)
    A = take(Nc*Nc).reshape(Nc, Nc); Btg = take(Nc); n2, status, c, y2, gDg, gE2 = take(6)
    ds = -np.linalg.solve(S, r)
    one = dict(gNg   = gS @ A @ gS + 2.0*(gS @ Btg) + gDg,
               gE2   = gE2,
               gnE2  = ds @ M @ ds + 2.0*(m @ ds) + c,
               gnEgE = -(y2 + (gS - r) @ ds))
    # the two-collective step: the second all-reduce sums what each rank computes from ITS frames with the summed g_S, d_s
    two = dict(gNg=0.0, gE2=0.0, gnE2=0.0, gnEgE=0.0)
    for L in local:
        g_loc = 0.0
        for f in L["fr"]:
            dE = -np.linalg.solve(L["D"][f], L["gE"][f] + L["Bt"][f] @ ds)
            two["gnE2"]  += dE @ dE
            two["gnEgE"] += dE @ L["gE"][f]
            two["gE2"]   += L["gE"][f] @ L["gE"][f]
            g_loc += 2.0*(L["gE"][f] @ (L["Bt"][f] @ gS)) + L["gE"][f] @ L["D"][f] @ L["gE"][f]
        two["gNg"] += gS @ L["A"] @ gS + g_loc          # (its own A_r against the SUMMED g_S)
    # ... and the whole thing against the dense J
    J = np.zeros((Nf*rows_per_frame, Nc + 6*Nf)); xx = np.concatenate(x)
    for f in range(Nf):
        J[f*rows_per_frame:(f+1)*rows_per_frame, :Nc] = Jc[f]
        J[f*rows_per_frame:(f+1)*rows_per_frame, Nc+6*f:Nc+6*f+6] = Jf[f]
    g = J.T @ xx; N = J.T @ J
    gn = -np.linalg.solve(N, g)
    dense = dict(gNg=g @ N @ g, gE2=g[Nc:] @ g[Nc:], gnE2=gn[Nc:] @ gn[Nc:], gnEgE=gn[Nc:] @ g[Nc:])
    for k in one:
        assert abs(one[k] - two[k])   < 1e-9*abs(two[k]),   (k, one[k], two[k])
        assert abs(one[k] - dense[k]) < 1e-8*abs(dense[k]), (k, one[k], dense[k])
    assert np.abs(ds - gn[:Nc]).max() < 1e-9*np.abs(gn[:Nc]).max()
    # without the summed A (the 2 Nc^2 buffer): g_S^T A g_S from the pieces a rank has is its own A_r only
    partial = gS @ local[0]["A"] @ gS
    assert abs(partial - gS @ A @ gS) > 1e-3*abs(gS @ A @ gS)
