"""The N>1 control flow (mrcal_amd/parallel.py) on CPU: world_size 2 and 3 over
gloo, with a numpy shard standing in for the GPU phase kernels. The numpy shard
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


class NumpyShard:
    """CPU stand-in for GpuShard (same interface), for ONE rank"""
    def __init__(self, ref_api, oi, frame_range, is_leader):
        import torch
        self.torch, self.api, self.oi = torch, ref_api, oi
        self.f0, self.f1 = frame_range
        self.is_leader = is_leader
        self.Nstate = ref_api.num_states(**oi)
        self.Nmeas_global = ref_api.num_measurements(**oi)
        Ncam_i = oi["intrinsics"].shape[0]
        self.Nintr = ref_api.num_intrinsics_optimization_params(**oi)
        self.Nce = oi["rt_cam_ref"].shape[0]
        self.Nf  = oi["rt_ref_frame"].shape[0]
        self.Nie = Ncam_i*self.Nintr + 6*self.Nce
        self.NE  = 6*self.Nf
        self.iwarp = ref_api.state_index_calobject_warp(**oi)
        self.Nc  = self.Nie + 2
        self.sidx = np.concatenate((np.arange(self.Nie), self.iwarp + np.arange(2)))
        H, W = oi["observations_board"].shape[1:3]
        self.HW = H*W
        self.Ncorners_global = oi["observations_board"].shape[0]*H*W
        self.do_outlier_rejection = bool(oi.get("do_apply_outlier_rejection", True))
        frames = oi["indices_frame_camintrinsics_camextrinsics"][:,0]
        self.local_obs = np.nonzero((frames >= self.f0) & (frames < self.f1))[0]
        rows = (self.local_obs[:,None]*2*H*W + np.arange(2*H*W)[None,:]).ravel()
        Nboard_rows = oi["observations_board"].shape[0]*2*H*W
        if is_leader:
            rows = np.concatenate((rows, np.arange(Nboard_rows, self.Nmeas_global)))
        self.rows = rows
        self.bufs = {}
        self.N = [None, None]
        self.xs = [None, None]
        self.current = 0
        # the seed
        b0 = ref_api.optimizer_callback(no_jacobian=True, no_factorization=True, **oi)[0]
        self.vec("b", 0)[:] = torch.from_numpy(b0)

    def context(self):
        return contextlib.nullcontext()
    def new_buffer(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64)
    def vec(self, name, iop):
        sizes = dict(b=self.Nstate, g=self.Nstate, step_cauchy=self.Nstate, step_gn=self.Nstate,
                     scalars=8, step=self.Nstate, schur=self.Nc*self.Nc+self.Nc, status=1)
        key = (name, iop if name not in ("step", "schur", "status") else 0)
        if key not in self.bufs:
            dt = self.torch.int32 if name == "status" else self.torch.float64
            self.bufs[key] = self.torch.zeros(sizes[name], dtype=dt)
        return self.bufs[key]
    def set_current(self, iop):
        self.current = iop

    def inputs_at(self, b):
        """optimization_inputs with the state b (all variables optimized)"""
        oi = dict(self.oi)
        u = b.copy()
        self.api.unpack_state(u, **self.oi)
        Ncam = oi["intrinsics"].shape[0]
        oi["intrinsics"]   = np.ascontiguousarray(u[:Ncam*self.Nintr].reshape(Ncam, self.Nintr))
        oi["rt_cam_ref"]   = np.ascontiguousarray(u[Ncam*self.Nintr:self.Nie].reshape(self.Nce, 6))
        oi["rt_ref_frame"] = np.ascontiguousarray(u[self.Nie:self.Nie+self.NE].reshape(self.Nf, 6))
        oi["calobject_warp"] = np.ascontiguousarray(u[self.iwarp:self.iwarp+2])
        return oi

    def evaluate(self, iop):
        b = self.vec("b", iop).numpy()
        _, x, J, _ = self.api.optimizer_callback(no_factorization=True, **self.inputs_at(b))
        Jl = J[self.rows].toarray()
        xl = x[self.rows]
        self.xs[iop] = x
        self.N[iop] = Jl.T @ Jl
        self.vec("g", iop)[:] = self.torch.from_numpy(Jl.T @ xl)
        sc = self.vec("scalars", iop)
        sc.zero_()
        sc[0] = float(xl @ xl)
    def quadform(self, iop, v, out):
        vn = v.numpy()
        out += float(vn @ self.N[iop] @ vn)
    def _local_E(self):
        return np.concatenate([self.Nie + 6*f + np.arange(6) for f in range(self.f0, self.f1)]).astype(int) \
            if self.f1 > self.f0 else np.zeros((0,), dtype=int)
    def factor_local(self, iop, lam):
        N, g = self.N[iop], self.vec("g", iop).numpy()
        S = N[np.ix_(self.sidx, self.sidx)].copy()
        r = np.zeros(self.Nc)
        if self.is_leader:
            S += lam*np.eye(self.Nc)
            r += g[self.sidx]
        status = 0
        self.fact = []
        for f in range(self.f0, self.f1):
            e = self.Nie + 6*f + np.arange(6)
            D = N[np.ix_(e,e)] + lam*np.eye(6)
            B = N[np.ix_(self.sidx, e)]
            try:
                np.linalg.cholesky(D)
            except np.linalg.LinAlgError:
                status = 1
                D = np.eye(6)
            Dinv = np.linalg.inv(D)
            S -= B @ Dinv @ B.T
            r -= B @ Dinv @ g[e]
            self.fact.append((e, B, Dinv))
        sr = self.vec("schur", 0).numpy()
        sr[:self.Nc*self.Nc] = S.ravel()
        sr[self.Nc*self.Nc:] = r
        self.vec("status", 0)[0] = status
    def solve_backsub(self, iop):
        sr = self.vec("schur", 0).numpy()
        S = sr[:self.Nc*self.Nc].reshape(self.Nc, self.Nc)
        r = sr[self.Nc*self.Nc:]
        g = self.vec("g", iop).numpy()
        gn = self.vec("step_gn", iop).numpy()
        gn[:] = 0
        try:
            np.linalg.cholesky(S)
            ds = -np.linalg.solve(S, r)
        except np.linalg.LinAlgError:
            self.vec("status", 0)[0] = 1
            ds = np.zeros(self.Nc)
        gn[self.sidx] = ds
        for e, B, Dinv in self.fact:
            gn[e] = -Dinv @ (g[e] + B.T @ ds)
    def _local_corners(self, iop):
        x = self.xs[iop]
        pool = self.oi["observations_board"].reshape(-1,3)
        idx = (self.local_obs[:,None]*self.HW + np.arange(self.HW)[None,:]).ravel()
        return idx, x[2*idx], x[2*idx+1], pool
    def outlier_stats(self, iop, thresh_sq):
        idx, dx, dy, pool = self._local_corners(iop)
        w = pool[idx,2]
        inl = w > 0
        nbig = 0
        if thresh_sq >= 0:
            nbig = int(np.count_nonzero(inl & ((dx*dx > thresh_sq) | (dy*dy > thresh_sq))))
        return self.torch.tensor([float(np.count_nonzero(~inl)), float(nbig),
                                  float(np.sum((dx*dx + dy*dy)[inl]))], dtype=self.torch.float64)
    def mark_outliers(self, iop, thresh_sq):
        idx, dx, dy, pool = self._local_corners(iop)
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
    b      = shard.vec("b", dl.ib).numpy().copy()
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
                 Ncollectives=dl.comm.Ncollectives)
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
        assert int(r["Ncollectives"]) > 0

    # and against the reference's own mrcal_optimize() (+ restated libdogleg)
    oi, _ = make_calibration_problem(ref_api, Ncameras=2, Nframes=9, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=6, object_height_n=5, seed=3)
    oi = copy_inputs(oi)
    s = ref_api.optimize(**oi)
    assert abs(s["rms_reproj_error__pixels"] - float(r2["rms"])) < 1e-6*s["rms_reproj_error__pixels"]
    assert np.abs(s["b_packed"] - r2["b"]).max() < 2e-5
    assert int(s["Noutliers_board"]) == int(r2["Noutliers"])
    assert np.array_equal(oi["observations_board"][...,2] < 0, r2["weights"] < 0)
