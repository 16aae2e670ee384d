import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem, ShardedDogleg, GpuShard, Communicator
    from mrcal_amd.resident import Problem
    from mrcal_amd.synthetic import make_calibration_problem
    oi = make_calibration_problem(mrcal_amd._api, Ncameras=3, Nframes=11, lensmodel="LENSMODEL_OPENCV8", seed=5)[0]
    sp = ShardedProblem(**oi)
    dl, s = sp.dogleg, sp.shard
    dl.evaluate(0)
    g = s.vec("g",0).cpu().numpy().copy()
    sc = s.vec("scalars",0).cpu().numpy().copy()
    with s.context():
        s.factor_local(0, 0.0)
        st_loc = s.vec("status",0).cpu().numpy().copy()
        dl.comm.sum(s.vec("schur",0))
        S = s.vec("schur",0).cpu().numpy().copy()
    with s.context():
        gn = s.vec("step_gn",0)
        s.solve_backsub(0)
        st_after = s.vec("status",0).cpu().numpy().copy()
        gn_local = gn.cpu().numpy().copy()
        dl.comm.sum(gn[s.Nie:s.Nie+s.NE])
        gn_glob = gn.cpu().numpy().copy()
        status = s.vec("status",0)
        dl.comm.max(status)
        print(rank, "status after solve", st_after, "after max", status.cpu().numpy(), "gn local finite", np.isfinite(gn_local).all(), "norm", np.linalg.norm(gn_glob), flush=True)
    if rank == 0:
        # unsharded reference in the same process
        class NoComm:
            world=1; rank=0; Ncollectives=0
            def sum(self,t): pass
            def max(self,t): pass
        p1 = Problem(**oi)
        s1 = GpuShard(p1, sp.Nmeas_global, s.Ncorners_global, True)
        d1 = ShardedDogleg(s1, NoComm())
        d1.evaluate(0)
        g1 = s1.vec("g",0).cpu().numpy(); sc1 = s1.vec("scalars",0).cpu().numpy()
        with s1.context():
            s1.factor_local(0, 0.0)
            S1 = s1.vec("schur",0).cpu().numpy()
        print("frames", sp.frame_range, "status_local", st_loc)
        print("g diff", np.abs(g-g1).max(), np.abs(g1).max(), "scalars", sc[:3], sc1[:3])
        Nc = s.Nc
        A = S[:Nc*Nc].reshape(Nc,Nc); A1 = S1[:Nc*Nc].reshape(Nc,Nc)
        iu = np.triu_indices(Nc)
        print("S upper diff", np.abs(A[iu]-A1[iu]).max(), np.abs(A1[iu]).max(), "r diff", np.abs(S[Nc*Nc:]-S1[Nc*Nc:]).max())
    else:
        print("rank1 frames", sp.frame_range, "status_local", st_loc)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, 29711), nprocs=2, join=True)
