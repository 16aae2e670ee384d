import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MRCAL_AMD_HOST_COMM_TIMEOUT="60")
    if rank == 0: os.environ["MRCAL_AMD_DEBUG_SOLVER"] = "1"
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mrcal_amd
    from mrcal_amd.parallel import ShardedProblem
    from mrcal_amd.synthetic import make_calibration_problem
    oi = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", object_width_n=10, object_height_n=10, seed=0)[0]
    sp = ShardedProblem(_driver="host", **oi)
    st = sp.solve()
    if rank == 0: print("SHARDED", {k: st[k] for k in ("Niterations","Nevaluations","Noutliers_board","Noutlier_passes","norm2_x")}, flush=True)
    sp.close(); dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if world > 0:
        import torch.multiprocessing as mp
        mp.spawn(worker, args=(world, 29811), nprocs=world, join=True)
    else:
        os.environ["MRCAL_AMD_DEBUG_SOLVER"] = "1"
        import mrcal_amd
        from mrcal_amd.synthetic import make_calibration_problem
        from mrcal_amd.resident import Problem
        oi = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", object_width_n=10, object_height_n=10, seed=0)[0]
        with Problem(**oi) as p:
            st = p.solve()
            print("SINGLE", {k: st[k] for k in ("Niterations","Nevaluations","Noutliers_board","Noutlier_passes","norm2_x")})
