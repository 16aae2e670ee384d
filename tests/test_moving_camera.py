"""The elimination set is a property of the problem: a camera that moves in front of a stationary
board is many rt_cam_ref and one frame, and then the EXTRINSICS are the numerous, mutually independent 6x6 blocks
(test_calibration_helpers.py:422-493, _apply_moving_ref, builds such problems from ordinary ones). The state
vector, x and J stay the reference's; what changes is which blocks the solver eliminates (NormalDims,
csrc/solver_kernels.hpp): the camera block shrinks from 6 Ncameras + ... to the intrinsics + the frame + the warp.

  - the callback does not depend on it (x, J against the compiled reference)
  - the normal equations in either partition == J^T J
  - the solve with the extrinsics eliminated == the solve with the frames eliminated == the reference's
    mrcal_optimize() on the checker: outliers, rms, state
  - which one is taken: the stationary problems of the other tests keep the frames' elimination
"""
import os
import numpy as np
import pytest

from conftest import relative_error
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from test_solver_parity import blocks_to_dense, dense_normal


def moving_camera_problem(api, Nposes, ref_frame0, seed=3, lensmodel="LENSMODEL_OPENCV4", noise=True):
    """one camera, Nposes images of ONE stationary board: what _apply_moving_ref makes of a monocular
    calibration. ref_frame0: the reference coordinate system is the board (one locked frame at identity, every
    pose of the camera an rt_cam_ref); else the reference is the camera's first pose (Nposes-1 rt_cam_ref, the
    board's pose a frame that is optimized)"""
    from mrcal_amd.poseutils import compose_rt, invert_rt
    oi, _ = make_calibration_problem(api, Ncameras=1, Nframes=Nposes, lensmodel=lensmodel,
                                     object_width_n=10, object_height_n=10, seed=seed)
    rt_cam0_board = np.array(oi["rt_ref_frame"])          # monocular: the frames are the board in the camera
    idx = oi["indices_frame_camintrinsics_camextrinsics"]
    idxf, idxci = idx[:,0].copy(), idx[:,1].copy()
    if ref_frame0:
        oi["indices_frame_camintrinsics_camextrinsics"] = np.ascontiguousarray(np.column_stack((0*idxf, idxci, idxf)).astype(np.int32))
        oi["rt_cam_ref"]   = np.ascontiguousarray(rt_cam0_board)
        oi["rt_ref_frame"] = np.zeros((1,6))
        oi["do_optimize_frames"] = False
    else:
        oi["indices_frame_camintrinsics_camextrinsics"] = np.ascontiguousarray(np.column_stack((0*idxf, idxci, idxf - 1)).astype(np.int32))
        oi["rt_cam_ref"]   = np.ascontiguousarray(compose_rt(rt_cam0_board[1:], invert_rt(rt_cam0_board[0])))
        oi["rt_ref_frame"] = np.ascontiguousarray(rt_cam0_board[:1])
        oi["do_optimize_frames"] = True
    oi["do_optimize_extrinsics"] = True
    return oi


class eliminating:
    """mrcal_amd_set_elimination() for the problems created inside (None: the library's own choice)"""
    def __init__(self, what): self.what = what
    def __enter__(self):
        import mrcal_amd
        self.f = mrcal_amd._api.clib.mrcal_amd_set_elimination
        self.old_env = os.environ.pop("MRCAL_AMD_ELIMINATE", None)
        self.old = self.f({None: 0, "frames": 1, "extrinsics": 2}[self.what])
    def __exit__(self, *a):
        self.f(self.old)
        if self.old_env is not None: os.environ["MRCAL_AMD_ELIMINATE"] = self.old_env


def test_moving_camera_problem_is_the_monocular_one(ref_api):
    """the construction itself: the same pixels (x at the seed) as the monocular calibration it was made from"""
    for ref_frame0 in (False, True):
        oi0, _ = make_calibration_problem(ref_api, Ncameras=1, Nframes=7, lensmodel="LENSMODEL_OPENCV4",
                                          object_width_n=10, object_height_n=10, seed=3)
        oi = moving_camera_problem(ref_api, 7, ref_frame0)
        x0 = ref_api.optimizer_callback(**copy_inputs(oi0), no_jacobian=True, no_factorization=True)[1]
        x1 = ref_api.optimizer_callback(**copy_inputs(oi),  no_jacobian=True, no_factorization=True)[1]
        n = 2*7*100
        assert np.abs(x0[:n] - x1[:n]).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("ref_frame0", (False, True))
def test_callback_and_normal_equations(amd, ref_api, ref_frame0):
    from mrcal_amd.resident import Problem
    oi = moving_camera_problem(amd._api, 9, ref_frame0)
    oi["observations_board"][2,3,4,2] = -1.
    b_a, x_a, J_a, _ = amd.optimizer_callback(**copy_inputs(oi), no_factorization=True)
    b_r, x_r, J_r, _ = ref_api.optimizer_callback(**copy_inputs(oi), no_factorization=True)
    assert np.array_equal(J_a.indptr, J_r.indptr) and np.array_equal(J_a.indices, J_r.indices)
    assert relative_error(x_a, x_r).max() < 1e-6 and relative_error(J_a.data, J_r.data).max() < 1e-6
    seen = set()
    for what in ("frames", "extrinsics"):
        with eliminating(what), Problem(**copy_inputs(oi)) as p:
            ne = p.normal_equations()
            J, x = p.J(), p.x()
            d = p.gauss_newton_step()
        seen.add(ne["eliminates"])
        N, g = dense_normal(J, x)
        assert np.abs(blocks_to_dense(ne, p.Nstate) - N).max() < 1e-10*np.abs(N).max(), what
        assert np.abs(ne["g"] - g).max() < 1e-10*np.abs(g).max()
        assert abs(ne["norm2_x"] - x @ x) < 1e-10*(x @ x)
        # (gauge: with the frame free the whole rig can move; the step is that of the damped system then, so
        #  only the locked-frame problem is held to N d = -g)
        if ref_frame0: assert np.abs(N @ d + g).max() < 1e-6*max(np.abs(g).max(), 1.0), what
    assert seen == {"frames", "extrinsics"}
    # the camera block: 8 intrinsics + 2 warp (+ the frame's 6), against 6 more per pose of the camera
    with eliminating("extrinsics"), Problem(**copy_inputs(oi)) as p:
        assert p.normal_equations()["Nc"] == 8 + 2 + (0 if ref_frame0 else 6)
    with eliminating("frames"), Problem(**copy_inputs(oi)) as p:
        assert p.normal_equations()["Nc"] == 8 + 2 + 6*(9 if ref_frame0 else 8)


@pytest.mark.gpu
@pytest.mark.parametrize("Ncameras,Nframes", ((6, 4), (5, 3)))
def test_stationary_rig_with_few_frames_under_both_eliminations(amd, ref_api, Ncameras, Nframes):
    """A STATIONARY rig of many cameras seen in a handful of frames also has more extrinsics than frame variables,
    and the library eliminates its extrinsics by itself (test_the_choice). Unlike the moving camera's, this problem
    has a camera AT the reference (camera 0: no extrinsics, its observations belong to no eliminated block) and
    several cameras per frame. Callback against the reference, the block normal equations of either partition
    against JtJ, and the solve against the reference's mrcal_optimize() under both"""
    from mrcal_amd.resident import Problem
    oi, _ = make_calibration_problem(amd._api, Ncameras=Ncameras, Nframes=Nframes, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=10, object_height_n=10, seed=13)
    assert np.any(oi["indices_frame_camintrinsics_camextrinsics"][:,2] < 0)
    oi["observations_board"][1,2,3,2] = -1.
    b_a, x_a, J_a, _ = amd.optimizer_callback(**copy_inputs(oi), no_factorization=True)
    b_r, x_r, J_r, _ = ref_api.optimizer_callback(**copy_inputs(oi), no_factorization=True)
    assert np.array_equal(b_a, b_r)
    assert np.array_equal(J_a.indptr, J_r.indptr) and np.array_equal(J_a.indices, J_r.indices)
    assert relative_error(x_a, x_r).max() < 1e-6 and relative_error(J_a.data, J_r.data).max() < 1e-6
    with eliminating(None), Problem(**copy_inputs(oi)) as p:
        assert p.partition()["eliminates"] == "extrinsics"
    Nintr = 8*Ncameras
    for what in ("frames", "extrinsics"):
        with eliminating(what), Problem(**copy_inputs(oi)) as p:
            assert p.partition()["eliminates"] == what
            ne = p.normal_equations()
            J, x = p.J(), p.x()
            d = p.gauss_newton_step()
        N, g = dense_normal(J, x)
        assert np.abs(blocks_to_dense(ne, p.Nstate) - N).max() < 1e-10*np.abs(N).max(), what
        assert np.abs(ne["g"] - g).max() < 1e-10*np.abs(g).max()
        assert np.abs(N @ d + g).max() < 1e-6*max(np.abs(g).max(), 1.0), what
        # the dense block: intrinsics + warp + whichever poses are NOT eliminated; dims[4] ("Nie") = S_split
        assert ne["Nc"]  == Nintr + 2 + (6*(Ncameras-1) if what == "frames" else 6*Nframes)
        assert ne["Nie"] == ne["S_split"] == (Nintr + 6*(Ncameras-1) if what == "frames" else Nintr)
    oi["do_apply_outlier_rejection"] = True
    oi_r = copy_inputs(oi)
    s_r = ref_api.optimize(**oi_r)
    for what in ("frames", "extrinsics"):
        with eliminating(what):
            oi_a = copy_inputs(oi)
            s_a = amd.optimize(**oi_a)
        assert s_a["Noutliers_board"] == s_r["Noutliers_board"], what
        assert np.array_equal(oi_a["observations_board"][...,2] < 0, oi_r["observations_board"][...,2] < 0), what
        assert abs(s_a["rms_reproj_error__pixels"] - s_r["rms_reproj_error__pixels"]) < 1e-6*s_r["rms_reproj_error__pixels"], what
        assert np.abs(s_a["b_packed"] - s_r["b_packed"]).max() < 2e-5, what
        assert np.abs(s_a["x"] - s_r["x"]).max() < 1e-5, what


@pytest.mark.gpu
def test_the_choice(amd):
    """many camera poses and few frames: the extrinsics go; the stationary rigs of every other test: the frames"""
    from mrcal_amd.resident import Problem
    with eliminating(None):
        for ref_frame0 in (False, True):
            with Problem(**moving_camera_problem(amd._api, 9, ref_frame0)) as p:
                assert p.partition()["eliminates"] == "extrinsics"
        for Ncameras, Nframes in ((1, 9), (3, 11), (8, 20), (6, 4)):
            oi, _ = make_calibration_problem(amd._api, Ncameras=Ncameras, Nframes=Nframes, lensmodel="LENSMODEL_OPENCV4",
                                             object_width_n=10, object_height_n=10, seed=3)
            with Problem(**oi) as p:
                assert p.partition()["eliminates"] == ("extrinsics" if (Ncameras, Nframes) == (6, 4) else "frames")


@pytest.mark.gpu
@pytest.mark.parametrize("ref_frame0,Nposes,lensmodel", ((True, 12, "LENSMODEL_OPENCV4"), (False, 12, "LENSMODEL_OPENCV4"),
                                                         (True, 60, "LENSMODEL_OPENCV8")))
def test_solve(amd, ref_api, ref_frame0, Nposes, lensmodel):
    """the moving camera solved with its poses eliminated: the reference's optimum (checker: the reference's own
    mrcal_optimize() over the restated libdogleg), the frames-eliminated solve's optimum, the same bits twice"""
    from mrcal_amd.resident import Problem
    oi = moving_camera_problem(amd._api, Nposes, ref_frame0, lensmodel=lensmodel)
    oi["do_apply_outlier_rejection"] = True
    oi["observations_board"][1,2,4,:2] += 40.
    oi_r = copy_inputs(oi)
    s_r = ref_api.optimize(**oi_r)
    res = {}
    for what in ("extrinsics", "frames", "extrinsics"):
        with eliminating(what):
            oi_a = copy_inputs(oi)
            s_a = amd.optimize(**oi_a)
        assert s_a["Noutliers_board"] == s_r["Noutliers_board"] > 0, what
        assert np.array_equal(oi_a["observations_board"][...,2] < 0, oi_r["observations_board"][...,2] < 0), what
        assert abs(s_a["rms_reproj_error__pixels"] - s_r["rms_reproj_error__pixels"]) < 1e-6, what
        if ref_frame0:
            # (a free frame beside free cameras is a gauge: the optimum is a 6-dimensional family, and where on
            #  it a solver stops is its own business. With the frame locked the state itself is compared)
            assert np.abs(s_a["b_packed"] - s_r["b_packed"]).max() < 2e-5, what
        # (x in pixels x weight: 1e-6 of a pixel where the two solvers stopped)
        assert np.abs(s_a["x"] - s_r["x"]).max() < 1e-6*max(1.0, np.abs(s_r["x"]).max()), what
        if what in res:
            assert np.array_equal(res[what]["b_packed"], s_a["b_packed"]) and np.array_equal(res[what]["x"], s_a["x"])
        res[what] = s_a
    # the trial steps are cheaper with the small camera block (and not more numerous)
    with eliminating("extrinsics"), Problem(**copy_inputs(oi)) as p:
        assert p.partition()["eliminates"] == "extrinsics"
