import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFLIB_PATH = os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")
GOLDEN_DIR  = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def amd():
    """the product: the mrcal_amd package (libmrcal_amd.so underneath)"""
    import mrcal_amd
    return mrcal_amd


@pytest.fixture(scope="session")
def amd_api(amd):
    return amd._api


@pytest.fixture(scope="session")
def ref_api():
    """CHECKER: the reference's own C sources compiled as a CPU library
    (oracle/_ref, built by oracle/Makefile), driven through the same ctypes
    binding as the product"""
    if not os.path.exists(REFLIB_PATH):
        pytest.skip("oracle/_ref/libmrcal_ref.so is not built (make -C oracle)")
    from mrcal_amd._cabi import MrcalLib
    from mrcal_amd._api  import Api
    return Api(MrcalLib(REFLIB_PATH))


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN_DIR, "optimizer_callback_golden.npz"))


def golden_case_inputs(golden, icase):
    """optimization_inputs of case icase of the reference's
    test/test-optimizer-callback.py"""
    obs = golden["observations_board"].copy()
    for i in golden[f"outlier_indices_{icase}"]:
        obs.reshape(-1,3)[i,2] = -1.
    kw = dict(intrinsics   = golden["intrinsics"].copy(),
              rt_cam_ref   = golden["rt_cam_ref"].copy(),
              rt_ref_frame = golden["rt_ref_frame"].copy(),
              points       = golden["points"].copy(),
              observations_board = obs,
              indices_frame_camintrinsics_camextrinsics = golden["indices_frame_camintrinsics_camextrinsics"].copy(),
              observations_point = golden["observations_point"].copy(),
              indices_point_camintrinsics_camextrinsics = golden["indices_point_camintrinsics_camextrinsics"].copy(),
              lensmodel    = "LENSMODEL_OPENCV8",
              calobject_warp = golden["calobject_warp"].copy(),
              imagersizes  = golden["imagersizes"].copy(),
              calibration_object_spacing = 0.1,
              verbose = False)
    for k in ("do_optimize_intrinsics_core", "do_optimize_intrinsics_distortions",
              "do_optimize_extrinsics", "do_optimize_frames",
              "do_optimize_calobject_warp", "do_apply_regularization"):
        kw[k] = bool(golden[f"{k}_{icase}"])
    return kw


def relative_error(a, b, eps=1e-6):
    """the reference's own notion of relative error: testutils.py:105"""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.abs(a-b) / ((np.abs(a)+np.abs(b))/2. + eps)
