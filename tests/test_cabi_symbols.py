"""The C-ABI library loads and exports every entry point include/mrcal_amd.h
declares. No compute calls: runs without a GPU."""
import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mrcal_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(_?mrcal_[A-Za-z0-9_]+)\s*\(", text))
    # typedef'd struct names are not functions
    return sorted(n for n in names if not n.endswith("_t"))


def test_header_declares_something():
    names = declared_functions()
    assert "mrcal_optimize" in names
    assert "mrcal_optimizer_callback" in names
    assert "mrcal_amd_problem_create" in names
    assert len(names) > 40


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "mrcal_amd", "libmrcal_amd.so"))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/mrcal_amd.h but not exported: {missing}"


def test_no_gpu_means_loud_failure(amd):
    """the product has no CPU fallback"""
    import numpy as np
    import pytest
    if amd.gpu_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        amd.optimizer_callback(intrinsics = np.array(((1000.,1000.,500.,500.),)),
                               lensmodel = "LENSMODEL_PINHOLE",
                               imagersizes = np.array(((1000,1000),), dtype=np.int32),
                               rt_ref_frame = np.array(((0.,0,0,0,0,2.),)),
                               observations_board = np.ones((1,3,3,3)),
                               indices_frame_camintrinsics_camextrinsics = np.array(((0,0,-1),), dtype=np.int32),
                               calibration_object_spacing = 0.1,
                               do_optimize_calobject_warp = False)
