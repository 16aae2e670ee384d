#!/usr/bin/env python3
"""Generates tests/golden/optimizer_callback_golden.npz from the reference's OWN
test fixtures. Runs only where /root/reference exists (the dev container); the
resulting .npz is committed, so the tests never need /root/reference.

What goes in:

  inputs     exactly what test/test-optimizer-callback.py:43-88 assembles: 2
             OPENCV8 cameras (test/data/cam{0,1}.opencv8.cameramodel), 4 board
             observations (rows 1,2,4,5 of
             test/data/synthetic-board-observations.vnl), 5 point observations,
             3 points, 3 frames, warp (1e-3,2e-3)
  x_ref_N    the reference's shipped golden vectors
  J_ref_N    test/data/test-optimizer-callback-ref-{x,J}-N.npy for the six
             do_optimize_* cases (:90-131). J is dense and in UNPACKED units,
             as that test stores it
  x_lib_N, J_lib_N
             the same quantities recomputed NOW by the reference's own C
             sources compiled as oracle/_ref/libmrcal_ref.so. They are stored
             because the shipped goldens of cases 0,1,3 predate the reference's
             current regularization scales (mrcal.c:5703-5713,5789-5791,
             5859-5860): their regularization rows are stale, while all 810
             observation rows agree. See SURVEY.md section 8c.

Usage: python3 tests/golden/make_golden.py
"""
import ast
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF  = os.environ.get("MRCAL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api  import Api


def read_cameramodel(path):
    with open(path) as f:
        d = ast.literal_eval(f.read())
    return d


def read_vnl_corners(path, W, H):
    """filename x y level -> dict filename -> (H,W,3) with weight = 2^-level,
    or -1 for undetected ('-')"""
    out = {}
    order = []
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            fields = line.split()
            name = fields[0]
            if name not in out:
                out[name] = []
                order.append(name)
            if fields[1] == "-" or fields[3] == "-" or float(fields[3]) < 0:
                out[name].append((-1., -1., -1.))
            else:
                out[name].append((float(fields[1]), float(fields[2]), 1./(1 << int(float(fields[3])))))
    return order, {k: np.array(v).reshape(H,W,3) for k,v in out.items()}


def linspace_shaped(*shape):
    return np.linspace(0, 1, int(np.prod(shape))).reshape(*shape)


def main():
    reflib_path = os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")
    reflib = MrcalLib(reflib_path)
    api    = Api(reflib)

    # pose utilities of the reference, only needed to assemble the inputs
    lib = reflib.lib
    dp = C.POINTER(C.c_double)
    def invert_rt(rt):
        out = np.zeros(6)
        lib.mrcal_invert_rt_full(out.ctypes.data_as(dp), 0, None,0,0, None,0,0,
                                 rt.ctypes.data_as(dp), 0)
        return out
    def compose_rt(a, b):
        out = np.zeros(6)
        lib.mrcal_compose_rt_full(out.ctypes.data_as(dp), 0,
                                  None,0,0, None,0,0, None,0,0, None,0,0, None,0,0, None,0,0,
                                  a.ctypes.data_as(dp), 0, b.ctypes.data_as(dp), 0,
                                  False, False)
        return out

    m0 = read_cameramodel(f"{REF}/test/data/cam0.opencv8.cameramodel")
    m1 = read_cameramodel(f"{REF}/test/data/cam1.opencv8.cameramodel")
    assert m0["lensmodel"] == m1["lensmodel"] == "LENSMODEL_OPENCV8"
    intrinsics  = np.array((m0["intrinsics"], m1["intrinsics"]), dtype=float)
    imagersizes = np.array((m0["imagersize"], m1["imagersize"]), dtype=np.int32)
    rt_cam_ref  = compose_rt(np.array(m1["extrinsics"], dtype=float),
                             invert_rt(np.array(m0["extrinsics"], dtype=float))).reshape(1,6)

    order, corners = read_vnl_corners(f"{REF}/test/data/synthetic-board-observations.vnl", 10, 10)
    # frame*-cam0.xxx, frame*-cam1.xxx sorted by (frame,camera)
    import re
    rows = []
    for name in order:
        m = re.match(r"frame(\d+)-cam(\d+)\.xxx", name)
        rows.append((int(m.group(1)), int(m.group(2)), name))
    rows.sort()
    observations = np.array([corners[r[2]] for r in rows])
    idx = np.zeros((len(rows),3), dtype=np.int32)
    idx[:,0] = [r[0] for r in rows]
    idx[:,1] = [r[1] for r in rows]
    idx[:,2] = idx[:,1] - 1
    sel = (1,2,4,5)
    observations = np.ascontiguousarray(observations[sel, ...])
    idx          = np.ascontiguousarray(idx[sel, ...])

    rt_ref_frame = linspace_shaped(3,6)
    rt_ref_frame[:,5] += 5
    idx_point = np.array(((0,1,-1), (1,0,-1), (1,1,0), (2,0,-1), (2,1,0)), dtype=np.int32)
    points = 10. + 2.*linspace_shaped(3,3)
    observations_point = np.concatenate((1000. + 500.*linspace_shaped(5,2),
                                         np.array((0.9, 0.8, 0.9, 1.3, 1.8))[:,None]), axis=-1)
    calobject_warp = np.array((1e-3, 2e-3))

    cases = (dict(do_optimize_intrinsics_core=False, do_optimize_intrinsics_distortions=True,
                  do_optimize_extrinsics=False, do_optimize_frames=False,
                  do_optimize_calobject_warp=False, do_apply_regularization=True),
             dict(do_optimize_intrinsics_core=True,  do_optimize_intrinsics_distortions=False,
                  do_optimize_extrinsics=False, do_optimize_frames=False,
                  do_optimize_calobject_warp=False, do_apply_regularization=True),
             dict(do_optimize_intrinsics_core=False, do_optimize_intrinsics_distortions=False,
                  do_optimize_extrinsics=False, do_optimize_frames=True,
                  do_optimize_calobject_warp=False, do_apply_regularization=True),
             dict(do_optimize_intrinsics_core=True,  do_optimize_intrinsics_distortions=True,
                  do_optimize_extrinsics=False, do_optimize_frames=True,
                  do_optimize_calobject_warp=False, do_apply_regularization=True),
             dict(do_optimize_intrinsics_core=True,  do_optimize_intrinsics_distortions=True,
                  do_optimize_extrinsics=True,  do_optimize_frames=True,
                  do_optimize_calobject_warp=True,  do_apply_regularization=False),
             dict(do_optimize_intrinsics_core=True,  do_optimize_intrinsics_distortions=True,
                  do_optimize_extrinsics=True,  do_optimize_frames=True,
                  do_optimize_calobject_warp=True,  do_apply_regularization=False,
                  outlier_indices=(1,2)))

    out = dict(intrinsics=intrinsics, imagersizes=imagersizes, rt_cam_ref=rt_cam_ref,
               rt_ref_frame=rt_ref_frame, points=points,
               observations_board=observations,
               indices_frame_camintrinsics_camextrinsics=idx,
               observations_point=observations_point,
               indices_point_camintrinsics_camextrinsics=idx_point,
               calobject_warp=calobject_warp,
               Ncases=len(cases))

    for icase, case in enumerate(cases):
        case = dict(case)
        obs = observations.copy()
        outliers = case.pop("outlier_indices", ())
        for i in outliers:
            obs.reshape(-1,3)[i,2] = -1.
        out[f"outlier_indices_{icase}"] = np.array(outliers, dtype=np.int32)
        for k,v in case.items():
            out[f"{k}_{icase}"] = v

        optimization_inputs = dict(
            intrinsics=intrinsics, rt_cam_ref=rt_cam_ref, rt_ref_frame=rt_ref_frame, points=points,
            observations_board=obs, indices_frame_camintrinsics_camextrinsics=idx,
            observations_point=observations_point, indices_point_camintrinsics_camextrinsics=idx_point,
            lensmodel="LENSMODEL_OPENCV8", calobject_warp=calobject_warp, imagersizes=imagersizes,
            calibration_object_spacing=0.1, verbose=False, **case)
        b, x, J, _ = api.optimizer_callback(no_factorization=True, **optimization_inputs)
        J = J.toarray()
        api.pack_state(J, **optimization_inputs)  # unpacked units, as the goldens store it

        x_ref = np.load(f"{REF}/test/data/test-optimizer-callback-ref-x-{icase}.npy")
        J_ref = np.load(f"{REF}/test/data/test-optimizer-callback-ref-J-{icase}.npy")
        out[f"x_ref_{icase}"] = x_ref
        out[f"J_ref_{icase}"] = J_ref
        out[f"x_lib_{icase}"] = x
        out[f"J_lib_{icase}"] = J
        out[f"b_lib_{icase}"] = b

        Nobs_rows = 4*100*2 + 5*2
        dx = np.abs(x - x_ref)
        print(f"case {icase}: x shape {x.shape}, J shape {J.shape}; "
              f"observation rows max|x-x_ref| = {dx[:Nobs_rows].max():.3g}, "
              f"all rows = {dx.max():.3g}; "
              f"max|J-J_ref| obs rows = {np.abs(J-J_ref)[:Nobs_rows].max():.3g}")

    path = os.path.join(HERE, "optimizer_callback_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KB)")


if __name__ == "__main__":
    main()
