#!/usr/bin/env python3
"""Randomized parity sweep (dev tool, GPU box): random small calibration problems - lens model, cameras, frames,
board size, which blocks are optimized, discrete points, outliers on input - through mrcal_amd and through the
reference's own code (oracle/_ref): callback (b_packed and CSR structure bit-exact, x and J to 1e-6), then the solve
(same outliers, rms to 1e-5 relative; where they differ, both solutions are solved again by BOTH solvers: the same
numbers from the same starting point mean the first solves differed by their path, not by the implementation).
Prints one line per case and a summary; exit code 1 on any mismatch.

    python tools/fuzz_parity.py [Ncases] [seed] [--callbacks-only]

board_cases() is the generator of the board/point problems, importable (tools/diag_splined_pd.py and the tests
rebuild a case of a sweep from its number and the sweep's seed with it; it needs no GPU when given the reference's API).
"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from test_callback_parity import compare_callbacks, _with_points

MODELS = ("LENSMODEL_PINHOLE", "LENSMODEL_STEREOGRAPHIC", "LENSMODEL_OPENCV4", "LENSMODEL_OPENCV5", "LENSMODEL_OPENCV8",
          "LENSMODEL_OPENCV12", "LENSMODEL_CAHVOR", "LENSMODEL_CAHVORE_linearity=0.37",
          "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120")


def board_cases(N, rng, api):
    """The board/point problems of a sweep, in order: (icase, what, oi, splined, with_points, sel, W, H, Nf).
    Every draw from rng happens here, whatever the caller does with a case"""
    for icase in range(N):
        lens  = MODELS[rng.randint(len(MODELS))]
        Ncam  = int(rng.randint(1, 5)); Nf = int(rng.randint(2, 13))
        W, H  = int(rng.randint(3, 11)), int(rng.randint(3, 11))
        oi, _ = make_calibration_problem(api, Ncameras=Ncam, Nframes=Nf, lensmodel=lens,
                                         object_width_n=W, object_height_n=H, seed=int(rng.randint(1 << 30)))
        splined = "SPLINED" in lens
        sel = dict(do_optimize_intrinsics_core        = bool(rng.rand() < 0.7) and not splined,
                   do_optimize_intrinsics_distortions = bool(rng.rand() < 0.7) and oi["intrinsics"].shape[1] > 4,
                   do_optimize_extrinsics             = bool(rng.rand() < 0.8) and Ncam > 1,
                   do_optimize_frames                 = bool(rng.rand() < 0.85),
                   do_optimize_calobject_warp         = bool(rng.rand() < 0.6),
                   do_apply_regularization            = bool(rng.rand() < 0.7))
        if not any(sel[k] for k in list(sel)[:5]): sel["do_optimize_frames"] = True
        oi.update(sel)
        if not sel["do_optimize_calobject_warp"] and rng.rand() < 0.5: oi["calobject_warp"] = None
        with_points = rng.rand() < 0.35 and not splined
        if with_points:
            oi = _with_points(oi, rng, Npoints=int(rng.randint(3, 9)), Npoints_fixed=int(rng.randint(0, 3)))
        if rng.rand() < 0.5:
            oi["observations_board"][rng.randint(oi["observations_board"].shape[0]), rng.randint(H), rng.randint(W), 2] = -1.
        what = f"case {icase}: {lens.replace('LENSMODEL_','')[:22]} {Ncam} cam {Nf} fr {W}x{H} " + \
               "".join(c for c, k in zip("cdefwr", sel) if sel[k]) + (" +points" if with_points else "")
        yield icase, what, oi, splined, with_points, sel, W, H, Nf


def well_posed_for_a_solve(api, oi, splined, with_points, sel, W, H, Nf):
    """The solve is compared only where the problem is well posed enough for two solvers to be expected at the same
    point: enough data per unknown, the distortions regularized, no discrete points at made-up pixels (a point seen
    once has no depth: a singular 3x3 block; seen twice at random pixels it sits wherever its rays happen to pass),
    and at least three board measurements per unknown (eight for a splined surface, whose knots away from the boards
    no measurement sees): a splined surface of 88 knots fitted to five small boards is held by its regularization
    alone, and where two dog-leg implementations run out of iterations on that plateau says nothing about either"""
    Nstate = api.num_states(**oi)
    return (not with_points) and W*H >= 30 and Nf >= 5 and \
           (sel["do_apply_regularization"] or not sel["do_optimize_intrinsics_distortions"]) and \
           oi["observations_board"].size//3*2 >= (8 if splined else 3)*Nstate


def compare_solves(ref, oi, what, tally):
    """Both solves of one problem; where they differ each solution is solved again by BOTH solvers (no outlier
    rejection): a solution that either solver improves further was not a stationary point. Returns (ok, message)"""
    oa, orr = copy_inputs(oi), copy_inputs(oi)
    sa, sr = mrcal_amd.optimize(**oa), ref.optimize(**orr)
    r_a, r_r = sa["rms_reproj_error__pixels"], sr["rms_reproj_error__pixels"]
    if sa["Noutliers_board"] == sr["Noutliers_board"] and abs(r_a - r_r) < 1e-5*r_r:
        return True, ""
    def again(api, o):
        o = copy_inputs(o); o["do_apply_outlier_rejection"] = False
        return api.optimize(**o)["rms_reproj_error__pixels"]
    r_aa, r_ra = again(mrcal_amd, oa), again(ref, oa)      # from OUR solution
    r_ar, r_rr = again(mrcal_amd, orr), again(ref, orr)    # from the REFERENCE's solution
    msg = (f"SOLVE DIFFERS: rms {r_a:.9g} vs {r_r:.9g}, outliers {sa['Noutliers_board']} vs {sr['Noutliers_board']}; "
           f"again from ours: ours {r_aa:.9g} ref {r_ra:.9g}; again from the reference's: ours {r_ar:.9g} ref {r_rr:.9g}")
    # the same behaviour from the same starting point = the first solves differed by their PATH (both stop at the
    # iteration limit or where the gain ratio drowns in rounding on these small, often unregularized problems)
    alike = abs(r_aa - r_ra) < 2e-3*r_ra and abs(r_ar - r_rr) < 2e-3*r_rr
    # ... or the checker's dog-leg (a restatement of libdogleg, oracle/dogleg_restated.c) gave up on a flat stretch:
    # our solution is lower, NEITHER solver moves away from it, and BOTH improve the checker's
    early = (not alike) and r_a < r_r and abs(r_aa - r_a) < 1e-6*r_a and abs(r_ra - r_a) < 1e-6*r_a and \
            r_ar < r_r*(1 - 1e-4) and r_rr < r_r*(1 - 1e-4)
    tally["path"] += alike; tally["early"] += early
    msg = ("path-dependent, the two solvers alike from either solution: " if alike else
           "the checker stopped early (ours is stationary for both solvers, its own is improved by both): " if early else "") + msg
    return (alike or early), msg


def run(N, seed, callbacks_only=False, out=print):
    """The sweep. callbacks_only: no solves (what tests/test_fuzz_parity.py runs). Returns the number of mismatches"""
    from test_triangulated import sfm_problem, compare_callbacks_with_pairs
    from test_moving_camera import moving_camera_problem
    from mrcal_amd.resident import Problem
    ref   = Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")))
    rng   = np.random.RandomState(seed)
    tally = dict(path=0, early=0)
    bad = 0; nsolved = 0
    for icase, what, oi, splined, with_points, sel, W, H, Nf in board_cases(N, rng, mrcal_amd._api):
        try:
            compare_callbacks(mrcal_amd.optimizer_callback(no_factorization=True, **copy_inputs(oi)),
                              ref.optimizer_callback(no_factorization=True, **copy_inputs(oi)), what)
            if callbacks_only or not well_posed_for_a_solve(mrcal_amd, oi, splined, with_points, sel, W, H, Nf):
                out(what, "ok (callback only)", flush=True)
                continue
            nsolved += 1
            ok, msg = compare_solves(ref, oi, what, tally)
            out(what, "ok" if not msg else msg, flush=True)
            bad += not ok
        except Exception as e:
            bad += 1
            out(what, "FAILED:", str(e).splitlines()[0][:200], flush=True)
    # structure-from-motion shapes: triangulated points (+ board frames), intrinsics locked (mrcal.c:6043-6051). The pairs'
    # rows are held to the rounding envelope of their formula (tests/test_triangulated.py compare_callbacks_with_pairs)
    Nsfm = max(N//5, 4)
    for icase in range(Nsfm):
        lens = ("LENSMODEL_PINHOLE", "LENSMODEL_OPENCV4")[rng.randint(2)]
        Ncam = int(rng.randint(2, 6)); Np = int(rng.randint(6, 80)); Nbf = int(rng.randint(0, 5))
        oi, _ = sfm_problem(lens, Ncam=Ncam, Npoints=Np, seed=int(rng.randint(1 << 30)), noise=float(rng.uniform(0.1, 2.0)),
                            Nboard_frames=Nbf, board_wh=(int(rng.randint(3, 8)), int(rng.randint(3, 8))))
        oi["do_apply_regularization_unity_cam01"] = bool(rng.rand() < 0.6)
        what = f"sfm case {icase}: {lens.replace('LENSMODEL_','')} {Ncam} cam {Np} points {Nbf} board frames unity={oi['do_apply_regularization_unity_cam01']}"
        try:
            m0 = mrcal_amd.measurement_index_points_triangulated(**oi)
            compare_callbacks_with_pairs(mrcal_amd.optimizer_callback(no_factorization=True, **copy_inputs(oi)),
                                         ref.optimizer_callback(no_factorization=True, **copy_inputs(oi)),
                                         m0, m0 + mrcal_amd.num_measurements_points_triangulated(**oi), what)
            out(what, "ok (callback only)", flush=True)
        except Exception as e:
            bad += 1
            out(what, "FAILED:", str(e).splitlines()[0][:200], flush=True)
    N += Nsfm
    # a moving camera (tests/test_moving_camera.py, the reference's _apply_moving_ref): the library eliminates the
    # extrinsics where it finds more of them than frame variables; callback, and the solve against the reference's
    Nmov = max(N//8, 4)
    for icase in range(Nmov):
        lens = MODELS[rng.randint(len(MODELS) - 1)]             # (not the splined one: that keeps the frames' elimination)
        Nposes = int(rng.randint(3, 40)); ref_frame0 = bool(rng.rand() < 0.5)
        oi = moving_camera_problem(mrcal_amd._api, Nposes, ref_frame0, seed=int(rng.randint(1 << 30)), lensmodel=lens)
        oi["do_optimize_calobject_warp"] = bool(rng.rand() < 0.7)
        oi["do_optimize_intrinsics_core"] = bool(rng.rand() < 0.8)
        if rng.rand() < 0.5: oi["observations_board"][rng.randint(Nposes), rng.randint(10), rng.randint(10), 2] = -1.
        with Problem(**copy_inputs(oi)) as p: eliminates = p.partition()["eliminates"]
        what = f"moving camera case {icase}: {lens.replace('LENSMODEL_','')[:22]} {Nposes} poses ref_frame0={ref_frame0} eliminates {eliminates}"
        try:
            compare_callbacks(mrcal_amd.optimizer_callback(no_factorization=True, **copy_inputs(oi)),
                              ref.optimizer_callback(no_factorization=True, **copy_inputs(oi)), what)
            if callbacks_only or Nposes < 6: out(what, "ok (callback only)", flush=True); continue
            nsolved += 1
            ok, msg = compare_solves(ref, oi, what, tally)
            out(what, "ok" if not msg else msg, flush=True)
            bad += not ok
        except Exception as e:
            bad += 1
            out(what, "FAILED:", str(e).splitlines()[0][:200], flush=True)
    N += Nmov
    out(f"{N - bad}/{N} cases agree: every callback; {nsolved} solves compared ({tally['path']} of them: solves that end on "
        f"different points of the same valley, both solvers alike when restarted; {tally['early']}: the checker stopped short "
        "of a stationary point that both solvers then reach))")
    return bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sys.exit(1 if run(int(args[0]) if len(args) > 0 else 40, int(args[1]) if len(args) > 1 else 0,
                      "--callbacks-only" in sys.argv) else 0)
