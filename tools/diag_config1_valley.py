"""Why two converged solves of BASELINE configuration 1 (4 cameras x 400 frames OPENCV8) sit a few 1e-4 packed
units apart (tests/test_full_size.py::test_solve_matches_reference_at_baseline_size): CPU only, the reference's
own mrcal_optimize() + callback (oracle/_ref). From the reference's returned state, exact Gauss-Newton steps
(scipy sparse LU on JtJ) keep moving one high-order distortion coefficient by 1.6e-4..1.8e-4 per step while the
cost falls by 5e-9, 2e-9, 1e-9 of 704242.95 (7e-15 relative: below what the dog-leg's gain ratio can resolve in
FP64): Gauss-Newton on a problem with 1.5-pixel residuals converges LINEARLY along the flattest directions
(smallest eigenvalues of JtJ 0.13, largest 4.9e10), so every GN-based solver stops where the cost change drowns
in rounding, not at the stationary point. Output of the last run:
  0 cost 704242.952464425  |g| 0.00528 |d|max 0.000160 at 45
  1 cost 704242.9524644196 |g| 0.00293 |d|max 0.000168 at 45
  2 cost 704242.9524644173 |g| 0.00338 |d|max 0.000174 at 45
  3 cost 704242.9524644162 |g| 0.00325 |d|max 0.000182 at 45
  eig min/max [0.131 0.139 0.167 1.29 3.51] 4.88e10
"""
import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
import scipy.sparse.linalg as spla
ref = Api(MrcalLib('/root/repo/oracle/_ref/libmrcal_ref.so'))
oi,_ = make_calibration_problem(ref, object_width_n=10, object_height_n=10, seed=2, Ncameras=4,Nframes=400,lensmodel="LENSMODEL_OPENCV8")
o = copy_inputs(oi); s = ref.optimize(**o)
o["do_apply_outlier_rejection"]=False
for it in range(4):
    b,x,J,_ = ref.optimizer_callback(no_factorization=True, **o)
    N = (J.T@J).tocsc(); g = J.T@x
    d = -spla.spsolve(N, g)
    print(it, "cost", x@x, "|g|", np.abs(g).max(), "|d|max", np.abs(d).max(), np.abs(d).argmax())
    bn = b + d
    ref.unpack_state(bn, **o)
    # write back
    Ni = 12
    o["intrinsics"][:] = bn[:48].reshape(4,12)
    o["rt_cam_ref"][:] = bn[48:66].reshape(3,6)
    o["rt_ref_frame"][:] = bn[66:66+2400].reshape(400,6)
    o["calobject_warp"][:] = bn[-2:]
Nd = N.toarray()
w = np.linalg.eigvalsh(Nd)
print("eig min/max", w[:5], w[-1])
