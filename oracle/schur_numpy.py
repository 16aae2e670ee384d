"""TEST / BASELINE INFRASTRUCTURE ONLY: nothing under mrcal_amd/ imports this.

The product's linear algebra restated on the host with numpy + LAPACK, for the CPU baseline of bench.py (SURVEY.md
section 8d: "the reference callback under the same Schur-complement algorithm on CPU/numpy") and for the tests that
pin it to dense numpy: the Gauss-Newton step  (JtJ) d = -Jt x  of a chessboard calibration (mrcal.c:4603-4898 rows:
every measurement row touches at most ONE frame) by

    per-observation Grams  ->  A (camera block), Bt (frame x camera couplings), D (6x6 per frame), g
    L_f = chol(D_f)  (batched dpotrf), Wt_f = L_f^-1 Bt_f, y_f = L_f^-1 g_f
    S = A - sum_f Wt_f^T Wt_f,  r = g_S - sum_f Wt_f^T y_f
    d_S = -chol(S)^-1 r  (ONE dense dpotrf of the camera block: 140 x 140 at the metric's size)
    d_f = -L_f^-T (y_f + Wt_f d_S)

which is what csrc/assembly.hip, schur.hip, cholesky_lds.hip and step.hip do on the GPU (assemble_factor, schur_syrk, schur_cholesky_solve, backsub), and -
for the third baseline variant - the same blocks assembled into a scipy.sparse matrix for SuperLU (LU, NOT Cholesky).
Boards + regularization rows only (the benchmark's problem); frames eliminated."""
import numpy as np
import scipy.linalg
import scipy.sparse as sp


class BoardLayout:
    """What the block algorithm needs to know about the rows of J, from the layout functions of either library"""
    def __init__(self, api, oi):
        idx = oi["indices_frame_camintrinsics_camextrinsics"]
        self.Nobs    = idx.shape[0]
        self.iframe  = np.ascontiguousarray(idx[:, 0]).astype(np.int64)
        self.Npts    = oi["observations_board"].shape[1]*oi["observations_board"].shape[2]
        self.Nstate  = api.num_states(**oi)
        self.Nframes = oi["rt_ref_frame"].shape[0]
        self.i_frames = api.state_index_frames(0, **oi)
        assert self.i_frames is not None and api.num_states_frames(**oi) == 6*self.Nframes
        assert api.num_states_points(**oi) == 0 and api.num_measurements_points(**oi) == 0
        self.Nmeas_boards = api.num_measurements_boards(**oi)
        assert self.Nmeas_boards == 2*self.Npts*self.Nobs
        # S: every state variable that is not a frame's, in state order; E: the frames
        s_of_state = np.full(self.Nstate, -1, dtype=np.int64)
        notframe = np.ones(self.Nstate, dtype=bool)
        notframe[self.i_frames:self.i_frames + 6*self.Nframes] = False
        self.S_states = np.nonzero(notframe)[0]
        s_of_state[self.S_states] = np.arange(len(self.S_states))
        self.s_of_state = s_of_state
        self.Nc = len(self.S_states)


def normal_equations(J, x, lay):
    """A (Nc,Nc), Bt (Nframes,6,Nc), D (Nframes,6,6), gS (Nc), gE (Nframes,6) of JtJ, Jt x from the CSR Jacobian:
    a board observation's 2*Npts rows share two column patterns (its x rows and its y rows), so its rows are a
    dense (2*Npts, k) block of J.data and its Gram one small matrix product"""
    Nc, Nf, P = lay.Nc, lay.Nframes, lay.Npts
    A  = np.zeros((Nc, Nc)); Bt = np.zeros((Nf, 6, Nc)); D = np.zeros((Nf, 6, 6))
    gS = np.zeros(Nc); gE = np.zeros((Nf, 6))
    indptr, indices, data = J.indptr, J.indices, J.data
    row0 = 2*P*np.arange(lay.Nobs)
    k_of_obs = indptr[row0 + 1] - indptr[row0]
    for k in np.unique(k_of_obs):
        obs = np.nonzero(k_of_obs == k)[0]
        n0  = indptr[row0[obs]].astype(np.int64)
        V   = data[(n0[:, None] + np.arange(2*P*k)[None, :])].reshape(len(obs), 2*P, k)
        xo  = x[(row0[obs][:, None] + np.arange(2*P)[None, :])]
        cx  = indices[(n0[:, None] + np.arange(k)[None, :])].astype(np.int64)          # columns of the x rows
        cy  = indices[(n0[:, None] + k + np.arange(k)[None, :])].astype(np.int64)      # ... of the y rows
        for V_, c_, x_ in ((V[:, 0::2, :], cx, xo[:, 0::2]), (V[:, 1::2, :], cy, xo[:, 1::2])):
            G = np.matmul(V_.transpose(0, 2, 1), V_)                 # (obs, k, k)
            g = np.matmul(V_.transpose(0, 2, 1), x_[:, :, None])[:, :, 0]
            isf = (c_ >= lay.i_frames) & (c_ < lay.i_frames + 6*Nf)
            # the same positions of a row are frame columns in every observation of this class
            assert np.all(isf == isf[0]) and isf[0].sum() in (0, 6)
            fpos, spos = np.nonzero(isf[0])[0], np.nonzero(~isf[0])[0]
            s = lay.s_of_state[c_[:, spos]]                          # (obs, ks)
            np.add.at(A,  (s[:, :, None], s[:, None, :]), G[:, spos[:, None], spos[None, :]])
            np.add.at(gS, s, g[:, spos])
            if len(fpos):
                f = lay.iframe[obs]
                np.add.at(D,  f, G[:, fpos[:, None], fpos[None, :]])
                np.add.at(gE, f, g[:, fpos])
                np.add.at(Bt, (f[:, None, None], np.arange(6)[None, :, None], s[:, None, :]), G[:, fpos[:, None], spos[None, :]])
    # the rows behind the boards (regularization): camera-block columns only
    Jr = J[lay.Nmeas_boards:]
    if Jr.shape[0]:
        assert np.all(lay.s_of_state[Jr.indices] >= 0)
        Jr = sp.csr_matrix((Jr.data, lay.s_of_state[Jr.indices], Jr.indptr), shape=(Jr.shape[0], Nc))
        A  += (Jr.T @ Jr).toarray()
        gS += Jr.T @ x[lay.Nmeas_boards:]
    return A, Bt, D, gS, gE


def _damp(A, D, mu):
    if mu > 0.0:
        A = A + mu*np.eye(A.shape[0]); D = D + mu*np.eye(6)[None]
    return A, D


def gauss_newton_step_schur(J, x, lay, mu=0.0):
    """d (Nstate) with (JtJ + mu I) d = -Jt x, by block elimination of the frames + one dense Cholesky"""
    A, Bt, D, gS, gE = normal_equations(J, x, lay)
    A, D = _damp(A, D, mu)
    L  = np.linalg.cholesky(D)                                       # batched dpotrf
    Li = np.linalg.inv(L)                                            # 6x6 triangles: cheaper than a batched gesv of 140 columns
    Wt = np.matmul(Li, Bt)                                           # L^-1 Bt   (Nf,6,Nc)
    y  = np.matmul(Li, gE[:, :, None])[:, :, 0]
    W2 = Wt.reshape(-1, lay.Nc)
    S  = A - W2.T @ W2
    r  = gS - W2.T @ y.reshape(-1)
    c  = scipy.linalg.cho_factor(S, lower=True)                      # dpotrf
    dS = -scipy.linalg.cho_solve(c, r)
    dE = -np.matmul(Li.transpose(0, 2, 1), (y + Wt @ dS)[:, :, None])[:, :, 0]
    d  = np.zeros(lay.Nstate)
    d[lay.S_states] = dS
    d[lay.i_frames:lay.i_frames + 6*lay.Nframes] = dE.reshape(-1)
    return d


def gauss_newton_step_superlu(J, x, lay, mu=0.0):
    """the same blocks assembled into one sparse matrix and handed to scipy.sparse.linalg.splu (SuperLU: an LU
    factorization with partial pivoting, NOT a Cholesky - the SciPy solver SURVEY.md 8d names)"""
    import scipy.sparse.linalg
    A, Bt, D, gS, gE = normal_equations(J, x, lay)
    A, D = _damp(A, D, mu)
    Nf, Nc = lay.Nframes, lay.Nc
    B  = sp.csr_matrix(Bt.reshape(6*Nf, Nc))
    Dm = sp.block_diag([D[f] for f in range(Nf)], format="csr") if Nf < 64 else \
         sp.bsr_matrix((D, np.arange(Nf), np.arange(Nf + 1)), shape=(6*Nf, 6*Nf)).tocsr()
    N  = sp.bmat([[sp.csr_matrix(A), B.T], [B, Dm]], format="csc")
    lu = scipy.sparse.linalg.splu(N)
    sol = -lu.solve(np.concatenate((gS, gE.reshape(-1))))
    d = np.zeros(lay.Nstate)
    d[lay.S_states] = sol[:Nc]
    d[lay.i_frames:lay.i_frames + 6*Nf] = sol[Nc:]
    return d


def with_state(api, oi, b_packed, copy_inputs):
    """a copy of the inputs with the packed state b written into the arrays a callback reads (the inverse of the
    packing of mrcal.c:5990-6148, through the library's own unpack_state() and state_index_*())"""
    o = copy_inputs(oi)
    b = np.array(b_packed, dtype=float)
    api.unpack_state(b, **o)
    core  = bool(o.get("do_optimize_intrinsics_core", True))
    dist  = bool(o.get("do_optimize_intrinsics_distortions", True))
    if core or dist:
        Nopt = api.num_intrinsics_optimization_params(**o)
        for i in range(o["intrinsics"].shape[0]):
            i0 = api.state_index_intrinsics(i, **o)
            if i0 is None: continue
            dst = o["intrinsics"][i, (0 if core else 4):(None if dist else 4)]
            assert dst.size == Nopt
            dst[:] = b[i0:i0+Nopt]
    if o.get("do_optimize_extrinsics", True):
        for i in range(o["rt_cam_ref"].shape[0]):
            i0 = api.state_index_extrinsics(i, **o)
            if i0 is not None: o["rt_cam_ref"].reshape(-1,6)[i] = b[i0:i0+6]
    if o.get("do_optimize_frames", True):
        for i in range(o["rt_ref_frame"].shape[0]):
            i0 = api.state_index_frames(i, **o)
            if i0 is not None: o["rt_ref_frame"].reshape(-1,6)[i] = b[i0:i0+6]
    if o.get("do_optimize_calobject_warp", False) and o.get("calobject_warp") is not None:
        i0 = api.state_index_calobject_warp(**o)
        if i0 is not None: o["calobject_warp"][:] = b[i0:i0+2]
    return o


def timed_trial_steps(api, oi, Nsteps, solver, copy_inputs):
    """Nsteps trial steps on the host, each EXACTLY the metric's unit (one residual + Jacobian evaluation
    and one normal-equation solve): the step from the current point by `solver` (gauss_newton_step_schur / _superlu)
    with the current damping, then one optimizer_callback() of `api` (the reference's) at the trial point. Levenberg's
    rule stands in for the trust region: a trial whose cost went up is rejected and the damping raised (x10, from
    1e-3 of the mean diagonal), an accepted one lowers it (/10, to 0 below 1e-6 of it) - so a rejected trial is followed
    by a NEW solve, as a dog-leg step with a smaller radius would reuse the old one: this loop never solves less often
    than the product does. Returns dict(seconds, seconds_callback, seconds_solve, Ntrials, Nsolves, cost0, cost1)"""
    import time
    lay = BoardLayout(api, oi)
    t_cb = t_sv = 0.0
    t0 = time.perf_counter()
    b, x, J, _ = api.optimizer_callback(no_factorization=True, **copy_inputs(oi))
    cost = cost0 = float(x @ x)
    scale = float((J.multiply(J)).sum())/J.shape[1]                 # mean diagonal of JtJ
    mu = 0.0
    Ntrials = Nsolves = 0
    while Ntrials < Nsteps:
        t = time.perf_counter(); d = solver(J, x, lay, mu); t_sv += time.perf_counter() - t
        Nsolves += 1
        t = time.perf_counter()
        bt, xt, Jt, _ = api.optimizer_callback(no_factorization=True, **with_state(api, oi, b + d, copy_inputs))
        t_cb += time.perf_counter() - t
        Ntrials += 1
        ct = float(xt @ xt)
        if ct < cost:
            b, x, J, cost = bt, xt, Jt, ct
            mu = mu/10.0 if mu > 1e-6*scale else 0.0
        else:
            mu = max(10.0*mu, 1e-3*scale)
    return dict(seconds=time.perf_counter() - t0, seconds_callback=t_cb, seconds_solve=t_sv, Ntrials=Ntrials,
                Nsolves=Nsolves, cost0=cost0, cost1=cost)
