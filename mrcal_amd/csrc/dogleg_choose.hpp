// The dog-leg step selection of the device-controlled solver step, as device functions: used by
// step2_choose_kernel (step.hip) and, where an evaluation with a board prologue follows, by that
// prologue's launch (kernels.hip board_prologue_kernel<CHOOSE>): one launch less per trial step.
//
// Every workgroup that calls dogleg_choose_scalars() derives the same numbers from the same data in the same
// order (given the same workgroup size); ONE thread of the launch records them (dogleg_choose_record); the
// elementwise part - the Cauchy step of a new point, the step, the trial state b[ia] = b[ib] + step - is written
// by the threads that call dogleg_choose_elementwise(). What is recorded is not read back by dogleg_choose_scalars()
// in a way that changes the outcome of the launch (see the notes at the reads).
// A workgroup that needs entries of the TRIAL state in the same launch computes them itself: TrialState[i] runs
// the same instructions as the elementwise writer, so it holds the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include "solver_kernels.hpp"

namespace mrcal_amd {

// flags derived from the control state, for the kernels' skip pointers
//   skip_factor: this trial does not need a factorization
//   skip_eval:   this trial does not evaluate a new point
//   (the fused step) elim_mode: 0 nothing to eliminate; 1 the trial point was evaluated: its blocks come from the
//   Grams; 2 the current point is re-eliminated from its stored blocks. elim_sel: which operating point that is.
//   skip_elim = (elim_mode == 0), skip_asm = (elim_mode != 1); skip_chol / skip_backsub: set by the finish logic
struct SolverCtlFlags { int skip_factor, skip_eval;
                        int elim_mode, elim_sel, skip_elim, skip_asm, skip_chol, skip_backsub; };
static_assert(sizeof(SolverCtlFlags) == 32, "");


#define COMM2_GNG    0
#define COMM2_GGE    1
#define COMM2_GNE2   2
#define COMM2_GNE_GE 3

__device__ __forceinline__ void ctl_raise_lambda(SolverCtl* ctl)
{
    double lam = ctl->lambda;
    lam = (lam == 0.0) ? 1e-10 : lam*10.0;
    ctl->lambda = lam;
    if(!(lam < 1e30)) { ctl->error = 1; ctl->done = 1; }
}

// Fixed-order sums by a whole workgroup. thread_sum_fixed: this thread's share of n values v(i) (items t, t + W,
// ...), NOUT values at once, added in index order into acc. block_sum_finish: the threads' shares into one sum,
// in an order that depends on the workgroup size alone; the result in every thread.
// Eight items per thread are asked for at once (from always-valid indices: no branch around a load), then added
// in index order: the sum is what a one-at-a-time loop gives, but a thread waits for memory once per eight items
// (every workgroup of the prologue launch runs these sums in front of its own work)
template<int NOUT, class F>
__device__ __forceinline__ void thread_sum_fixed(int n, F&& v, double* __restrict__ acc)
{
    constexpr int U = 8;
    for(int i0 = threadIdx.x; i0 < n; i0 += U*blockDim.x)
    {
        double t[U][NOUT];
#pragma unroll
        for(int u = 0; u < U; u++)
        {
            const int i = i0 + u*blockDim.x;
            v(i < n ? i : n - 1, t[u]);
        }
#pragma unroll
        for(int u = 0; u < U; u++)
        {
            const bool in = i0 + u*(int)blockDim.x < n;
#pragma unroll
            for(int k = 0; k < NOUT; k++) acc[k] += in ? t[u][k] : 0.0;
        }
    }
}
// The same sums with the loads of SEVERAL of them in flight together (round 5): a thread's first U items of a sum are
// asked for by chunk_load() - nothing is added yet -, the sums' chunks one after the other; chunk_add() then adds them
// in index order, and thread_sum_fixed_from() goes on behind item U W the way thread_sum_fixed() does. The order of
// every thread's additions is the one thread_sum_fixed() has (items t, t + W, t + 2 W, ...): the same bits; what
// changes is that the choice's four sums wait for memory once between them instead of once or twice each
template<int NOUT, int U, class F>
__device__ __forceinline__ void chunk_load(int n, F&& v, double (&t)[U][NOUT])
{
#pragma unroll
    for(int u = 0; u < U; u++)
    {
        const int i = threadIdx.x + u*blockDim.x;
        v(i < n ? i : (n > 0 ? n - 1 : 0), t[u]);
    }
}
template<int NOUT, int U>
__device__ __forceinline__ void chunk_add(int n, const double (&t)[U][NOUT], double* __restrict__ acc)
{
#pragma unroll
    for(int u = 0; u < U; u++)
    {
        const bool in = (int)(threadIdx.x + u*blockDim.x) < n;
#pragma unroll
        for(int k = 0; k < NOUT; k++) acc[k] += in ? t[u][k] : 0.0;
    }
}
template<int NOUT, class F>
__device__ __forceinline__ void thread_sum_fixed_from(int first, int n, F&& v, double* __restrict__ acc)
{
    constexpr int U = 8;
    for(int i0 = first + threadIdx.x; i0 < n; i0 += U*blockDim.x)
    {
        double t[U][NOUT];
#pragma unroll
        for(int u = 0; u < U; u++)
        {
            const int i = i0 + u*blockDim.x;
            v(i < n ? i : n - 1, t[u]);
        }
#pragma unroll
        for(int u = 0; u < U; u++)
        {
            const bool in = i0 + u*(int)blockDim.x < n;
#pragma unroll
            for(int k = 0; k < NOUT; k++) acc[k] += in ? t[u][k] : 0.0;
        }
    }
}
template<int NOUT>
__device__ __forceinline__ void block_sum_finish(double (&acc)[NOUT], double (&out)[NOUT], double* __restrict__ scratch /* [17][NOUT] */)
{
#pragma unroll
    for(int k = 0; k < NOUT; k++)
        for(int off=32; off>0; off>>=1) acc[k] += __shfl_down(acc[k], off);
    __syncthreads();            // scratch is free
    if((threadIdx.x & 63) == 0)
#pragma unroll
        for(int k = 0; k < NOUT; k++) scratch[(threadIdx.x >> 6)*NOUT + k] = acc[k];
    __syncthreads();
    const int nw = blockDim.x >> 6;
    if(threadIdx.x < NOUT)
    {
        double t = 0.0;
        for(int w = 0; w < nw; w++) t += scratch[w*NOUT + threadIdx.x];
        scratch[16*NOUT + threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for(int k = 0; k < NOUT; k++) out[k] = scratch[16*NOUT + k];
}
template<int NOUT, class F>
__device__ __forceinline__ void block_sum_fixed(int n, F&& v, double (&out)[NOUT], double* __restrict__ scratch /* [17][NOUT] */)
{
    double acc[NOUT];
#pragma unroll
    for(int k = 0; k < NOUT; k++) acc[k] = 0.0;
    thread_sum_fixed<NOUT>(n, v, acc);
    block_sum_finish<NOUT>(acc, out, scratch);
}

struct ChooseArgs
{
    NormalDims nd; const OpDev* ops; SolverCtl* ctl; SolverCtlFlags* fl;
    int* chol_status; double* step;
    const double* qf_part; int qf_n;           // per-workgroup partials of g^T N g, |g_E|^2 (quadratic-form workgroups)
    const double* dots_part; int dots_n;       // per-block partials of |gn_E|^2, gn_E . g_E (back-substitution)
    const double* comm2;                       // sharded: those four sums over the ranks instead; NULL: single GPU
};
struct ChooseOut
{
    int    done_already, ib, ia, derive, voided, fresh_gn, gn_nan, edge;
    int    skip_eval;                          // what the trial's evaluation will see in fl->skip_eval
    double gNg, gg, kcau, norm2a, gn_lensq, gn_dot_g, norm2b, ab, kc, kg, len_sq;
};

// one entry of the dog-leg step and of the trial state. ci: the entry of the Cauchy step; c_raw: g[i] when the
// point is new (derive: its Cauchy step is kcau g), step_cauchy[i] otherwise. No contraction across the
// statements: the two places that evaluate this must round alike
__device__ __forceinline__
double dogleg_trial_value(double b_i, double c_raw, double gn_i, double kcau, double kc, double kg, int derive,
                          double* ci_out, double* sv_out)
{
    const double ci = derive ? __dmul_rn(kcau, c_raw) : c_raw;
    double sv = __dmul_rn(kc, ci);
    if(kg != 0.0) sv = __fma_rn(kg, gn_i, sv);
    *ci_out = ci; *sv_out = sv;
    return __dadd_rn(b_i, sv);
}

// the whole workgroup; scratch [17*7] doubles of LDS. Reads only
__device__ __forceinline__
ChooseOut dogleg_choose_scalars(const ChooseArgs& a, double* __restrict__ scratch)
{
    const NormalDims& nd = a.nd;
    const SolverCtl* ctl = a.ctl;
    ChooseOut c;
    memset(&c, 0, sizeof(c));
    // (done: set by this launch's record only when the step is shorter than the termination threshold or lambda
    //  ran away; a workgroup that starts late and sees it skips the evaluation, which is what the record decided)
    if(ctl->done) { c.done_already = 1; c.skip_eval = 1; return c; }
    const int ib = ctl->ib, ia = ctl->ia;
    c.ib = ib; c.ia = ia;
    const OpDev& from = a.ops[ib];
    const bool derive = ctl->derive != 0;
    c.derive = derive;
    auto s_to_state = [&](int i) -> int { return S_to_state(nd, i); };

    // Every sum this choice may need, accumulated per thread first and reduced across the workgroup ONCE:
    //   [0] |g_S|^2  [1] g^T N g  [2] |g_E|^2                      (a new point: derive)
    //   [3] |gn_S|^2 [4] gn_S.g_S [5] |gn_E|^2 [6] gn_E.g_E        (a fresh Gauss-Newton step; computed whenever
    //                                                                one is there: whether it is used is known after [0..2])
    const bool have_fresh = ctl->gn_fresh != 0 && !ctl->refactor;
    double o[7];
    if(derive || have_fresh)
    {
        double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // (the four sums' loads in flight together, then their additions - each in its own, unchanged order)
        constexpr int US = 4, UL = 16;          // a thread's first items of the short sums (the camera block) and of the long ones (the partials)
        auto f_g2  = [&](int i, double (&t)[1]) { const double v = from.g[s_to_state(i)]; t[0] = v*v; };
        auto f_qf  = [&](int i, double (&t)[2]) { t[0] = a.qf_part[4*i]; t[1] = a.qf_part[4*i + 2]; };
        auto f_gn  = [&](int i, double (&t)[2]) { const int is = s_to_state(i); const double gn = from.step_gn[is]; t[0] = gn*gn; t[1] = gn*from.g[is]; };
        auto f_dot = [&](int i, double (&t)[2]) { t[0] = a.dots_part[2*i]; t[1] = a.dots_part[2*i + 1]; };
        const bool do_qf = derive && a.comm2 == NULL, do_dot = have_fresh && a.comm2 == NULL;
        double t_g2[US][1], t_qf[UL][2], t_gn[US][2], t_dot[UL][2];
        if(derive)     chunk_load<1, US>(nd.Nc,    f_g2,  t_g2);
        if(do_qf)      chunk_load<2, UL>(a.qf_n,   f_qf,  t_qf);
        if(have_fresh) chunk_load<2, US>(nd.Nc,    f_gn,  t_gn);
        if(do_dot)     chunk_load<2, UL>(a.dots_n, f_dot, t_dot);
        if(derive)     { chunk_add<1, US>(nd.Nc,    t_g2,  acc + 0); thread_sum_fixed_from<1>(US*blockDim.x, nd.Nc,    f_g2,  acc + 0); }
        if(do_qf)      { chunk_add<2, UL>(a.qf_n,   t_qf,  acc + 1); thread_sum_fixed_from<2>(UL*blockDim.x, a.qf_n,   f_qf,  acc + 1); }
        if(have_fresh) { chunk_add<2, US>(nd.Nc,    t_gn,  acc + 3); thread_sum_fixed_from<2>(US*blockDim.x, nd.Nc,    f_gn,  acc + 3); }
        if(do_dot)     { chunk_add<2, UL>(a.dots_n, t_dot, acc + 5); thread_sum_fixed_from<2>(UL*blockDim.x, a.dots_n, f_dot, acc + 5); }
        block_sum_finish<7>(acc, o, scratch);
        if(a.comm2 != NULL) { o[1] = a.comm2[COMM2_GNG]; o[2] = a.comm2[COMM2_GGE]; o[5] = a.comm2[COMM2_GNE2]; o[6] = a.comm2[COMM2_GNE_GE]; }
    }
    // the point's own numbers
    double gNg, gg, kcau, norm2a;
    if(derive)
    {
        gNg = o[1]; gg = o[0] + o[2];
        kcau = (gNg > 0.0) ? -gg/gNg : 0.0;
        norm2a = kcau*kcau*gg;
    }
    else
    {
        // (written by the record only when derive: not this branch)
        gNg = from.scalars[SC_G_GNG]; gg = from.scalars[SC_G_GG];
        kcau = (gNg > 0.0) ? -gg/gNg : 0.0;
        norm2a = ctl->cauchy_lensq[ib];
    }
    c.gNg = gNg; c.gg = gg; c.kcau = kcau; c.norm2a = norm2a;
    const double tr = ctl->trustregion, dsq = tr*tr;
    const bool cauchy_only = norm2a >= dsq;
    // (refactor, gn_valid: the record sets them only where this comes out true anyway)
    const bool voided = ctl->refactor || (!cauchy_only && !ctl->gn_valid[ib]);
    c.voided = voided;
    const bool fresh_gn = !voided && !cauchy_only && ctl->gn_fresh != 0;
    c.fresh_gn = fresh_gn;

    double gn_lensq = ctl->gn_lensq[ib], gn_dot_g = ctl->gn_dot_g[ib];
    if(fresh_gn) { gn_lensq = o[3] + o[5]; gn_dot_g = o[4] + o[6]; }
    c.gn_lensq = gn_lensq; c.gn_dot_g = gn_dot_g;
    const bool gn_nan = fresh_gn && !(gn_lensq == gn_lensq);
    c.gn_nan = gn_nan;

    double kc = 0.0, kg = 0.0, len_sq = 0.0;
    int edge = 0;
    double norm2b = 0.0, ab = 0.0;
    if(!voided && !gn_nan)
    {
        if(cauchy_only)
        {
            kc = tr/sqrt(norm2a); kg = 0.0; len_sq = dsq; edge = 1;
        }
        else
        {
            norm2b = gn_lensq;
            ab     = kcau*gn_dot_g;            // step_gn . step_cauchy
            if(norm2b <= dsq)
            {
                kc = 0.0; kg = 1.0; len_sq = norm2b; edge = 0;
            }
            else
            {
                // point on the Cauchy->GN segment at the trust-region edge:
                // |a + k(b-a)|^2 = dsq, a = Cauchy, b = GN
                const double l2    = norm2a - 2.0*ab + norm2b;   // |a-b|^2
                const double neg_c = norm2a - ab;                // a.(a-b)
                double disc = neg_c*neg_c - l2*(norm2a - dsq);
                if(disc < 0.0) disc = 0.0;
                const double k = (neg_c + sqrt(disc))/l2;
                kc = 1.0 - k; kg = k;
                len_sq = kc*kc*norm2a + 2.0*kg*kc*ab + kg*kg*norm2b;
                edge = 1;
            }
        }
    }
    c.kc = kc; c.kg = kg; c.len_sq = len_sq; c.edge = edge; c.norm2b = norm2b; c.ab = ab;
    c.skip_eval = (voided || gn_nan ||
                   (ctl->check_termination && len_sq < ctl->update_threshold*ctl->update_threshold)) ? 1 : 0;
    return c;
}

// thread i of Nstate: the Cauchy step of a new point, the step, the trial state
__device__ __forceinline__
void dogleg_choose_elementwise(const ChooseArgs& a, const ChooseOut& c, int i)
{
    if(c.done_already || i < 0 || i >= a.nd.Nstate) return;
    const OpDev& from = a.ops[c.ib];
    const double c_raw = c.derive ? from.g[i] : from.step_cauchy[i];
    const double gn_i  = (c.kg != 0.0) ? from.step_gn[i] : 0.0;
    double ci, sv;
    const double bt = dogleg_trial_value(from.b[i], c_raw, gn_i, c.kcau, c.kc, c.kg, c.derive, &ci, &sv);
    if(c.derive) from.step_cauchy[i] = ci;
    if(!c.voided && !c.gn_nan)
    {
        a.step[i] = sv;
        a.ops[c.ia].b[i] = bt;
    }
}

// The trial state, entry by entry, for a workgroup of the launch that chooses it (the stored copy is being
// written by other workgroups of the same launch)
struct TrialState
{
    const double* b; const double* c_raw; const double* gn; double kcau, kc, kg; int derive;
    __device__ __forceinline__ double operator[](int i) const
    {
        double ci, sv;
        return dogleg_trial_value(b[i], c_raw[i], (kg != 0.0) ? gn[i] : 0.0, kcau, kc, kg, derive, &ci, &sv);
    }
};
__device__ __forceinline__ TrialState dogleg_trial_state(const ChooseArgs& a, const ChooseOut& c)
{
    const OpDev& from = a.ops[c.ib];
    TrialState t = { from.b, c.derive ? from.g : from.step_cauchy, from.step_gn, c.kcau, c.kc, c.kg, c.derive };
    return t;
}

// ONE thread of the launch: the control block, the flags of this trial, the point's scalars
__device__ __forceinline__
void dogleg_choose_record(const ChooseArgs& a, const ChooseOut& c)
{
    SolverCtl* ctl = a.ctl;
    SolverCtlFlags* fl = a.fl;
    if(c.done_already)
    {
        fl->skip_eval = 1; fl->elim_mode = 0; fl->skip_elim = 1; fl->skip_asm = 1;
        return;
    }
    const int ib = c.ib, ia = c.ia;
    const OpDev& from = a.ops[ib];
    if(c.voided)
    {
        // no step can be chosen: this trial eliminates the current point (again)
        ctl->refactor = 1; ctl->abort_step = 1;
        *a.chol_status = 0;
        fl->skip_eval = 1; fl->elim_mode = 2; fl->elim_sel = ib; fl->skip_elim = 0; fl->skip_asm = 1;
        // (the Cauchy step of a new point is still recorded below)
    }
    if(c.derive)
    {
        from.scalars[SC_G_GNG] = c.gNg; from.scalars[SC_G_GG] = c.gg; from.scalars[SC_G_GG2] = c.gg;
        ctl->cauchy_lensq[ib] = c.norm2a;
    }
    if(c.voided) return;
    if(c.gn_nan)
    {
        // a Gauss-Newton step that is not a number: treat the factorization as failed
        ctl_raise_lambda(ctl);
        ctl->refactor = 1; ctl->abort_step = 1; ctl->gn_valid[ib] = 0;
        *a.chol_status = 0;
        fl->skip_eval = 1; fl->elim_mode = ctl->done ? 0 : 2; fl->elim_sel = ib;
        fl->skip_elim = ctl->done ? 1 : 0; fl->skip_asm = 1;
        return;
    }
    if(c.fresh_gn)
    {
        ctl->gn_lensq[ib] = c.gn_lensq;
        ctl->gn_dot_g[ib] = c.gn_dot_g;
        from.scalars[SC_GN_LENSQ] = c.gn_lensq; from.scalars[SC_GN_DOT_CAUCHY] = c.ab;
    }
    // The expected improvement |x|^2 - |x + J s|^2 = -2 g.s - s^T N s WITHOUT a
    // pass over N: the step is kc s_c + kg s_gn with s_c = k g and
    // (N + lambda I) s_gn = -g, so every term is a dot product already at hand:
    //   s_c^T N s_c   = k^2 g^T N g
    //   s_c^T N s_gn  = -k g.g - lambda s_c.s_gn
    //   s_gn^T N s_gn = -g.s_gn - lambda |s_gn|^2
    {
        const double kc = c.kc, kg = c.kg, kcau = c.kcau;
        double sNs = kc*kc*kcau*kcau*c.gNg, gs = kc*kcau*c.gg;
        if(kg != 0.0)
        {
            const double aa = c.gn_dot_g, lam = ctl->gn_lambda[ib];
            sNs += 2.0*kc*kg*(-kcau*c.gg - lam*c.ab) + kg*kg*(-aa - lam*c.norm2b);
            gs  += kg*aa;
        }
        from.scalars[SC_STEP_SNS] = sNs;
        from.scalars[SC_STEP_GS]  = gs;
        from.scalars[SC_STEP_SS]  = c.len_sq;
    }
    ctl->k_cauchy = c.kc; ctl->k_gn = c.kg;
    ctl->step_len_sq = c.len_sq;
    ctl->did_step_to_edge[ib] = c.edge;
    ctl->abort_step = 0;
    ctl->Ntrials++;
    *a.chol_status = 0;
    if(ctl->check_termination && c.len_sq < ctl->update_threshold*ctl->update_threshold)
    {
        ctl->done = 1;
        fl->skip_eval = 1; fl->elim_mode = 0; fl->skip_elim = 1; fl->skip_asm = 1;
    }
    else
    {
        fl->skip_eval = 0; fl->elim_mode = 1; fl->elim_sel = ia; fl->skip_elim = 0; fl->skip_asm = 0;
    }
}

} // namespace mrcal_amd
