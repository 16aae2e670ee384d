// Device functions and small types that more than one translation unit of the solver's kernels needs
// (round 6: solver_kernels.hip, one 6800-line translation unit, was cut by phase: assembly.hip, assembly_splined.hip,
// schur.hip, cholesky_lds.hip, cholesky_large.hip, step.hip, factorization_solve.hip). Everything here is
// __forceinline__ / inline / a template: each unit that uses a function compiles its own copy (no relocatable device code).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include "problem.hpp"
#include "solver_kernels.hpp"
#include "dogleg_choose.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// index helpers
////////////////////////////////////////////////////////////////////////////////
// E index -> (block, offset in block, block size, E index of block start)
__device__ __forceinline__ void E_to_block(const NormalDims& nd, int e, int* blk, int* a, int* de, int* e0)
{
    if(e < 6*nd.Nfb) { *blk = e/6; *a = e - 6*(*blk); *de = 6; *e0 = 6*(*blk); }
    else
    {
        const int ee = e - 6*nd.Nfb;
        const int ib = ee/3;
        *blk = nd.Nfb + ib; *a = ee - 3*ib; *de = 3; *e0 = 6*nd.Nfb + 3*ib;
    }
}

// Every destination of the camera-block part - an entry of A, of g (S part) or
// |x|^2 - adds its sources (chunk partials of the pairs that touch it) in the
// order of the plan. One thread per destination that has sources. What the rows
// that do not come from Grams added earlier (atomically, into the zeroed
// buffers: regularization rows, which have disjoint destinations; discrete
// points) stays. |x|^2 also takes those rows' per-workgroup partials, in order
// 16 lanes (one DPP row) per destination: the lanes split the chunks of each source,
// then add up in a fixed order
// (round 6: LANES a parameter - the planned rows' destinations are few and their sources' chunks many (configuration 4: 343
//  destinations, up to three groups of 87 chunks each): with a whole wave a destination their finalize is one trip to
//  memory a source instead of three, 10.6 -> see LEDGER R6.19. The Grams' plan keeps its 16 - and its bits)
#define FIN_LANES 16
#define FIN_LANES_GEN 64
template<int LANES = FIN_LANES>
__device__ __forceinline__
void assemble_finalize(int npos, const NormalDims& nd, const OpDev& O, const AssemblyPlan& plan,
                       int gid /* global thread index: destination gid/LANES, lane gid%LANES */)
{
    constexpr int FIN_L = LANES;
    const int k = gid / FIN_L, j = gid % FIN_L;
    const bool live = k < plan.Ndest;
    const int kc = live ? k : 0;
    const int d  = plan.dest_id[kc];
    const int s0 = plan.dest_begin[kc], s1 = live ? plan.dest_begin[kc+1] : s0;
    double acc = 0.0;
    for(int s = s0; s < s1; s++)
    {
        const int src = plan.dest_src[s];
        const int pair = src >> 10, pos = src & 1023;
        const int c0 = plan.pair_chunk_begin[pair], c1 = plan.pair_chunk_begin[pair+1];
        const double* __restrict__ cp = plan.chunk_part + pos;
        double a0 = 0.0, a1 = 0.0;
        int c = c0 + j;
        for(; c + FIN_L < c1; c += 2*FIN_L)
        {
            const double v0 = cp[(size_t)c*npos], v1 = cp[(size_t)(c + FIN_L)*npos];
            a0 += v0; a1 += v1;
        }
        if(c < c1) a0 += cp[(size_t)c*npos];
        acc += a0 + a1;
    }
    // (all the lanes of the row take part, whether the destination is live or not)
    for(int off = FIN_L/2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if(!live || j != 0) return;
    const int nA = nd.Nc*nd.Nc;
    if(d < nA)              O.A[d] += acc;
    else if(d < nA + nd.Nc) { const int sc = d - nA; O.g[S_to_state(nd, sc)] += acc; }
    else
    {
        for(int b = 0; b < plan.row_part_n; b++) acc += plan.row_part[b];
        O.scalars[SC_NORM2_X] += acc;
    }
}
// Rows that do not come from board observations (discrete points,
// regularization): one lane per CSR row, scattered with atomics. These are few
__device__ __forceinline__
void rows_generic_row(const NormalDims& nd, const OpDev& O, int r, int row1,
                      const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                      double* n2_local = NULL /* if given: x^2 goes there instead of into |x|^2 (atomically) */)
{
    const double* __restrict__ Jv = O.Jv;
    const double* __restrict__ x  = O.x;
    double* __restrict__ A  = O.A;
    double* __restrict__ Bt = O.Bt;
    double* __restrict__ D  = O.D;
    double* __restrict__ g  = O.g;
    double* __restrict__ norm2_x = &O.scalars[SC_NORM2_X];
    if(r >= row1) return;
    const int p0 = Jp[r], p1 = Jp[r+1];
    const double xr = x[r];
    if(n2_local) *n2_local = xr*xr; else atomicAdd(norm2_x, xr*xr);
    for(int p = p0; p < p1; p++)
    {
        const int    ci = Ji[p];
        const double vi = Jv[p];
        if((unsigned)ci >= (unsigned)nd.Nstate) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }
        atomicAdd(&g[ci], vi*xr);
        const int si = state_to_SE(nd, ci);
        for(int q = p0; q < p1; q++)
        {
            const int    cj = Ji[q];
            if((unsigned)cj >= (unsigned)nd.Nstate) continue;
            const double v  = vi*Jv[q];
            const int    sj = state_to_SE(nd, cj);
            if(si >= 0 && sj >= 0)
                atomicAdd(&A[(size_t)si*nd.Nc + sj], v);
            else if(si < 0 && sj >= 0)
                atomicAdd(&Bt[(size_t)(-si-1)*nd.Nc + sj], v);
            else if(si < 0 && sj < 0)
            {
                int bi, ai, di, e0i, bj, aj, dj, e0j;
                E_to_block(nd, -si-1, &bi, &ai, &di, &e0i);
                E_to_block(nd, -sj-1, &bj, &aj, &dj, &e0j);
                if(bi == bj) atomicAdd(&D[(size_t)bi*36 + ai*6 + aj], v);
                else         O.scalars[SC_BAD_STRUCTURE] = 1.0;      // no row may touch two E blocks
            }
        }
    }
}
// the same, riding along in the SYRK launch (row nslices of its grid): the two are independent
struct FinalizeRide
{
    int           row0;      // first row (blockIdx.y) of the grid that is the ride's
    int           npos;      // 0: nothing rides along
    const OpDev*  ops;
    const int*    sel;
    const int*    skip;
    AssemblyPlan  plan;
};
__device__ __forceinline__ void finalize_ride(const FinalizeRide& fr, const NormalDims& nd)
{
    if(fr.skip != NULL && *fr.skip) return;
    const OpDev& O = fr.ops[fr.sel ? *fr.sel : 0];
    assemble_finalize(fr.npos, nd, O, fr.plan, ((blockIdx.y - fr.row0)*gridDim.x + blockIdx.x)*blockDim.x + threadIdx.x);
}

// lane `lane`'s value of v, to every lane (a wave-uniform lane)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
// The fused step (see "dog-leg control" below) puts the end of a trial - accept
// or reject, the trust region, does the current point need its Gauss-Newton
// step - in front of the factorization, in the same launch (FINISH)
struct Step2Dev
{
    NormalDims nd; const OpDev* ops; SolverCtl* ctl; SolverCtlFlags* fl;
    int initial; const double* comm1_tail;       // [g_S (Nc) | |x|^2 | status] behind S and r
};

////////////////////////////////////////////////////////////////////////////////
// dog-leg control (libdogleg's trust-region logic, on the device)
////////////////////////////////////////////////////////////////////////////////
// rho test, trust-region update, accept/reject (one thread)
__device__ __forceinline__ void ctl_accept(const OpDev* __restrict__ ops, SolverCtl* ctl)
{
    const int ib = ctl->ib, ia = ctl->ia;
    const OpDev& from = ops[ib];
    // expected improvement: |x|^2 - |x + J s|^2 = -2 g.s - s^T N s
    const double expected = -2.0*from.scalars[SC_STEP_GS] - from.scalars[SC_STEP_SNS];
    ctl->expected_improvement = expected;
    const double observed = ctl->norm2_x[ib] - ctl->norm2_x[ia];
    double rho = observed/expected;
    // a trial point where the cost function is not finite (or a 0/0) is a
    // rejected step with a shrinking trust region, not a comparison with NaN
    // that neither accepts nor shrinks
    if(!(rho == rho) || !(ctl->norm2_x[ia] == ctl->norm2_x[ia])) rho = -1.0;
    double tr = ctl->trustregion;
    if(rho < ctl->trustregion_decrease_threshold)
        tr *= ctl->trustregion_decrease_factor;
    else if(rho > ctl->trustregion_increase_threshold && ctl->did_step_to_edge[ib])
        tr *= ctl->trustregion_increase_factor;
    ctl->trustregion = tr;
    if(rho > 0.0)
    {
        ctl->ib = ia; ctl->ia = ib;
        ctl->Nsteps_accepted++;
    }
    else if(ctl->check_termination &&
            (tr < ctl->trustregion_threshold || tr == 0.0 || !(tr == tr)))
        ctl->done = 1;
}
// End of a trial, in ONE workgroup, in front of the factorization: the rho test
// with accept/reject (ctl_accept), the termination tests; then: does the
// (possibly new) current point get its Gauss-Newton step now? Returns that, to
// every thread. comm1 = [S | r | g_S | |x|^2 | status] is complete (summed over
// the ranks when sharded): the camera-block part of the new point's gradient is
// taken from it
// no_unpack: the caller is the reduction that made the tail from this very gradient (single GPU): nothing to copy back
inline __device__ bool step2_finish(const Step2Dev& sd, int* chol_status, bool no_unpack = false)      // one workgroup; true: factor
{
    const NormalDims& nd = sd.nd;
    SolverCtl* ctl = sd.ctl;
    SolverCtlFlags* fl = sd.fl;
    const int t = threadIdx.x, nt = blockDim.x;
    const int mode = fl->elim_mode;
    __shared__ int s_go, s_unpack;
    const int ip = sd.initial ? ctl->ib : ctl->ia;        // the point that was evaluated (mode 1)
    const double* __restrict__ tail = sd.comm1_tail;
    const double norm2_x = tail[nd.Nc];
    const bool   eblock_failed = tail[nd.Nc + 1] != 0.0;
    __syncthreads();                      // everyone has read the control state
    if(t == 0)
    {
        if(mode == 1)
        {
            sd.ops[ip].scalars[SC_NORM2_X] = norm2_x;
            ctl->norm2_x[ip]  = norm2_x;
            ctl->gn_valid[ip] = 0;
            ctl->did_step_to_edge[ip] = 0;
            ctl->Nevaluations++;
            if(!sd.initial) ctl_accept(sd.ops, ctl);
        }
        int go = 0, unpack = 0;
        ctl->gn_fresh = 0;
        ctl->derive   = 0;
        if(!ctl->done && ctl->check_termination && ctl->Nsteps_accepted >= ctl->max_iterations)
            ctl->done = 1;
        const int ib = ctl->ib;
        if(mode == 1 && ib == ip) { unpack = 1; ctl->derive = 1; }       // a new current point
        if(!ctl->done)
        {
            if(mode == 2)      go = 1;
            else if(mode == 1) go = (ib == ip);
        }
        if(go && eblock_failed)
        {
            // a 6x6 (3x3) block was not positive definite: regularize, like libdogleg does
            ctl_raise_lambda(ctl);
            ctl->refactor = 1;
            go = 0;
        }
        if(go) ctl->Nfactorizations++;
        fl->skip_backsub = 1;             // until the factorization has succeeded
        fl->skip_chol    = go ? 0 : 1;
        s_go = go; s_unpack = unpack;
        (void)chol_status;
    }
    __syncthreads();
    if(s_unpack && !no_unpack)
    {
        const OpDev& O = sd.ops[ip];
        for(int i = t; i < nd.Nc; i += nt) O.g[S_to_state(nd, i)] = tail[i];
    }
    return s_go != 0;
}
inline __device__ void step2_chol_done(const Step2Dev& sd, bool not_positive_definite)       // one thread
{
    SolverCtl* ctl = sd.ctl;
    if(not_positive_definite)
    {
        ctl_raise_lambda(ctl);
        ctl->refactor = 1;
        sd.fl->skip_backsub = 1;
    }
    else
    {
        const int ib = ctl->ib;
        ctl->refactor      = 0;
        ctl->gn_valid[ib]  = 1;
        ctl->gn_lambda[ib] = ctl->lambda;
        ctl->gn_fresh      = 1;
        sd.fl->skip_backsub = 0;
    }
}

// four doubles: an accumulator of v_mfma_f64_16x16x4 (the SYRK kernels, the large Cholesky)
typedef double syrk_d4 __attribute__((ext_vector_type(4)));

} // namespace mrcal_amd
