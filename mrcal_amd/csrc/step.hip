// The dog-leg trial step's own kernels (quadratic forms, outliers, the reduction's and the factorization's neighbours) and its launchers
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "chol_diag16.hpp"
#include "solver_kernel_decls.hpp"
#include "step_device.hpp"

namespace mrcal_amd {


// (quadform_body, backsub_eblock: step_device.hpp - cholesky_lds.hip runs them beside the factorization)
__global__ __launch_bounds__(256)
void quadform_kernel(NormalDims nd, OpRef R, const double* __restrict__ v_in, int v_is_g,
                     double* __restrict__ out_in, int out_in_scalars_at, int nout)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const double* __restrict__ v   = v_is_g ? O.g : v_in;
    double*       __restrict__ out = (out_in != NULL) ? out_in : (O.scalars + out_in_scalars_at);
    const double mine = quadform_body(nd, O, v, blockIdx.x);
    if(threadIdx.x < nout) atomicAdd(&out[threadIdx.x], mine);
}

__global__ __launch_bounds__(256)
void dot_kernel(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out)
{
    double acc = 0.0;
    for(int i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += gridDim.x*blockDim.x)
        acc += a[i]*b[i];
    for(int off=32; off>0; off>>=1) acc += __shfl_down(acc, off);
    __shared__ double part[4];
    if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if(threadIdx.x == 0) atomicAdd(out, part[0]+part[1]+part[2]+part[3]);
}

// y = alpha a + beta b
__global__ __launch_bounds__(256)
void axpby_kernel(int n, double alpha, const double* __restrict__ a, double beta, const double* __restrict__ b,
                  double* __restrict__ y)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i < n) y[i] = alpha*a[i] + ((b != NULL) ? beta*b[i] : 0.0);
}

////////////////////////////////////////////////////////////////////////////////
// outlier rejection
////////////////////////////////////////////////////////////////////////////////
// marks board-corner outliers: weight *= -1 for inliers with |x| > k sigma in
// either coordinate (mrcal.c:4320-4345). counts[0] += number newly marked
__global__ __launch_bounds__(256)
void mark_outliers_kernel(int Npoints_board, double thresh_sq,
                          const double* __restrict__ x, double* __restrict__ pool,
                          int* __restrict__ counts)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= Npoints_board) return;
    const double w = pool[3*(size_t)i + 2];
    if(w <= 0.0) return;
    const double dx = x[2*(size_t)i], dy = x[2*(size_t)i+1];
    if(dx*dx > thresh_sq || dy*dy > thresh_sq)
    {
        pool[3*(size_t)i + 2] = -w;
        atomicAdd(&counts[0], 1);
    }
}

// outlier statistics (mrcal.c:4107-4124, 4282-4306): counts[0] = current
// outliers (weight <= 0), counts[1] = inliers beyond k1 sigma given var,
// sums[0] = sum of inlier x^2
// Fixed grid, grid-stride loop, per-workgroup partial sums in part[], summed in
// order by outlier_stats_sum_kernel: the variance (and with it the outlier
// threshold) does not depend on scheduling
#define OUTLIER_BLOCKS 512
__global__ __launch_bounds__(256)
void outlier_stats_kernel(int Npoints_board, double thresh_sq,
                          const double* __restrict__ x, const double* __restrict__ pool,
                          int* __restrict__ counts, double* __restrict__ part)
{
    double s = 0.0;
    int nout = 0, nbig = 0;
    for(int i = blockIdx.x*blockDim.x + threadIdx.x; i < Npoints_board; i += gridDim.x*blockDim.x)
    {
        const double w = pool[3*(size_t)i + 2];
        if(w <= 0.0) nout++;
        else
        {
            const double dx = x[2*(size_t)i], dy = x[2*(size_t)i+1];
            s += dx*dx + dy*dy;
            if(thresh_sq >= 0.0 && (dx*dx > thresh_sq || dy*dy > thresh_sq)) nbig++;
        }
    }
    for(int off=32; off>0; off>>=1)
    {
        s    += __shfl_down(s, off);
        nout += __shfl_down(nout, off);
        nbig += __shfl_down(nbig, off);
    }
    __shared__ double ps[4];
    if((threadIdx.x & 63) == 0)
    {
        ps[threadIdx.x >> 6] = s;
        if(nout) atomicAdd(&counts[0], nout);       // (integers: any order gives the same sum)
        if(nbig) atomicAdd(&counts[1], nbig);
    }
    __syncthreads();
    if(threadIdx.x == 0) part[blockIdx.x] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
}
__global__ __launch_bounds__(64)
void outlier_stats_sum_kernel(int n, const double* __restrict__ part, double* __restrict__ sums)
{
    if(threadIdx.x != 0) return;
    double s = 0.0;
    for(int i = 0; i < n; i++) s += part[i];
    sums[0] += s;
}

////////////////////////////////////////////////////////////////////////////////
// the fused step: choose | evaluate | assemble+eliminate | SYRK+finalize | reduce
// | finish+Cholesky | backsub+quadform.  See solver_kernels.hpp
////////////////////////////////////////////////////////////////////////////////
// The Gauss-Newton step is computed EAGERLY: a point that is accepted is
// factored in the launch that accepts it, from the elimination that rode along
// in its assembly. libdogleg computes it lazily at the start of the next trial;
// the step taken is the same. ctl->refactor: the current point must be
// eliminated (again) before a step can be chosen from it - lambda was raised
// after a failed factorization (libdogleg: "singular JtJ: adding lambda I from now
// on"), or its Gauss-Newton step was never computed.
//
// The SAME step runs sharded over several GPUs (frames partitioned over the
// ranks; mrcal_amd/parallel.py, solver.cpp). What a rank computes from its own
// frames only is summed over the ranks in TWO collectives per trial:
//   comm1 (after the reduce):    [ S | r | g_S | |x|^2 | status ]        Nc^2 + 2 Nc + 2 doubles
//   comm2 (after the backsub):   [ g^T N g | |g_E|^2 | |gn_E|^2 | gn_E . g_E ]      4 doubles
// Everything else is either rank-local (the frame part of the state, of g, of the
// steps) or REPLICATED: the camera block of the state, and the control block,
// which every rank advances with the same kernels on the same sums, so that all
// ranks take the same decisions without talking about them. For that the sums a
// rank forms by itself from replicated data must be bit-identical on all ranks:
// fixed-order reductions everywhere, no atomics.
//   comm1 == F.S (S, r and the tail are contiguous); comm2 given to the kernels:
//   sharded. NULL: single GPU, the partial sums are read where they were left.
// The dog-leg step from the current point: dogleg_choose.hpp. As a launch of its own where the evaluation that
// follows has no prologue launch to carry it (problems without boards, the splined models, the protocol driver)
__global__ __launch_bounds__(PRO_T)
void step2_choose_kernel(ChooseArgs a)
{
    __shared__ double scratch[17*7];
    const ChooseOut c = dogleg_choose_scalars(a, scratch);
    dogleg_choose_elementwise(a, c, blockIdx.x*blockDim.x + threadIdx.x);
    if(blockIdx.x == 0 && threadIdx.x == 0) dogleg_choose_record(a, c);
}


// the same around the multi-launch Cholesky of big camera blocks
__global__ __launch_bounds__(1024)
void step2_finish_kernel(Step2Dev sd, int* chol_status)
{
    (void)step2_finish(sd, chol_status);
}
__global__ __launch_bounds__(64)
void step2_post_kernel(Step2Dev sd, const int* __restrict__ chol_status)
{
    if(threadIdx.x == 0 && !sd.fl->skip_chol) step2_chol_done(sd, *chol_status != 0);
}

// After the factorization, side by side in one launch (256 threads):
//   workgroups [0, nbs)   back-substitution d_e = -L^-T (y_e + Wt_e d_s), one WAVE per E block; each block
//                         leaves (|d_e|^2, d_e . g_e) in dots_part[block]
//   workgroup  nbs        d_s into the state-ordered step
//   the rest              the quadratic form g^T N g of a new current point (ctl->derive): per-workgroup
//                         partials into qf_part[.][0], and the frame/point part of |g|^2 into qf_part[.][2]
__global__ __launch_bounds__(256)
void step2_backsub_quadform_kernel(NormalDims nd, BlockRanges br, const OpDev* __restrict__ ops,
                                   const SolverCtl* __restrict__ ctl, const SolverCtlFlags* __restrict__ fl,
                                   const double* __restrict__ Wt, const double* __restrict__ LD,
                                   const double* __restrict__ y, const double* __restrict__ ds,
                                   double* __restrict__ dots_part, double* __restrict__ qf_part, int nbs,
                                   SolverCtl* __restrict__ snap, const unsigned* __restrict__ occ, int nocc)
{
    const OpDev& O = ops[ctl->ib];
    const int b = blockIdx.x;
    // The control block is final for this step (its last writer is the launch before this one): the
    // host's snapshot of it is written straight into pinned memory. As a hipMemcpyAsync it was a copy
    // kernel of its own behind every step, 4-6 us on the stream
    if(snap != NULL && b == 0 && threadIdx.x < (int)(sizeof(SolverCtl)/sizeof(int)))
        ((int*)snap)[threadIdx.x] = ((const int*)ctl)[threadIdx.x];
    if(b > nbs)
    {
        if(!ctl->derive) return;
        const int qb = b - nbs - 1;
        const double mine = quadform_body(nd, O, O.g, qb, true, occ, nocc);
        if(threadIdx.x < 3) qf_part[4*qb + threadIdx.x] = mine;
        return;
    }
    if(fl->skip_backsub) return;
    double* __restrict__ step = O.step_gn;
    if(b == nbs)
    {
        for(int i = threadIdx.x; i < nd.Nc; i += blockDim.x)
            step[S_to_state(nd, i)] = ds[i];
        return;
    }
    backsub_eblock(nd, br, O, Wt, LD, y, ds, dots_part, 4*b + (int)(threadIdx.x >> 6), occ, nocc, [] { return true; });
}

// sharded: this rank's summands of comm2, each summed in a fixed order. One workgroup
__global__ __launch_bounds__(256)
void step2_pack2_kernel(const SolverCtl* __restrict__ ctl, const SolverCtlFlags* __restrict__ fl,
                        const double* __restrict__ qf_part, int qf_n,
                        const double* __restrict__ dots_part, int dots_n, double* __restrict__ comm2)
{
    __shared__ double scratch[17*7];
    double o[2] = {0.0, 0.0}, o2[2] = {0.0, 0.0};
    if(ctl->derive)
        block_sum_fixed<2>(qf_n, [&](int i, double (&t)[2]) { t[0] = qf_part[4*i]; t[1] = qf_part[4*i + 2]; }, o, scratch);
    if(!fl->skip_backsub)
        block_sum_fixed<2>(dots_n, [&](int i, double (&t)[2]) { t[0] = dots_part[2*i]; t[1] = dots_part[2*i + 1]; }, o2, scratch);
    if(threadIdx.x == 0)
    {
        comm2[COMM2_GNG] = o[0]; comm2[COMM2_GGE] = o[1]; comm2[COMM2_GNE2] = o2[0]; comm2[COMM2_GNE_GE] = o2[1];
    }
}


// Phase 2: dense Cholesky of the (summed) Schur complement, d_S, and the
// back-substitution of the local E blocks into step_gn of the point
hipError_t launch_solve_backsub(const NormalDims& nd, const BlockRanges& br,
                                const OpRef& R, const FactorBuffers& F, const int* skip_also, bool keep_factor,
                                hipStream_t stream)
{
    {
        const int n = nd.Nc;
        if(chol_fits_lds(n))
        {
            Step2Dev none; memset(&none, 0, sizeof(none));
            launch_cholesky_lds(0, n, R.skip, keep_factor ? 1 : 0, F.S, F.r, F.status, none, stream);
        }
        else if(F.Linv != NULL)
            launch_cholesky_large(n, R.skip, F.S, F.Linv, F.status, stream, NULL, NULL, NULL, NULL, 0, NULL, NULL, false, F.use_sweep != 0, F.diag_minmax);
        else
            launch_cholesky_global(n, R.skip, F.S, F.r, F.status, stream);
    }
    hipLaunchKernelGGL(backsub_kernel, dim3(br.count()+1), dim3(64), 0, stream,
                       nd, br, R, skip_also, F.Wt, F.LD, F.y, F.r);
    return hipGetLastError();
}

static int quadform_blocks(const NormalDims& nd)
{
    return (nd.Nc + nd.NE + 4*QF_ROWS_PER_WAVE - 1)/(4*QF_ROWS_PER_WAVE);
}
hipError_t launch_quadform(const NormalDims& nd, const OpRef& R, const double* v, double* out,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(quadform_kernel, dim3(quadform_blocks(nd)), dim3(256), 0, stream,
                       nd, R, v, 0, out, 0, 1);
    return hipGetLastError();
}
hipError_t launch_dot(int n, const double* a, const double* b, double* out, hipStream_t stream)
{
    int nb = (n + 255)/256; if(nb > 1024) nb = 1024; if(nb < 1) nb = 1;
    hipLaunchKernelGGL(dot_kernel, dim3(nb), dim3(256), 0, stream, n, a, b, out);
    return hipGetLastError();
}
hipError_t launch_axpby(int n, double alpha, const double* a, double beta, const double* b, double* y,
                        hipStream_t stream)
{
    if(n <= 0) return hipSuccess;
    hipLaunchKernelGGL(axpby_kernel, dim3((n+255)/256), dim3(256), 0, stream, n, alpha, a, beta, b, y);
    return hipGetLastError();
}
size_t outlier_partial_doubles() { return OUTLIER_BLOCKS; }
hipError_t launch_outlier_stats(int Npoints_board, double thresh_sq, const double* x, const double* pool,
                                int* counts, double* sums, double* part, hipStream_t stream)
{
    if(Npoints_board <= 0) return hipSuccess;
    int nb = (Npoints_board + 255)/256; if(nb > OUTLIER_BLOCKS) nb = OUTLIER_BLOCKS;
    hipLaunchKernelGGL(outlier_stats_kernel, dim3(nb), dim3(256), 0, stream,
                       Npoints_board, thresh_sq, x, pool, counts, part);
    hipLaunchKernelGGL(outlier_stats_sum_kernel, dim3(1), dim3(64), 0, stream, nb, part, sums);
    return hipGetLastError();
}
hipError_t launch_mark_outliers(int Npoints_board, double thresh_sq, const double* x, double* pool,
                                int* counts, hipStream_t stream)
{
    if(Npoints_board <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_outliers_kernel, dim3((Npoints_board+255)/256), dim3(256), 0, stream,
                       Npoints_board, thresh_sq, x, pool, counts);
    return hipGetLastError();
}

// ---- the device-controlled step. ctl is followed in memory by its SolverCtlFlags
static SolverCtlFlags* ctl_flags(SolverCtl* ctl) { return (SolverCtlFlags*)(ctl + 1); }
const int* solver_ctl_skip_factor(const SolverCtl* ctl) { return &((const SolverCtlFlags*)(ctl + 1))->skip_factor; }
const int* solver_ctl_skip_eval  (const SolverCtl* ctl) { return &((const SolverCtlFlags*)(ctl + 1))->skip_eval; }
size_t     solver_ctl_bytes() { return sizeof(SolverCtl) + sizeof(SolverCtlFlags); }
// host image of [ctl | flags] before a run: the first thing the fused step does is
// the assembly + elimination of the starting point icur
void       solver_ctl_init_flags(void* ctl_image, int icur)
{
    SolverCtlFlags* fl = (SolverCtlFlags*)((SolverCtl*)ctl_image + 1);
    memset(fl, 0, sizeof(*fl));
    fl->elim_mode = 1; fl->elim_sel = icur; fl->skip_chol = 1; fl->skip_backsub = 1;
}

// what this shard does not own of a state vector, zeroed: the sum over the shards is then the state
__global__ __launch_bounds__(256)
void mask_state_kernel(NormalDims nd, BlockRanges br, int is_leader, double* __restrict__ b)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= nd.Nstate) return;
    bool mine;
    const int se = state_to_SE(nd, i);
    if(se >= 0) mine = is_leader != 0;        // the camera block
    else
    {
        const int e = -se - 1;
        int lo, hi, lo2, hi2;
        br.e_range(nd, 0, &lo, &hi);
        br.e_range(nd, 1, &lo2, &hi2);
        mine = (e >= lo && e < hi) || (e >= lo2 && e < hi2);
    }
    if(!mine) b[i] = 0.0;
}
hipError_t launch_mask_state(const NormalDims& nd, const BlockRanges& br, bool is_leader, double* b, hipStream_t stream)
{
    hipLaunchKernelGGL(mask_state_kernel, dim3((nd.Nstate + 255)/256), dim3(256), 0, stream, nd, br, is_leader ? 1 : 0, b);
    return hipGetLastError();
}

// ---- the fused step
static int quadform_blocks(const NormalDims& nd);
const int* solver_ctl_skip_eval2(const SolverCtl* ctl) { return &((const SolverCtlFlags*)(ctl + 1))->skip_eval; }

ChooseArgs step2_choose_args(const Step2Args& a)
{
    ChooseArgs c;
    c.nd = *a.nd; c.ops = a.ops; c.ctl = a.ctl; c.fl = ctl_flags(a.ctl); c.chol_status = a.F->status; c.step = a.step;
    c.qf_part = a.plan->qf_part; c.qf_n = quadform_blocks(*a.nd); c.dots_part = a.plan->dots_part; c.dots_n = a.br->count();
    c.comm2 = a.comm2;
    return c;
}
hipError_t launch_step2_choose(const Step2Args& a, hipStream_t stream)
{
    const NormalDims& nd = *a.nd;
    // (workgroups of 64, like the prologue's when the choice rides there: the fixed-order sums depend on the
    //  workgroup size, and the ranks of a sharded solve - with or without boards in their shard - must get the same bits)
    hipLaunchKernelGGL(step2_choose_kernel, dim3((nd.Nstate + PRO_T - 1)/PRO_T), dim3(PRO_T), 0, stream, step2_choose_args(a));
    return hipGetLastError();
}

// (launch_step2_assemble left work on the side stream: launch_step2_reduce, which always follows it, joins)
static thread_local bool step2_side_pending = false;
// the block normal equations of the point the flags name, and the elimination of its frame/point blocks
hipError_t launch_step2_assemble(const Step2Args& a, bool initial, hipStream_t stream)
{
    step2_side_pending = false;
    const DeviceProblem& P = *a.P;
    const NormalDims& nd = *a.nd;
    const BlockRanges& br = *a.br;
    SolverCtlFlags* fl = ctl_flags(a.ctl);
    const int* sel_eval = initial ? &a.ctl->ib : &a.ctl->ia;
    const bool by_rows = (P.lens_type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC);
    int nframes_fused = 0;
    if(P.Nobs_board > 0 && !by_rows)
    {
        const int row0 = 2*P.W*P.H*P.Nobs_board;
        nframes_fused = br.frame_hi - br.frame_lo;
        // (round 6) the planned rows of the evaluated point are the launch's last workgroups (a trial without an evaluation
        // skips them: elim_mode != 1). Their sums are added after the Grams' (launch_step2_reduce)
        const int ngen = gen_ride_blocks(*a.plan);
        hipLaunchKernelGGL(assemble_factor_kernel, dim3(nframes_fused + a.plan->Nchunks*assemble_chunk_slices(P) + assemble_row_blocks(P, *a.plan) + ngen), dim3(256),
                           assemble_lds_bytes_with_gen(nd, *a.plan), stream, P, nd, br, a.ops, sel_eval, &a.ctl->ib, a.ctl, (const int*)NULL,
                           &fl->elim_mode, 0, 1, 0.0, *a.plan, a.gram, *a.F, nframes_fused, assemble_row0(P, *a.plan), P.Nmeas, a.Jp, a.Ji, ngen);
        (void)row0;
    }
    else
    {
        // no Grams (splined models, problems without boards): the atomic row-by-row assembly of the evaluated point
        EvalBuffers B; memset(&B, 0, sizeof(B));
        B.R = OpRef{ a.ops, sel_eval, &fl->skip_asm }; B.Jp = (int32_t*)a.Jp; B.Ji = (int32_t*)a.Ji;
        bool forked = false;
        const hipError_t e = launch_assemble(P, nd, br, *a.plan, B, stream, a.side, a.ev_fork, a.ev_join, &forked);
        if(e != hipSuccess) return e;
        step2_side_pending = forked;
    }
    // the blocks the fused kernel did not eliminate: all of them on the row-by-row
    // path; the point blocks otherwise (their rows are accumulated in the same launch)
    const int nrest = br.count() - nframes_fused;
    if(nrest > 0)
    {
        const OpRef R = { a.ops, &fl->elim_sel, &fl->skip_elim };
        // (the occupancy of Wt's tiles is tracked only when EVERY block comes through here: F.occ is only
        //  allocated for the splined models, whose blocks all do)
        hipLaunchKernelGGL(eblock_factor_kernel, dim3(nrest), dim3(nd.Nc > 255 ? 256 : 64), 0, stream,
                           nd, br, nframes_fused, R, 0.0, a.ctl, a.F->Wt, a.F->LD, a.F->y, a.F->status,
                           (nframes_fused == 0) ? a.F->occ : (unsigned*)NULL, occ_words(nd),
                           (nframes_fused == 0) ? a.F->Wtile : (double*)NULL);
    }
    return hipGetLastError();
}

// SYRK (+ finalize of A, g, |x|^2) | S, r and the tail of comm1. (Sharded: comm1 is all-reduced after this)
// Does the end-of-trial logic (step2_finish) ride in the reduction's launch (round 5)? On a single GPU the tail it reads -
// g_S, |x|^2, the block elimination's status - is complete when the reduction's last workgroup has written it, and that
// workgroup can decide the trial there and then, beside the others: the factorization's first launch starts on its matrix
// at once (and may be several workgroups: the dissection's). Sharded, the tail is summed over the ranks behind this launch.
// (With the backward sweep - FactorBuffers::use_sweep - the end-of-trial logic and the verdict are launches of their own.)
static bool step2_finish_rides(const Step2Args& a)
{
    return a.comm2 == NULL && !a.F->use_sweep;
}
hipError_t launch_step2_reduce(const Step2Args& a, hipStream_t stream, int initial)
{
    const DeviceProblem& P = *a.P;
    const NormalDims& nd = *a.nd;
    const BlockRanges& br = *a.br;
    const FactorBuffers& F = *a.F;
    SolverCtlFlags* fl = ctl_flags(a.ctl);
    const bool with_grams = P.Nobs_board > 0 && P.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC;
    FinalizeRide ride; memset(&ride, 0, sizeof(ride));
    if(with_grams && a.plan->Ndest > 0)
    {
        ride.npos = gram_stride(P.Ndist); ride.ops = a.ops; ride.sel = &fl->elim_sel; ride.skip = &fl->skip_asm; ride.plan = *a.plan;
    }
    const unsigned char* live = NULL;
    const int nslots = launch_syrk(nd, br, &fl->skip_elim, F, ride.npos ? &ride : NULL, stream, &live);
    if(with_grams)
    {
        // (after the ride: one adder per destination at a time)
        const hipError_t e = launch_gen_finalize(nd, *a.plan, OpRef{ a.ops, &fl->elim_sel, &fl->skip_asm }, stream);
        if(e != hipSuccess) return e;
    }
    // A, g of the camera block and |x|^2 may still be on their way on the side stream
    if(step2_side_pending)
    {
        const hipError_t e = hipStreamWaitEvent(stream, a.ev_join, 0);
        if(e != hipSuccess) return e;
        step2_side_pending = false;
    }
    const int nb = (nd.Nc + 15)/16, npairs = nb*(nb+1)/2;
    const int nred = ((npairs*256 + nb*16)*(live ? 1 : SRED_SPLIT) + 255)/256;
    // (the dissection: its matrices' borders and pads by nfill more workgroups; where its launches follow, the end of the
    //  trial step rides in this launch's last ordinary workgroup - launch_step2_factor() then leaves it out)
    const bool rides = initial >= 0 && step2_finish_rides(a);
    const bool nd_on = F.ndMA != NULL && F.cperm_cur != NULL;
    const bool nd_launches = nd_on && F.nd_lim.rounds > 0 && rides;
    int nfill = 0;
    if(nd_launches)
    {
        // (a thread an entry of what the dissection's matrices hold beside the camera block's entries: step2_reduce_kernel)
        const long long NSp = F.nd_lim.ns_max, nxm = (long long)ND_PANEL*F.nd_lim.rounds, pads = ND_PANEL - 1;
        const long long per = (NSp + 1)*NSp + pads*nxm + (NSp + 1)*pads;
        nfill = (int)((2*per + 255)/256);
        const int ncopy = (int)((std::max<long long>(2*nd.Nc + 1, (long long)nd_plan_ints(nd.Nc)) + 255)/256);
        nfill = std::max(nfill, ncopy);
    }
    Step2Dev sd; memset(&sd, 0, sizeof(sd));
    if(rides) { sd.nd = nd; sd.ops = a.ops; sd.ctl = a.ctl; sd.fl = fl; sd.initial = initial ? 1 : 0; sd.comm1_tail = F.r + nd.Nc; }
    hipLaunchKernelGGL(step2_reduce_kernel, dim3(nred + 1 + nfill), dim3(256), 0, stream,
                       nd, a.ops, a.ctl, fl, a.is_leader ? 1 : 0, nred, nslots, F.Spart, F.S, F.r, F.status, live, F.cperm_cur, F.iso, &a.ctl->error,
                       nd_launches ? F.ndMA : (double*)NULL, nd_launches ? F.ndMB : (double*)NULL, nd_on ? F.ndp_cur : (int*)NULL, nfill,
                       rides ? 1 : 0, sd,
                       // (a packed copy of S for the one-workgroup Cholesky's launch behind this one - launch_cholesky_lds_quadform())
                       (rides && chol_fits_lds(nd.Nc)) ? factor_S_packed(F, nd.Nc) : (double*)NULL);
    return hipGetLastError();
}
int64_t step2_comm1_doubles(const NormalDims& nd) { return (int64_t)nd.Nc*nd.Nc + 2*nd.Nc + 2; }

// finish + Cholesky | back-substitution + quadratic form | (sharded) this rank's summands of comm2
hipError_t launch_step2_factor(const Step2Args& a, bool initial, hipStream_t stream)
{
    const NormalDims& nd = *a.nd;
    const BlockRanges& br = *a.br;
    const FactorBuffers& F = *a.F;
    SolverCtlFlags* fl = ctl_flags(a.ctl);
    Step2Dev sd;
    sd.nd = nd; sd.ops = a.ops; sd.ctl = a.ctl; sd.fl = fl; sd.initial = initial ? 1 : 0;
    sd.comm1_tail = F.r + nd.Nc;
    bool qf_rode = false;
    {
        const int n = nd.Nc;
        // (round 5, single GPU: the end-of-trial logic has run in the reduction's launch - step2_finish_rides())
        const bool finish_done = step2_finish_rides(a);
        if(chol_fits_lds(n))
        {
            // (round 6, single GPU: the quadratic form's workgroups in the factorization's launch - they never needed it)
            if(finish_done) { launch_cholesky_lds_quadform(n, nd, F, sd, a.plan->qf_part, quadform_blocks(nd), stream); qf_rode = true; }
            else            launch_cholesky_lds(1, n, (const int*)NULL, 0, F.S, F.r, F.status, sd, stream);
        }
        else
        {
            // (round 5: finish and post ride in the factorization's first and last launch; with the backward sweep the
            //  factorization's last launch is another: launches of their own then)
            const bool separate = F.use_sweep != 0;
            bool fused = false;
            if(separate) hipLaunchKernelGGL(step2_finish_kernel, dim3(1), dim3(1024), 0, stream, sd, F.status);
            // (the splined models: the camera block as the reduction left it - without the control points no board covers)
            LcholCompact cp; memset(&cp, 0, sizeof(cp));
            const bool compact = F.cperm_cur != NULL;       // (what the reduction went by; never with the backward sweep: solver.cpp)
            if(compact) { cp.cperm = F.cperm_cur; cp.iso = F.iso; cp.dout = F.r; cp.Nc = n; }
            // (the dissection's launches, where the host has provided for them: learn_likely_size())
            const bool nd_launches = compact && F.ndMA != NULL && F.nd_lim.rounds > 0 && finish_done;
            LcholNdLaunch nds; memset(&nds, 0, sizeof(nds));
            if(nd_launches)
            {
                const int* h = F.ndp_cur;
                nds.A = LcholChain{ F.ndMA, F.ndLinvA, h + NDH_NA, h + NDH_NS };
                nds.B = LcholChain{ F.ndMB, F.ndLinvB, h + NDH_NB, h + NDH_NS };
                nds.ndh = h; nds.lim = F.nd_lim;
                cp.ndh = h; cp.ndMA = F.ndMA; cp.ndMB = F.ndMB; cp.ndpart = F.ndPart;
            }
            launch_cholesky_large(n, &fl->skip_chol, F.S, F.Linv, F.status, stream, separate ? NULL : &sd, &fused,
                                  compact ? (nd_launches ? F.ndp_cur + NDH_NSEFF : F.cperm_cur + 2*n) : (const int*)NULL,
                                  compact ? &cp : (const LcholCompact*)NULL,
                                  compact ? (nd_launches ? F.nd_likely_panels : F.lchol_likely_panels) : 0,
                                  compact ? (unsigned*)(F.cperm_cur + 2*n + 1) : (unsigned*)NULL, nd_launches ? &nds : (const LcholNdLaunch*)NULL,
                                  finish_done, F.use_sweep != 0, F.diag_minmax);
            if(separate) hipLaunchKernelGGL(step2_post_kernel, dim3(1), dim3(64), 0, stream, sd, F.status);
            else if(!fused) return hipErrorInvalidValue;
        }
    }
    const int nbs = (br.count() + 3)/4, nqf = quadform_blocks(nd);
    hipLaunchKernelGGL(step2_backsub_quadform_kernel, dim3(nbs + 1 + (qf_rode ? 0 : nqf)), dim3(256), 0, stream,
                       nd, br, a.ops, a.ctl, fl, F.Wt, F.LD, F.y, F.r, a.plan->dots_part, a.plan->qf_part, nbs, a.snap,
                       (nd.Nc > SYRK_STRIP_FROM) ? F.occ : (const unsigned*)NULL, occ_words(nd));
    if(a.comm2 != NULL)
        hipLaunchKernelGGL(step2_pack2_kernel, dim3(1), dim3(256), 0, stream,
                           a.ctl, fl, a.plan->qf_part, nqf, a.plan->dots_part, br.count(), (double*)a.comm2);
    return hipGetLastError();
}


} // namespace mrcal_amd
