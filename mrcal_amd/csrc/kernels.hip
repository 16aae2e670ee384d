// HIP kernels (gfx950 / CDNA4) for the optimizer_callback() hot path:
// residuals x and the CSR Jacobian J of a calibration problem.
//
// Reference behaviour being reproduced: mrcal.c:4444-5970 optimizer_callback()
// (board loop :4603-4898, point loop :4902-5176, regularization :5655-5955) on
// top of project() mrcal.c:2572-2863 and the lens models (opencv.c:50-152,
// mrcal.c:1436-1858).
//
// Execution plan per evaluation:
//   1. board_prologue_kernel   one LANE per board observation: compose the
//      camera and frame poses (6-variable forward-mode duals), Rodrigues
//      matrix + its 27 partials, and fold the chain rule through the joint
//      rotation. 84 doubles per observation. This is wave-uniform data for
//      step 2; computing it redundantly in all 64 lanes of every wave would
//      cost more than the whole HBM budget of the evaluation.
//   2. board_kernel            one WAVEFRONT (64 lanes) per board observation,
//      one lane per chessboard corner. Each lane projects its corner and forms
//      its two CSR rows in registers; rows go through a 64-row LDS tile (odd
//      row stride => conflict-free ds_write_b64), half a pass at a time, and
//      the tile is streamed to HBM as one contiguous, 16-byte-per-lane
//      coalesced run: consecutive CSR rows of one observation are adjacent in
//      J's value array, so the whole 2*W*H x k block is ONE contiguous extent.
//   3. point_kernel / regularization_kernel: one lane per row pair / row.
//      These are a few hundred rows; they are launch-latency, not bandwidth.
//
// The CSR structure (rowptr, colidx) never changes between evaluations, so it
// is produced once by the *_structure kernels at problem creation.
#include <hip/hip_runtime.h>
#include <utility>
#include <stdlib.h>
#include "problem.hpp"
#include "device_math.hpp"
#include "lens_models.hpp"
#include "triangulation.hpp"
#include "kernels.hpp"
#include "dogleg_choose.hpp"
#include "kernels_shared.hpp"

namespace mrcal_amd {

// (the state's accessors: kernels_shared.hpp)
////////////////////////////////////////////////////////////////////////////////
// 1. prologue: joint pose + folded chain rule, one lane per board observation
////////////////////////////////////////////////////////////////////////////////
__device__ __forceinline__
void joint_pose_record(double* __restrict__ out,
                       const double* rt_cam, // NULL: camera at the reference
                       const double* rt_frame)
{
    double R[9], dR[27];
    if(rt_cam == NULL)
    {
        R_from_r_with_grad(R, dR, rt_frame);
        for(int i=0;i<9;i++) out[JOINT_R + i] = R[i];
        for(int i=0;i<3;i++) out[JOINT_T + i] = rt_frame[3+i];
        for(int j=0;j<3;j++)
            for(int i=0;i<3;i++)
                for(int l=0;l<3;l++)
                {
                    out[JOINT_MC + 9*j + 3*i + l] = 0.0;
                    out[JOINT_MF + 9*j + 3*i + l] = dR[9*i + 3*j + l];
                }
        for(int i=0;i<3;i++)
            for(int l=0;l<3;l++)
            {
                out[JOINT_DTJ_DRC + 3*i + l] = 0.0;
                out[JOINT_DTJ_DTF + 3*i + l] = (i==l) ? 1.0 : 0.0;
            }
        return;
    }

    // rj = rc o rf ; independent variables 0..2 = rc, 3..5 = rf
    Dual<6> rc[3], rf[3], rj[3];
    for(int i=0;i<3;i++)
    {
        rc[i] = Dual<6>::variable(rt_cam  [i], i);
        rf[i] = Dual<6>::variable(rt_frame[i], 3+i);
    }
    compose_r_dual<6>(rj, rc, rf);

    // tj = R(rc) tf + tc ; independent variables 0..2 = rc, 3..5 = tf
    Dual<6> tf[3], tj[3];
    for(int i=0;i<3;i++) tf[i] = Dual<6>::variable(rt_frame[3+i], 3+i);
    rotate_point_r_dual<6>(tj, rc, tf, false);

    double rjv[3];
    for(int i=0;i<3;i++) rjv[i] = rj[i].x;
    R_from_r_with_grad(R, dR, rjv);

    for(int i=0;i<9;i++) out[JOINT_R + i] = R[i];
    for(int i=0;i<3;i++) out[JOINT_T + i] = tj[i].x + rt_cam[3+i];
    for(int j=0;j<3;j++)
        for(int i=0;i<3;i++)
            for(int l=0;l<3;l++)
            {
                double mc = 0.0, mf = 0.0;
                for(int k=0;k<3;k++)
                {
                    mc += dR[9*i + 3*j + k] * rj[k].d[l];
                    mf += dR[9*i + 3*j + k] * rj[k].d[3+l];
                }
                out[JOINT_MC + 9*j + 3*i + l] = mc;
                out[JOINT_MF + 9*j + 3*i + l] = mf;
            }
    for(int i=0;i<3;i++)
        for(int l=0;l<3;l++)
        {
            out[JOINT_DTJ_DRC + 3*i + l] = tj[i].d[l];
            out[JOINT_DTJ_DTF + 3*i + l] = tj[i].d[3+l];
        }
}

// The same record by PRO_LPO lanes (round 6): forward-mode differentiation is one independent computation per
// tangent direction, so lane l < 6 runs the pose chain on Dual<1> numbers seeded with variable l - the instructions
// Dual<6> runs for component l, on the same operands: THE SAME BITS in every entry - and every lane runs the values.
// What a lane leaves: column l of Mc (l < 3) or column l - 3 of Mf and of the translation's partials; lane 0 also
// R and t. A lane's chain is the values' (three sincos, an acos, the square roots and divisions) plus ONE tangent
// instead of six
__device__ __forceinline__
void joint_pose_record_lanes(double* __restrict__ out,
                             const double* rt_cam, // NULL: camera at the reference
                             const double* rt_frame, const int l)
{
    double R[9], dR[27];
    if(rt_cam == NULL)
    {
        R_from_r_with_grad(R, dR, rt_frame);
        if(l == 0)
        {
            for(int i=0;i<9;i++) out[JOINT_R + i] = R[i];
            for(int i=0;i<3;i++) out[JOINT_T + i] = rt_frame[3+i];
        }
        // (ll a constant of the unrolled loop: dR stays in registers)
#pragma unroll
        for(int ll=0;ll<3;ll++)
            if(l == ll)
            {
                for(int j=0;j<3;j++)
                    for(int i=0;i<3;i++)
                    {
                        out[JOINT_MC + 9*j + 3*i + ll] = 0.0;
                        out[JOINT_MF + 9*j + 3*i + ll] = dR[9*i + 3*j + ll];
                    }
                for(int i=0;i<3;i++)
                {
                    out[JOINT_DTJ_DRC + 3*i + ll] = 0.0;
                    out[JOINT_DTJ_DTF + 3*i + ll] = (i==ll) ? 1.0 : 0.0;
                }
            }
        return;
    }
    // rj = rc o rf ; this lane's variable: l in 0..2 = rc[l], 3..5 = rf[l-3] (none for the lanes past 5)
    Dual<1> rc[3], rf[3], rj[3];
    for(int i=0;i<3;i++)
    {
        rc[i] = Dual<1>(rt_cam  [i]); rc[i].d[0] = (l == i)   ? 1.0 : 0.0;
        rf[i] = Dual<1>(rt_frame[i]); rf[i].d[0] = (l == 3+i) ? 1.0 : 0.0;
    }
    compose_r_dual<1>(rj, rc, rf);

    // tj = R(rc) tf + tc ; the lane's variable: 0..2 = rc, 3..5 = tf
    Dual<1> tf[3], tj[3];
    for(int i=0;i<3;i++) { tf[i] = Dual<1>(rt_frame[3+i]); tf[i].d[0] = (l == 3+i) ? 1.0 : 0.0; }
    rotate_point_r_dual<1>(tj, rc, tf, false);

    double rjv[3];
    for(int i=0;i<3;i++) rjv[i] = rj[i].x;
    R_from_r_with_grad(R, dR, rjv);

    if(l == 0)
    {
        for(int i=0;i<9;i++) out[JOINT_R + i] = R[i];
        for(int i=0;i<3;i++) out[JOINT_T + i] = tj[i].x + rt_cam[3+i];
    }
    if(l >= 6) return;
    // column l of Mc, or column l - 3 of Mf: M[j][i] = sum_k dR[i][j][k] drj[k]/dvar_l, in Dual<6>'s order of k
    const int M0 = (l < 3) ? JOINT_MC + l : JOINT_MF + (l - 3);
    for(int j=0;j<3;j++)
        for(int i=0;i<3;i++)
        {
            double mm = 0.0;
            for(int k=0;k<3;k++)
                mm += dR[9*i + 3*j + k] * rj[k].d[0];
            out[M0 + 9*j + 3*i] = mm;
        }
    const int T0 = (l < 3) ? JOINT_DTJ_DRC + l : JOINT_DTJ_DTF + (l - 3);
    for(int i=0;i<3;i++) out[T0 + 3*i] = tj[i].d[0];
}

// (Measured alternative, not used: R(rc o rf) = R(rc) R(rf) gives the same record
// from two R_from_r with gradients and ~230 multiply-adds, 3-4 us less for this
// kernel. It is the more accurate route where the reference's is ill-conditioned
// - compositions near a multiple of a full turn - and therefore does NOT
// reproduce the reference there: small Jacobian entries differ by 1e-10 absolute,
// 2e-5 by the reference's relative-error measure. Parity first: the reference's
// route is followed literally)

template<bool WITH_J, bool WITH_STRUCTURE, class BV>
__device__ __forceinline__
void regularization_row_at(const DeviceProblem& P, const BV& b, double* __restrict__ x, double* __restrict__ Jv,
                           int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx, const int i);
template<bool WITH_J, bool WITH_STRUCTURE>
__device__ __forceinline__
void regularization_row(const DeviceProblem& P, const OpRef& R,
                        int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx, const int i)
{
    const double* b = opref_get(R).b;
    regularization_row_at<WITH_J,WITH_STRUCTURE>(P, b, opref_get(R).x, opref_get(R).Jv, rowptr, colidx, i);
}

// the workgroups of the prologue launch that clear a point's normal equations: izb of nblocks_zero
// (round 6: array by array, 16 bytes a store. It was one element a store behind a chain of five compares; at
//  BASELINE configuration 2 - 58 MB of A and Bt - that was what the prologue launch waited for, not the poses)
__device__ __forceinline__
void prologue_clear(const EvalBuffers& B, const OpDev& O, const int izb, const int nblocks_zero)
{
    const long long nthreads = (long long)nblocks_zero*PRO_T;
    const long long t0 = (long long)izb*PRO_T + threadIdx.x;
    double* const arr[5] = { O.A, O.Bt, O.D, O.g, O.scalars };
#pragma unroll
    for(int q = 0; q < 5; q++)
    {
        double* __restrict__ p = arr[q];
        const long long n = B.zero_n[q], npairs = n >> 1;
        const double2 z = make_double2(0.0, 0.0);
        for(long long i = t0; i < npairs; i += nthreads) reinterpret_cast<double2*>(p)[i] = z;
        if((n & 1) && t0 == 0) p[n - 1] = 0.0;
    }
}
// Workgroups (PRO_T threads), in order: [pose records, PRO_LPO lanes per observation: joint_pose_record_lanes] [unpacking of the
// intrinsics and the warp] [clearing of the normal equations, if asked for]
// [regularization rows, one per thread: reg_mode 0 = x only, 1 = x and J, -1 = none]
// [CHOOSE: the dog-leg step, a state variable per thread].
// Only the first kind is long; the others are independent of it and of each
// other and ride along instead of costing launches of their own.
// CHOOSE (the solver's trial step): the launch also CHOOSES the trial point it evaluates (dogleg_choose.hpp) -
// every workgroup derives the step's scalars for itself, the pose / unpack / regularization workgroups compute
// the entries of the trial state they need from them (TrialState: the same instructions, hence the same bits,
// as the workgroups at the end of the grid that write the step and the trial state out), and whether the trial
// evaluates anything at all is taken from the derived numbers, not from the flag being written
template<class BV>
__device__ __forceinline__
void board_prologue_body(const DeviceProblem& P, const EvalBuffers& B, const BV& b, const OpDev& O,
                         int nblocks_unpack, int nblocks_zero, int reg_mode)
{
    double* __restrict__ joint = B.joint;
    const int nblocks_obs = prologue_obs_blocks(P.Nobs_board);
    if((int)blockIdx.x >= nblocks_obs + nblocks_unpack + nblocks_zero)
    {
        const int i = ((int)blockIdx.x - (nblocks_obs + nblocks_unpack + nblocks_zero))*PRO_T + threadIdx.x;
        if(reg_mode == 1)      regularization_row_at<true, false>(P, b, O.x, O.Jv, (int32_t*)NULL, (int32_t*)NULL, i);
        else if(reg_mode == 0) regularization_row_at<false,false>(P, b, O.x, O.Jv, (int32_t*)NULL, (int32_t*)NULL, i);
        return;
    }
    // the blocks past the observations unpack the intrinsics of every camera
    // and the board warp from the packed state (or copy the seeds); the blocks
    // past those clear the point's normal equations (a bandwidth job that runs
    // next to the long dependent chains of the pose blocks instead of costing a
    // launch of its own)
    if((int)blockIdx.x >= nblocks_obs + nblocks_unpack)
    {
        prologue_clear(B, O, (int)blockIdx.x - nblocks_obs - nblocks_unpack, nblocks_zero);
        return;
    }
    if((int)blockIdx.x >= nblocks_obs)
    {
        double* __restrict__ u = joint + (size_t)P.Nobs_board*JOINT_STRIDE;
        const int n = P.Ncameras_intrinsics*P.Nintrinsics;
        const int i = (blockIdx.x - nblocks_obs)*blockDim.x + threadIdx.x;
        if(i < n)
        {
            const int icam = i / P.Nintrinsics;
            u[i] = get_intrinsic(P, b, icam, i - icam*P.Nintrinsics);
        }
        else if(i < n + 2)
        {
            double w[2] = {0.0, 0.0};
            if(P.has_warp_seed) get_warp(w, P, b);
            u[i] = w[i - n];
        }
        return;
    }

    // PRO_LPO lanes an observation (round 6; one lane did all of it until round 5: ~3000 dependent instructions, the
    // launch as long as that chain whatever the number of observations)
    const int lpo  = prologue_lanes(P.Nobs_board);
    const int iobs = (int)blockIdx.x*(PRO_T/lpo) + (int)threadIdx.x/lpo;
    const int l    = (int)threadIdx.x % lpo;
    if(iobs >= P.Nobs_board) return;
    const BoardObsMeta m = P.board_meta[iobs];

    double rt_frame[6], rt_cam[6];
    get_rt_ref_frame(rt_frame, P, b, m.iframe);
    double* out = joint + (size_t)iobs*JOINT_STRIDE;
    if(lpo == 1)
    {
        // (many observations: a lane each, as until round 5)
        double rec[JOINT_REC];
        if(m.icam_extrinsics >= 0)
        {
            get_rt_cam_ref(rt_cam, P, b, m.icam_extrinsics);
            joint_pose_record(rec, rt_cam, rt_frame);
        }
        else
            joint_pose_record(rec, NULL, rt_frame);
        for(int i=0;i<JOINT_REC;i++) out[i] = rec[i];
    }
    else if(m.icam_extrinsics >= 0)
    {
        get_rt_cam_ref(rt_cam, P, b, m.icam_extrinsics);
        joint_pose_record_lanes(out, rt_cam, rt_frame, l);
    }
    else
        joint_pose_record_lanes(out, NULL, rt_frame, l);
    // splined models: the observation's box of control points starts empty (board_splined_kernel fills it)
    if(O.spl_box != NULL && l == 0) ((int4*)O.spl_box)[iobs] = make_int4(0x7fffffff, -1, 0x7fffffff, -1);
}
template<bool CHOOSE>
__global__ __launch_bounds__(PRO_T)
void board_prologue_kernel(DeviceProblem P, EvalBuffers B, int nblocks_unpack, int nblocks_zero, int reg_mode,
                           int nblocks_reg, ChooseArgs ca)
{
    const OpRef R = B.R;
    if constexpr(CHOOSE)
    {
        // (round 6) The workgroups that clear the trial point's normal equations need nothing of the choice but which point
        // that is - ctl->ia, which the choice does not touch: they do not derive the step's scalars. They were 1024 of the
        // launch's 1200 workgroups at the metric's size, each reading the 768 + 1000 partial sums of the quadratic form
        // and the back-substitution after an accepted step: 38 MB out of the L2s, 20.4 us a launch against 14.7 after a
        // rejected one (tools/exp/prologue_durations.py). A trial that evaluates nothing (a voided step, a step below
        // the threshold) now has its buffers cleared all the same: nothing reads them
        {
            const int z0 = prologue_obs_blocks(P.Nobs_board) + nblocks_unpack;
            if((int)blockIdx.x >= z0 && (int)blockIdx.x < z0 + nblocks_zero)
            {
                if(ca.ctl->done) return;
                prologue_clear(B, ca.ops[ca.ctl->ia], (int)blockIdx.x - z0, nblocks_zero);
                return;
            }
        }
        __shared__ double scratch[17*7];
        const ChooseOut c = dogleg_choose_scalars(ca, scratch);
        const int first = prologue_obs_blocks(P.Nobs_board) + nblocks_unpack + nblocks_zero + nblocks_reg;
        if((int)blockIdx.x >= first)
        {
            dogleg_choose_elementwise(ca, c, ((int)blockIdx.x - first)*PRO_T + threadIdx.x);
            if((int)blockIdx.x == first && threadIdx.x == 0) dogleg_choose_record(ca, c);
            return;
        }
        if(c.skip_eval) return;
        // (ia: not touched by the choice)
        board_prologue_body(P, B, dogleg_trial_state(ca, c), ca.ops[c.ia], nblocks_unpack, nblocks_zero, reg_mode);
    }
    else
    {
        if(opref_skip(R)) return;
        const double* b = opref_get(R).b;
        board_prologue_body(P, B, b, opref_get(R), nblocks_unpack, nblocks_zero, reg_mode);
    }
}

////////////////////////////////////////////////////////////////////////////////
// 2. board kernel
////////////////////////////////////////////////////////////////////////////////
//
// One wavefront per board observation, one lane per chessboard corner, 64
// corners per pass. A lane projects its corner and forms BOTH its Jacobian
// rows in registers (the x and y rows share most of the projection and all of
// the pose chain rule), at the fixed tile columns of problem.hpp.
//
// The rows then go through a 64-row LDS tile in two halves: lanes 0..31 store
// their 64 rows, the wave streams them out and accumulates their Gram; then
// lanes 32..63. A 64-row tile (14.8 KB at OPENCV8) instead of a 128-row one is
// what lets ~9 waves share a CU's 160 KB of LDS: the kernel is a mix of FP64
// arithmetic, HBM stores and exposed latencies, and only other resident waves
// can fill one wave's gaps.
//
// Copy-out: consecutive CSR rows of one observation are adjacent in J's value
// array, so the 64 x k half-tile is ONE contiguous HBM extent; it is streamed
// out 16 bytes per lane, whole rows per wave instruction (measured: this store
// pattern reaches 5.9 TB/s, a row-per-lane direct store 3.4 TB/s;
// tools/exp/store_patterns.hip). Each lane owns a fixed pair of CSR columns
// and walks down the rows, so the loop body is two LDS reads, one store and a
// few integer adds.
//
// Gram (WITH_GRAM): G = Tt T over the tile columns with v_mfma_f64_4x4x4f64
// (problem.hpp). On gfx950 the FP64 MFMAs run on the same FP64 lanes as the
// vector FMAs (measured: they do not overlap, tools/exp/launch_floor.hip), and
// the 4x4x4 form is 1.46x faster per flop than 16x16x4 and wastes no flops on
// padding a 27-column tile to 32: 350 instead of 150 MFMAs per observation,
// at 19 instead of 113 cycles each. The last tile column is the residual, so
// the last column of G is Tt x = the observation's slice of Jt x and its
// corner |x|^2.
//
// Nothing in the pass loop loads from global memory: the observed pixels are
// staged in LDS up front, the intrinsics were unpacked by the prologue kernel.
// The stores are fire-and-forget.

// CSR column c of a board row with image coordinate xy -> tile column. A
// chain of selects on wave-uniform boundaries: no divergent branches
__device__ __forceinline__
int board_csr_to_tile_col(const DeviceProblem& P, bool has_ext, int c, int xy)
{
    const int b1 = P.Ncore_state ? 2 : 0;                   // f, then c, of this row's own coordinate
    const int b2 = b1 + P.Ndist_state;
    const int b3 = b2 + (has_ext ? 6 : 0);
    const int b4 = b3 + (P.do_optimize_frames ? 6 : 0);
    int col = tile_warp0(P.Ndist) + (c - b4);
    col = (c < b4) ? tile_frame0(P.Ndist) + (c - b3) : col;
    col = (c < b3) ? tile_ext0(P.Ndist)   + (c - b2) : col;
    col = (c < b2) ? 4 + (c - b1)                    : col;
    col = (c < b1) ? 2*c + xy                        : col;
    return col;
}

// rotation within each 16-lane row: lane i takes the value of lane (i - N) mod 16
// (measured, tools/exp/dpp_ror_probe.hip)
template<int N>
__device__ __forceinline__ double row_ror_f64(double v)
{
    union { double d; int i[2]; } u; u.d = v;
    // bound_ctrl with full row/bank masks: no lane keeps its old value, so the
    // destination needs no initialization
    u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x120 + N, 0xf, 0xf, true);
    u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x120 + N, 0xf, 0xf, true);
    return u.d;
}

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) d2_t   gdouble2;

// tile[a] and tile[a + 16 doubles] in one DS instruction; the caller waits (lgkmcnt)
typedef double gram_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gram_d2 lds_read2_b64_16(unsigned lds_byte_address)
{
    gram_d2 v;
    asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(v) : "v"(lds_byte_address) : "memory");
    return v;
}

// Copy-out of a FULL half-tile (64 rows) whose rows have K nonzeros, K even and
// known at compile time: a lane owns one pair of CSR columns and every R-th
// row, R = 64/(K/2) rows per wave instruction. Two row steps (2R rows) keep the
// row parity of a lane fixed, so its four LDS byte offsets (two columns for
// each of its two rows per double step) are loop invariants (A0,A1: row rsub;
// B0,B1: row rsub+R) and everything else folds into immediate offsets: no
// address arithmetic, no selects in the loop. 4 rows per lane are in flight
template<int K, int KS>
__device__ __forceinline__
void copy_out_full(const double* __restrict__ tile, gdouble* __restrict__ out,
                   int A0, int A1, int B0, int B1, unsigned gofs, int rsub)
{
    constexpr int PAIRS = K/2, R = 64/PAIRS;
    constexpr int NIT   = (64 + 2*R - 1)/(2*R);       // double steps
    const char* __restrict__ t = reinterpret_cast<const char*>(tile);
#pragma unroll
    for(int it0 = 0; it0 < NIT; it0 += 2)
    {
        d2_t v[4];
#pragma unroll
        for(int q=0;q<4;q++)
        {
            const int it = it0 + (q >> 1), u = q & 1;
            const int rb = it*2*R + u*R;                // first row of this wave instruction
            if(it >= NIT || rb >= 64) continue;
            const int o  = it*2*R*KS*(int)sizeof(double);
            v[q].x = *reinterpret_cast<const double*>(t + (u ? B0 : A0) + o);
            v[q].y = *reinterpret_cast<const double*>(t + (u ? B1 : A1) + o);
        }
#pragma unroll
        for(int q=0;q<4;q++)
        {
            const int it = it0 + (q >> 1), u = q & 1;
            const int rb = it*2*R + u*R;
            if(it >= NIT || rb >= 64) continue;
            gdouble* __restrict__ o = out + rb*K;       // wave-uniform
            // (nontemporal: the 300 MB of Jacobian values are not read again by anything in the step. As
            //  ordinary stores they went through the L2 as dirty lines and pushed the pixels and joint
            //  records - read again by every launch - out of the caches: 82 -> 76 us for the kernel,
            //  and the kernels after it find more of their inputs still cached. The Gram, which the
            //  assembly reads next, stays an ordinary store: no difference either way)
            if(rb + R - 1 < 64)      __builtin_nontemporal_store(v[q], reinterpret_cast<gdouble2*>(&o[gofs]));
            else if(rsub < 64 - rb)  __builtin_nontemporal_store(v[q], reinterpret_cast<gdouble2*>(&o[gofs]));
        }
    }
}

// the MFMAs of one Gram k-step, candidates resolved at compile time (problem.hpp)
template<int NBLK, int MM> struct GramCand
{
    static constexpr int      c    = gram_cand(NBLK, MM);
    static constexpr GramDesc d    = gram_cand_desc(NBLK, c);
    static constexpr bool     roty = gram_rotated_y(NBLK);
    static constexpr int srcA = (d.kind == GRAM_YY) ? 1 : 0;                       // X, or Y / Y_0
    static constexpr int srcB = (d.kind == GRAM_XX) ? 0 : (roty ? 1 : 1 + d.q);   // X, Y or Y_q
    static constexpr int rotB = (d.kind == GRAM_XX || roty) ? d.q : 0;
};
struct GramRd { double v[4]; };     // the LDS reads of one k-step: X, then Y or Y_0..Y_2
template<int NBLK, int NACC, int... MM>
__device__ __forceinline__ void gram_mfmas(double (&acc)[NACC], const double (&op)[4][4], std::integer_sequence<int, MM...>)
{
    ((acc[MM] = __builtin_amdgcn_mfma_f64_4x4x4f64(op[GramCand<NBLK,MM>::srcA][0],
                                                   op[GramCand<NBLK,MM>::srcB][GramCand<NBLK,MM>::rotB],
                                                   acc[MM], 0, 0, 0)), ...);
}
// One k-step of the Gram: slot rotations + MFMAs. (Rotations nobody uses are
// dropped by the compiler)
template<int NBLK, int NM>
__device__ __forceinline__ void gram_step_ops(double (&acc)[NM], const GramRd& rd)
{
    double op[4][4];    // [read][slot rotation]
#pragma unroll
    for(int g=0; g<4; g++)
    {
        op[g][0] = rd.v[g];
        op[g][1] = row_ror_f64<12>(op[g][0]);
        op[g][2] = row_ror_f64<8 >(op[g][0]);
        op[g][3] = row_ror_f64<4 >(op[g][0]);
    }
    gram_mfmas<NBLK>(acc, op, std::make_integer_sequence<int, NM>{});
}

// a's lanes 32..63 and b's lanes 0..31 trade places (a[32+i] <-> b[i]): v_permlane32_swap_b32 on both halves of the
// doubles (measured semantics: tools/exp/permlane32_swap_probe.hip)
__device__ __forceinline__ void permlane32_swap_f64(double& a, double& b)
{
    union { double d; unsigned u[2]; } ua, ub;
    ua.d = a; ub.d = b;
#pragma unroll
    for(int i = 0; i < 2; i++)
    {
        const auto r = __builtin_amdgcn_permlane32_swap(ua.u[i], ub.u[i], false, false);
        ua.u[i] = r[0]; ub.u[i] = r[1];
    }
    a = ua.d; b = ub.d;
}

// ---- explicit DS instructions and waits for the fused Gram + copy-out path
template<int OFF>
__device__ __forceinline__ double lds_read_b64_at(unsigned lds_byte_address)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_address), "n"(OFF) : "memory");
    return v;
}

// Step S of the fused loop over a FULL half-tile (64 rows = 16 k-steps) with K
// nonzeros per row (K even, compile time). The MFMAs of a k-step keep the
// FP64 pipe busy for ~160 cycles during which the wave can issue anything else:
// the copy-out of row group S (R = 64/(K/2) rows, one wave-wide 16 B/lane store)
// rides in that shadow. Every LDS instruction and wait here is explicit:
//     request the Gram operands of step S+1
//     request this lane's two tile values of row group S
//     rotations + MFMAs of step S                       (operands waited for at the end of step S-1)
//     wait for everything requested; store row group S
// Lanes past the last whole row of a group (64 % (K/2) of them) repeat the
// work of the first lanes: same address, same data.
// "these registers are written by the DS reads waited for here"
template<int NREAD>
__device__ __forceinline__ void lgkm_wait0(GramRd& a, d2_t& b)
{
    if(NREAD == 2)      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(b) :: "memory");
    else if(NREAD == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(b) :: "memory");
    else                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b) :: "memory");
}
template<int S, bool FULL, bool STORE, int NBLK, int K, int KS, int NM>
__device__ __forceinline__
void fused_step(double (&acc)[NM], GramRd& cur, const unsigned (&gram_a)[4], unsigned tile_a0,
                gdouble* __restrict__ out, int A0, int A1, int B0, int B1, unsigned gofs, int rsub, int nrows)
{
    // a partial half (FULL = false: the last corners of a board): the steps past
    // its rows are skipped, the stores are predicated on the row
    if(!FULL && 4*S >= nrows) return;
    constexpr int PAIRS = K/2, R = 64/PAIRS, NGRP = (64 + R - 1)/R;
    constexpr int NREAD = gram_nreads(NBLK);
    // (STORE = false, round 6: the Gram alone - the solve's J-free mode. The same MFMAs on the same operands in the
    //  same order: the same bits in the Gram)
    constexpr bool have_grp = STORE && S < NGRP;
    constexpr int  u  = S & 1;
    constexpr int  go = (S >> 1)*2*R*KS*(int)sizeof(double);         // LDS byte offset of the group's double step
    constexpr int  rb = S*R;                                          // its first row
    GramRd nxt = cur;
    if(S < 15)
    {
        // FULL: k-step s takes the tile rows s, s+16, s+32, s+48 (gram_copy_fused)
        constexpr int so = (S < 15 ? S+1 : 0)*(FULL ? 1 : 4)*KS*(int)sizeof(double);
#pragma unroll
        for(int q=0;q<NREAD;q++) nxt.v[q] = lds_read_b64_at<so>(gram_a[q]);
    }
    d2_t v = {0.0, 0.0};
    if(have_grp)
    {
        v.x = lds_read_b64_at<go>(tile_a0 + (unsigned)(u ? B0 : A0));
        v.y = lds_read_b64_at<go>(tile_a0 + (unsigned)(u ? B1 : A1));
    }
    if(S == 0)
    {
        // the only wait for Gram operands: NREAD + 2 requests are younger than step 0's
        d2_t none = {0.0, 0.0};
        if(!FULL || !have_grp) lgkm_wait0<NREAD>(cur, none);      // (no copy-out reads behind them: NREAD younger requests, not NREAD + 2 - wait for all)
        else if(NREAD == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur.v[0]), "+v"(cur.v[1]), "+v"(none) :: "memory");
        else if(NREAD == 3) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(cur.v[0]), "+v"(cur.v[1]), "+v"(cur.v[2]), "+v"(none) :: "memory");
        else                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(cur.v[0]), "+v"(cur.v[1]), "+v"(cur.v[2]), "+v"(cur.v[3]), "+v"(none) :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    gram_step_ops<NBLK>(acc, cur);
    __builtin_amdgcn_sched_barrier(0);
    lgkm_wait0<NREAD>(nxt, v);
    if(have_grp)
    {
        gdouble* __restrict__ o = out + rb*K;       // wave-uniform
        if(FULL)
        {
            if(rb + R - 1 < 64)      __builtin_nontemporal_store(v, reinterpret_cast<gdouble2*>(&o[gofs]));
            else if(rsub < 64 - rb)  __builtin_nontemporal_store(v, reinterpret_cast<gdouble2*>(&o[gofs]));
        }
        else if(rb + rsub < nrows)   __builtin_nontemporal_store(v, reinterpret_cast<gdouble2*>(&o[gofs]));
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
}
template<bool FULL, bool STORE, int NBLK, int K, int KS, int NM, int... S>
__device__ __forceinline__
void fused_steps(double (&acc)[NM], GramRd& cur, const unsigned (&gram_a)[4], unsigned tile_a0,
                 gdouble* __restrict__ out, int A0, int A1, int B0, int B1, unsigned gofs, int rsub, int nrows,
                 std::integer_sequence<int, S...>)
{
    (fused_step<S,FULL,STORE,NBLK,K,KS>(acc, cur, gram_a, tile_a0, out, A0, A1, B0, B1, gofs, rsub, nrows), ...);
}
template<bool FULL, bool STORE, int NBLK, int K, int KS, int NM>
__device__ __forceinline__
void gram_copy_fused(double (&acc)[NM], const double* __restrict__ tile, const int (&goffs)[4],
                     gdouble* __restrict__ out, int A0, int A1, int B0, int B1, unsigned gofs, int rsub, int nrows)
{
    constexpr int NREAD = gram_nreads(NBLK);
    static_assert(NREAD >= 2 && NREAD <= 4, "board tiles have 5..8 column blocks");
    const unsigned tile_a0 = (unsigned)(size_t)tile;
    // WHICH four tile rows make up a k-step is free (the Gram is a sum over all
    // rows). Consecutive rows (4s .. 4s+3, goffs) sit 58 dwords apart: the two
    // rows that a 32-lane half of the wave reads overlap in 26 of their 32 LDS
    // banks - 795 of the kernel's 1028 active LDS cycles were bank conflicts. A
    // full half-tile therefore takes rows s, s+16, s+32, s+48: 16 rows apart is
    // 928 = 32 (mod 64) dwords, disjoint banks. (A partial one keeps the
    // consecutive rows: it stops after ceil(nrows/4) steps.) Measured: the
    // conflict cycles halve (the rest are the tile writes and the copy-out
    // reads); the kernel's time does not move - the LDS is not what it waits for
    const unsigned row_skew = FULL ? (unsigned)((threadIdx.x >> 4)*15*KS*sizeof(double)) : 0u;
    unsigned gram_a[4];
#pragma unroll
    for(int q=0;q<4;q++) gram_a[q] = tile_a0 + row_skew + (unsigned)(goffs[q < NREAD ? q : 0]*sizeof(double));
    GramRd cur;
#pragma unroll
    for(int q=0;q<4;q++) cur.v[q] = 0.0;
#pragma unroll
    for(int q=0;q<NREAD;q++) cur.v[q] = lds_read_b64_at<0>(gram_a[q]);
    fused_steps<FULL,STORE,NBLK,K,KS>(acc, cur, gram_a, tile_a0, out, A0, A1, B0, B1, gofs, rsub, nrows,
                                std::make_integer_sequence<int, 16>{});
}

// the ablation knob of the measurement builds (problem.hpp); a compile-time 0 in the shipped library
#ifdef MRCAL_AMD_DEV
#define ABLATE(P, bits) ((P).debug_ablate & (bits))
#else
#define ABLATE(P, bits) 0
#endif
#ifdef BOARD_TS
#define TS(i) do { ts[i] = clock64(); } while(0)
#define TSACC(i, t0) do { const long long _t = clock64(); ts[i] += _t - (t0); (t0) = _t; } while(0)
#else
#define TS(i)
#define TSACC(i, t0)
#endif

// The copy-out invariants of a lane (copy_out_full) when every variable group
// is optimized: k and the CSR -> tile column map are compile-time constants
// (divisions by constants, selects on constants)
template<int K, int NDIST, int KS, bool HAS_EXT>
__device__ __forceinline__
void copy_out_invariants(int lane, int& A0, int& A1, int& B0, int& B1, unsigned& gofs, int& rsub)
{
    constexpr int PPR = K/2, RPI = 64/PPR, NACT = RPI*PPR;
    constexpr int b1 = 2, b2 = 2 + NDIST, b3 = b2 + (HAS_EXT ? 6 : 0), b4 = b3 + 6;
    const int cl = (lane < NACT) ? lane : lane - NACT;
    rsub = cl / PPR;
    const int c0 = 2*(cl - rsub*PPR);
    auto tcol = [&](int c, int xy) -> int
    {
        int col = tile_warp0(NDIST) + (c - b4);
        col = (c < b4) ? tile_frame0(NDIST) + (c - b3) : col;
        col = (c < b3) ? tile_ext0(NDIST)   + (c - b2) : col;
        col = (c < b2) ? 4 + (c - b1)                  : col;
        col = (c < b1) ? 2*c + xy                      : col;
        return col;
    };
    const int y0 = rsub & 1, y1 = (rsub + RPI) & 1;
    A0 = (rsub*KS       + tcol(c0,   y0))*(int)sizeof(double);
    A1 = (rsub*KS       + tcol(c0+1, y0))*(int)sizeof(double);
    B0 = ((rsub+RPI)*KS + tcol(c0,   y1))*(int)sizeof(double);
    B1 = ((rsub+RPI)*KS + tcol(c0+1, y1))*(int)sizeof(double);
    gofs = (unsigned)(rsub*K + c0);
}

// ALLOPT: the caller vouches that the core, the distortions, the extrinsics,
// the frames and the warp are all being optimized. The variable groups are then
// not tested at run time: with the tests, the compiler zero-initializes every
// group's columns in front of its branch (170 register moves per pass, a
// quarter of the pass's vector instructions). The reference camera's
// extrinsics columns (absent from its rows) are computed regardless in this
// variant: they are never copied out, and the Gram entries they produce sit at
// positions that the assembly has no destination for
// STORE_J = false (round 6; WITH_J && WITH_GRAM only): rows, residuals and Gram as ever, the CSR values not streamed out
template<int PROJ, int NDIST, bool WITH_J, bool WITH_GRAM, bool ALLOPT, bool STORE_J>
__device__ __forceinline__
void board_observation(const DeviceProblem& P,
                       const OpRef& R,
                       const double* __restrict__ joint,
                       double*       __restrict__ gram,
                       const int iobs, double* __restrict__ lds)
{
    constexpr int EXT0   = 4 + NDIST;
    constexpr int FRAME0 = EXT0 + 6;
    constexpr int WARP0  = FRAME0 + 6;
    constexpr int XCOL   = WARP0 + 2;
    constexpr int NCOLS  = XCOL + 1;
    constexpr int NBLK   = (NCOLS + 3)/4;
    constexpr int NCOLS4 = 4*NBLK;              // columns written per row, incl. the zero padding
    constexpr int KS     = NCOLS4 | 1;          // odd LDS row stride
    constexpr int NM     = gram_nmfma_blk(NBLK); // accumulators (problem.hpp)
    static_assert(NBLK <= 8, "the Gram scheme covers 32 tile columns");

    const int lane = threadIdx.x;
#ifdef BOARD_TS
    long long ts[8] = {0,0,0,0,0,0,0,0};
    const long long wall0 = (long long)wall_clock64();      // (100 MHz, the same clock on every XCD; clock64() is per XCD)
    TS(0);
#endif
    const int NPTS = P.W*P.H;

    // FIRST: request everything whose address depends on the observation index
    // alone - the observed pixels (qx,qy,weight of every corner) and the joint
    // pose record. What comes after (the skip flag, the operating point's
    // pointers, the observation's metadata, the camera's intrinsics) is a chain
    // of small dependent scalar loads: it runs under the latency of these. No
    // lane predication (the tail lanes re-read the last element and write into
    // the padding of obs_lds, which is allocated in whole 64-element chunks)
    const double* __restrict__ pool      = P.board_pool + (size_t)iobs*NPTS*3;
    const double* __restrict__ jp_global = joint + (size_t)iobs*JOINT_STRIDE;
    const int n3 = 3*NPTS, last = n3 - 1;
    const double j0 = jp_global[lane];
    const double j1 = jp_global[(lane < JOINT_REC - 64) ? 64 + lane : JOINT_REC - 1];
    // the first 512 values (boards of up to 170 corners: all of them) in
    // straight-line code: 8 loads in flight
    double v_obs[8];
#pragma unroll
    for(int j=0;j<8;j++)
        if(64*j < n3)                       // wave-uniform
        {
            const int idx = 64*j + lane;
            v_obs[j] = pool[idx < last ? idx : last];
        }

    if(opref_skip(R)) return;
    // the output pointers come out of a table in memory: say that they are
    // global, or every store becomes a FLAT store (which also counts against the
    // LDS counter and stalls the LDS waits)
    gdouble* __restrict__ x  = (gdouble*)opref_get(R).x;
    gdouble* __restrict__ Jv = (gdouble*)opref_get(R).Jv;

    const BoardObsMeta m = P.board_meta[iobs];
    const int  k       = m.nnz_per_row;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;

    double* __restrict__ tile    = lds;
    double* __restrict__ obs_lds = lds + 64*KS;
    // the joint pose record of this observation, staged in LDS: it is too big
    // for the scalar registers (84 doubles), and re-reading it from memory in
    // every pass puts a memory latency (long, under this kernel's own write
    // stream) in front of the chain rule. Reads from here are broadcasts
    double* __restrict__ jp      = obs_lds + ((3*NPTS + 63) & ~63);

    // intrinsics of this camera and the board warp, unpacked by the prologue kernel
    double intr[4 + NDIST];
    double warp0, warp1;
    {
        const double* __restrict__ ip = P.unpacked + (size_t)m.icam_intrinsics*P.Nintrinsics;
#pragma unroll
        for(int i=0;i<4+NDIST;i++) intr[i] = ip[i];
        const double* __restrict__ wp = P.unpacked + (size_t)P.Ncameras_intrinsics*P.Nintrinsics;
        warp0 = wp[0]; warp1 = wp[1];
    }

    // wave-uniform reciprocals, once per observation: an FP64 division is ~12
    // instructions, and the pass loop had four of these per corner
    const double inv_Wm1 = P.inv_Wm1,   inv_Hm1 = P.inv_Hm1;
    const double inv_fx  = 1.0/intr[0], inv_fy  = 1.0/intr[1];

    // Gram operands (problem.hpp): at k-step s this lane reads
    // tile[4 s + lane/16][gram_read_col(iread, lane)] for each of the NREAD operands
    constexpr int NREAD = gram_nreads(NBLK);
    int goffs[4];
#pragma unroll
    for(int q=0;q<4;q++) goffs[q] = (lane >> 4)*KS + gram_read_col(NBLK, q < NREAD ? q : 0, lane);
    double acc[NM];
#pragma unroll
    for(int mm=0;mm<NM;mm++) acc[mm] = 0.0;

    // copy-out: this lane's fixed pair of CSR columns (k even), as tile
    // columns for an x row and for a y row
    const int  pairs_per_row = k >> 1;
    const int  rows_per_iter = (k > 0 && !(k & 1)) ? 64 / pairs_per_row : 0;
    const int  co_nactive    = rows_per_iter*pairs_per_row;
    const bool co_active     = lane < co_nactive;
    // the lanes past the last whole row repeat the work of the first ones
    // (harmless: same address, same data) instead of being masked off
    const int  co_lane       = co_active ? lane : lane - co_nactive;
    const int  co_rsub       = (rows_per_iter > 0) ? co_lane / pairs_per_row : 0;
    const int  co_c0         = 2*(co_lane - co_rsub*pairs_per_row);
    // (recomputed where needed rather than kept in registers for the whole kernel)
    auto co_tile_cols = [&](int* tx0, int* tx1, int* ty0, int* ty1)
    {
        *tx0 = board_csr_to_tile_col(P, has_ext, co_c0,   0);
        *tx1 = board_csr_to_tile_col(P, has_ext, co_c0+1, 0);
        *ty0 = board_csr_to_tile_col(P, has_ext, co_c0,   1);
        *ty1 = board_csr_to_tile_col(P, has_ext, co_c0+1, 1);
    };
    // the fast path's invariants (copy_out_full): LDS byte offsets for the rows
    // co_rsub and co_rsub + rows_per_iter, element offset in the output
    constexpr int KFULL = 2 + NDIST + 14;       // core, distortions, extrinsics, frame, warp: everything optimized
    static_assert(!ALLOPT || (WITH_J && WITH_GRAM && !(KFULL & 1)), "ALLOPT: the fused Gram + copy-out path only");
    const bool co_fast  = ALLOPT || (WITH_J && !(KFULL & 1) && (k == KFULL || k == KFULL - 6));
    int co_A0 = 0, co_A1 = 0, co_B0 = 0, co_B1 = 0;
    unsigned co_gofs = 0;
    int co_rsub_c = 0;
    if(ALLOPT)
    {
        if(has_ext) copy_out_invariants<KFULL,   NDIST, KS, true >(lane, co_A0, co_A1, co_B0, co_B1, co_gofs, co_rsub_c);
        else        copy_out_invariants<KFULL-6, NDIST, KS, false>(lane, co_A0, co_A1, co_B0, co_B1, co_gofs, co_rsub_c);
    }
    else if(co_fast)
    {
        int co_tx0, co_tx1, co_ty0, co_ty1;
        co_tile_cols(&co_tx0, &co_tx1, &co_ty0, &co_ty1);
        const bool y0 = (co_rsub & 1) != 0, y1 = ((co_rsub + rows_per_iter) & 1) != 0;
        co_A0 = (co_rsub*KS                 + (y0 ? co_ty0 : co_tx0))*(int)sizeof(double);
        co_A1 = (co_rsub*KS                 + (y0 ? co_ty1 : co_tx1))*(int)sizeof(double);
        co_B0 = ((co_rsub+rows_per_iter)*KS + (y1 ? co_ty0 : co_tx0))*(int)sizeof(double);
        co_B1 = ((co_rsub+rows_per_iter)*KS + (y1 ? co_ty1 : co_tx1))*(int)sizeof(double);
        co_gofs = (unsigned)(co_rsub*k + co_c0);
    }

    // ... and only now are the staged loads needed
#pragma unroll
    for(int j=0;j<8;j++)
        if(64*j < n3)
            obs_lds[64*j + lane] = v_obs[j];
    jp[lane] = j0;
    if(lane < JOINT_REC - 64) jp[64 + lane] = j1;
    for(int base = 512; base < n3; base += 64)
    {
        const int idx = base + lane;
        obs_lds[idx] = pool[idx < last ? idx : last];
    }
    __builtin_amdgcn_wave_barrier();   // obs_lds is complete (one wave: the LDS is in order)
    if(ABLATE(P, 16)) { if(obs_lds[lane] + intr[0] + warp0 + jp[0] == 12345.678) x[0] = 1.0; return; }

#ifdef BOARD_TS
    TS(1);
    long long tcur = ts[1];
#endif
    for(int pt0 = 0; pt0 < NPTS; pt0 += 64)
    {
        const int  pt    = pt0 + lane;
        const bool valid = pt < NPTS;

        // both rows of this corner, at the fixed tile columns
        double row[2][NCOLS4];
#pragma unroll
        for(int xy=0;xy<2;xy++)
#pragma unroll
            for(int c=0;c<NCOLS4;c++) row[xy][c] = 0.0;

        // The lanes past the last corner repeat the last corner's arithmetic
        // (and write nothing) rather than branching around it: ONE conditional
        // region below - the rows of a live inlier - means one zero
        // initialization of the row registers, not one per nesting level
        {
            const int ptc = valid ? pt : NPTS - 1;
            const int iy = ptc / P.W;
            const int ix = ptc - iy*P.W;
            const double bx = (double)ix * P.spacing;
            const double by = (double)iy * P.spacing;
            double bz = 0.0, dz_dw0 = 0.0, dz_dw1 = 0.0;
            if(ALLOPT || P.has_warp_seed)
            {
                // parabolic flex along each board axis, max deflection at the centre
                const double xr = (double)ix * inv_Wm1;
                const double yr = (double)iy * inv_Hm1;
                dz_dw0 = 4.0*xr*(1.0 - xr);
                dz_dw1 = 4.0*yr*(1.0 - yr);
                bz += warp0*dz_dw0;
                bz += warp1*dz_dw1;
            }

            double p[3];
#pragma unroll
            for(int i=0;i<3;i++)
                p[i] = jp[JOINT_R+3*i+0]*bx + jp[JOINT_R+3*i+1]*by + jp[JOINT_R+3*i+2]*bz + jp[JOINT_T+i];

            double q[2], dq_dp[2][3], dq_dk[2][NDIST > 0 ? NDIST : 1];
            if(ABLATE(P, 4))
            {
                q[0] = p[0]; q[1] = p[1];
#pragma unroll
                for(int i=0;i<3;i++) { dq_dp[0][i] = p[i]; dq_dp[1][i] = p[i] + 1.0; }
#pragma unroll
                for(int i=0;i<NDIST;i++) { dq_dk[0][i] = p[0] + (double)i; dq_dk[1][i] = p[1] + (double)i; }
            }
            else
            project_lens<PROJ,NDIST,WITH_J>(q, dq_dp, dq_dk, p, intr, P.cfg);

            const double qx_obs = obs_lds[3*ptc + 0];
            const double qy_obs = obs_lds[3*ptc + 1];
            const double w      = obs_lds[3*ptc + 2];
            const bool   inlier = valid && (w >= 0.0);

            double2 err;
            err.x = inlier ? (q[0] - qx_obs)*w : 0.0;
            err.y = inlier ? (q[1] - qy_obs)*w : 0.0;
            if(valid) { d2_t e2; e2.x = err.x; e2.y = err.y; __builtin_nontemporal_store(e2, reinterpret_cast<gdouble2*>(&x[m.i_meas0 + 2*pt])); }

            // outliers keep their columns and get all-zero values: everything
            // below is skipped for them and the rows stay 0
            if(WITH_J && inlier)
            {
                if(ALLOPT || P.Ncore_state)
                {
                    row[0][0] = (q[0] - intr[2])*inv_fx * w * SCALE_INTRINSICS_FOCAL_LENGTH;
                    row[0][2] = w * SCALE_INTRINSICS_CENTER_PIXEL;
                    row[1][1] = (q[1] - intr[3])*inv_fy * w * SCALE_INTRINSICS_FOCAL_LENGTH;
                    row[1][3] = w * SCALE_INTRINSICS_CENTER_PIXEL;
                }
                if(NDIST > 0 && (ALLOPT || P.Ndist_state))
                {
#pragma unroll
                    for(int xy=0;xy<2;xy++)
#pragma unroll
                        for(int i=0;i<NDIST;i++)
                            row[xy][4+i] = dq_dk[xy][i] * w * SCALE_DISTORTION;
                }
                if(ALLOPT || has_ext)
                {
                    // dp/drc = X Mc0 + Y Mc1 + Z Mc2 + dtj/drc ; dp/dtc = I
#pragma unroll
                    for(int l=0;l<3;l++)
                    {
                        double dp[3];
#pragma unroll
                        for(int i=0;i<3;i++)
                            dp[i] =
                                bx*jp[JOINT_MC + 0  + 3*i + l] +
                                by*jp[JOINT_MC + 9  + 3*i + l] +
                                bz*jp[JOINT_MC + 18 + 3*i + l] +
                                jp[JOINT_DTJ_DRC + 3*i + l];
#pragma unroll
                        for(int xy=0;xy<2;xy++)
                        {
                            const double g = dq_dp[xy][0]*dp[0] + dq_dp[xy][1]*dp[1] + dq_dp[xy][2]*dp[2];
                            row[xy][EXT0+l]   = g * w * SCALE_ROTATION_CAMERA;
                            row[xy][EXT0+3+l] = dq_dp[xy][l] * w * SCALE_TRANSLATION_CAMERA;
                        }
                    }
                }
                if(ALLOPT || P.do_optimize_frames)
                {
#pragma unroll
                    for(int l=0;l<3;l++)
                    {
                        double dpr[3], dpt[3];
#pragma unroll
                        for(int i=0;i<3;i++)
                        {
                            dpr[i] =
                                bx*jp[JOINT_MF + 0  + 3*i + l] +
                                by*jp[JOINT_MF + 9  + 3*i + l] +
                                bz*jp[JOINT_MF + 18 + 3*i + l];
                            dpt[i] = jp[JOINT_DTJ_DTF + 3*i + l];
                        }
#pragma unroll
                        for(int xy=0;xy<2;xy++)
                        {
                            const double gr = dq_dp[xy][0]*dpr[0] + dq_dp[xy][1]*dpr[1] + dq_dp[xy][2]*dpr[2];
                            const double gt = dq_dp[xy][0]*dpt[0] + dq_dp[xy][1]*dpt[1] + dq_dp[xy][2]*dpt[2];
                            row[xy][FRAME0+l]   = gr * w * SCALE_ROTATION_FRAME;
                            row[xy][FRAME0+3+l] = gt * w * SCALE_TRANSLATION_FRAME;
                        }
                    }
                }
                if(ALLOPT || P.has_warp_state)
                {
                    // dq/dwarp_i = (dq/dt . Rj[:,2]) dz/dwarp_i
#pragma unroll
                    for(int xy=0;xy<2;xy++)
                    {
                        const double d =
                            dq_dp[xy][0]*jp[JOINT_R + 2] +
                            dq_dp[xy][1]*jp[JOINT_R + 5] +
                            dq_dp[xy][2]*jp[JOINT_R + 8];
                        row[xy][WARP0+0] = (w*SCALE_CALOBJECT_WARP)*(d*dz_dw0);
                        row[xy][WARP0+1] = (w*SCALE_CALOBJECT_WARP)*(d*dz_dw1);
                    }
                }
            }
            if(WITH_GRAM)
            {
                row[0][XCOL] = err.x;
                row[1][XCOL] = err.y;
            }
        }

        if(!WITH_J) continue;
        TSACC(2, tcur);     // projection + rows

        // The two halves of the pass (its first and its second 32 corners) go through the 64-row tile one after
        // the other. All 64 lanes write each of them, one row each: lanes 0..31 the x rows of the half's corners,
        // lanes 32..63 their y rows. (32 lanes writing both rows of their own corners took twice the ds_write
        // instructions, each moving half a wave's worth: a ds_write_b64 costs the VGPR->LDS path its 6 clocks
        // whatever the number of lanes under exec, and the tile writes are on this kernel's critical path where
        // the arithmetic is not - profiles/r03_board_kernel_ablation.txt.) For that the y rows of the first 32
        // corners (lanes 0..31) and the x rows of the last 32 (lanes 32..63) trade places: v_permlane32_swap
        // (gfx950) exchanges the upper half of one register with the lower half of another, two per double.
        // After it slot 0 of a lane holds its row of the first half, slot 1 its row of the second.
        // (A 16-lane group now writes every second tile row, two lanes to a bank: SQ_LDS_BANK_CONFLICT went up by 220
        //  per observation while the instructions halved. The conflict-free variant - lanes 0..7 of each 16-lane row
        //  on the first half's corners, lanes 8..15 on the second's, a DPP rotation by 8 and two selects per value,
        //  16 consecutive rows per group - was built: same bits, 74.8 us against 73.5. It is the instruction count
        //  on the VGPR->LDS path that matters, not the conflicts)
#pragma unroll
        for(int c=0;c<NCOLS4;c++) permlane32_swap_f64(row[0][c], row[1][c]);
        for(int h = 0; h < 2; h++)
        {
            const int nc    = NPTS - pt0 - 32*h;          // corners in this half
            if(nc <= 0) break;
            const int nrows = 2*((nc < 32) ? nc : 32);

            // WAR: the previous half's tile reads are complete (in-order LDS)
            __builtin_amdgcn_wave_barrier();
            {
                double* __restrict__ t0 = tile + (size_t)(2*(lane & 31) + (lane >> 5))*KS;
                if(h == 0)
                {
#pragma unroll
                    for(int c=0;c<NCOLS4;c++) t0[c] = row[0][c];
                }
                else
                {
#pragma unroll
                    for(int c=0;c<NCOLS4;c++) t0[c] = row[1][c];
                }
            }
            __builtin_amdgcn_wave_barrier();
            TSACC(3, tcur);     // tile write

            // stream the half-tile out: rows row0 .. row0+nrows of the observation
            gdouble* __restrict__ out = Jv + m.i_nnz0 + (size_t)(2*(pt0 + 32*h))*k;
            if(ALLOPT || (WITH_GRAM && co_fast && !ABLATE(P, 3)))
            {
                // all the usual variables optimized: copy-out in the shadow of the Gram's MFMAs
                const int  rs    = ALLOPT ? co_rsub_c : co_rsub;
                const bool kfull = ALLOPT ? has_ext : (k == KFULL);
                if(nrows == 64)
                {
                    if(kfull) gram_copy_fused<true,STORE_J,NBLK,KFULL,  KS>(acc, tile, goffs, out, co_A0, co_A1, co_B0, co_B1, co_gofs, rs, 64);
                    else      gram_copy_fused<true,STORE_J,NBLK,KFULL-6,KS>(acc, tile, goffs, out, co_A0, co_A1, co_B0, co_B1, co_gofs, rs, 64);
                }
                else
                {
                    if(kfull) gram_copy_fused<false,STORE_J,NBLK,KFULL,  KS>(acc, tile, goffs, out, co_A0, co_A1, co_B0, co_B1, co_gofs, rs, nrows);
                    else      gram_copy_fused<false,STORE_J,NBLK,KFULL-6,KS>(acc, tile, goffs, out, co_A0, co_A1, co_B0, co_B1, co_gofs, rs, nrows);
                }
                TSACC(5, tcur);
                continue;
            }
            if(STORE_J && !ABLATE(P, 1))
            {
                if(co_fast && nrows == 64)
                {
                    if(k == KFULL) copy_out_full<KFULL,   KS>(tile, out, co_A0, co_A1, co_B0, co_B1, co_gofs, co_rsub);
                    else           copy_out_full<KFULL-6, KS>(tile, out, co_A0, co_A1, co_B0, co_B1, co_gofs, co_rsub);
                }
                else if(rows_per_iter > 0)
                {
                    int co_tx0, co_tx1, co_ty0, co_ty1;
                    co_tile_cols(&co_tx0, &co_tx1, &co_ty0, &co_ty1);
                    if(co_active)
                        for(int r = co_rsub; r < nrows; r += rows_per_iter)
                        {
                            const bool isy = (r & 1) != 0;
                            d2_t v;
                            v.x = tile[r*KS + (isy ? co_ty0 : co_tx0)];
                            v.y = tile[r*KS + (isy ? co_ty1 : co_tx1)];
                            __builtin_nontemporal_store(v, reinterpret_cast<gdouble2*>(&out[r*k + co_c0]));
                        }
                }
                else
                {
                    // odd k: a 16-byte pair can straddle two rows
                    const int nelem = nrows*k;
                    for(int e = 2*lane; e < nelem; e += 128)
                    {
                        const int r0 = e / k,       c0 = e - r0*k;
                        const int r1 = (e+1) / k,   c1 = (e+1) - r1*k;
                        d2_t v;
                        v.x = tile[r0*KS + board_csr_to_tile_col(P, has_ext, c0, r0 & 1)];
                        v.y = tile[r1*KS + board_csr_to_tile_col(P, has_ext, c1, r1 & 1)];
                        __builtin_nontemporal_store(v, reinterpret_cast<gdouble2*>(&out[e]));
                    }
                }
            }

            TSACC(4, tcur);     // copy-out
            if(WITH_GRAM && !ABLATE(P, 2))
            {
                // G += Tt T over this half. 4 tile rows per k-step; the rows
                // between nrows and the end of the last step belong to lanes
                // without a corner, which stored zeros
                // (the full halves of the usual problems take the fused path above)
                const int nsteps = (nrows + 3) >> 2;
                for(int s = 0; s < nsteps; s++)
                {
                    GramRd rd;
#pragma unroll
                    for(int q=0;q<4;q++) rd.v[q] = (q < NREAD) ? tile[s*4*KS + goffs[q]] : 0.0;
                    gram_step_ops<NBLK>(acc, rd);
                }
            }
#ifdef BOARD_TS
            { const double dep = acc[0]; asm volatile("" :: "v"(dep)); }
#endif
            TSACC(5, tcur);     // Gram
        }
    }

    if(WITH_J && WITH_GRAM)
    {
        gdouble* __restrict__ g = (gdouble*)gram + (size_t)iobs*(NM*64);
#pragma unroll
        for(int mm=0;mm<NM;mm++)
            g[mm*64 + lane] = acc[mm];
    }
#ifdef BOARD_TS
    TS(6);
    if(P.debug_ts != NULL && lane == 0)
    {
        long long* o = P.debug_ts + (size_t)iobs*10;
        // start, end of startup, [projection, tile write, copy-out, Gram] cycles, end, hw id
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = ts[0]; o[1] = ts[1]; o[2] = ts[2]; o[3] = ts[3]; o[4] = ts[4]; o[5] = ts[5]; o[6] = ts[6];
        o[7] = (long long)hwid | ((long long)(xcc & 0xf) << 32);
        o[8] = wall0; o[9] = (long long)wall_clock64();
    }
#endif
}

template<int PROJ, int NDIST, bool WITH_J, bool WITH_GRAM, bool ALLOPT = false, bool STORE_J = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
void board_kernel(DeviceProblem P,
                  OpRef R,
                  const double* __restrict__ joint,
                  double*       __restrict__ gram)
{
    static_assert(STORE_J || (WITH_J && WITH_GRAM), "without the stream the rows exist for the Gram's sake");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    board_observation<PROJ,NDIST,WITH_J,WITH_GRAM,ALLOPT,STORE_J>(P, R, joint, gram, blockIdx.x, lds);
}

// CSR structure of the board rows: rowptr and colidx. Same tiling as above,
// written once at problem creation
__global__ __launch_bounds__(64)
void board_structure_kernel(DeviceProblem P, int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx)
{
    const int iobs = blockIdx.x;
    const int lane = threadIdx.x;
    const BoardObsMeta m = P.board_meta[iobs];
    const int k     = m.nnz_per_row;
    const int nrows = 2*P.W*P.H;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;

    for(int r = lane; r < nrows; r += 64)
        rowptr[m.i_meas0 + r] = (int32_t)(m.i_nnz0 + (int64_t)r*k);

    const int64_t nelem = (int64_t)nrows*k;
    for(int64_t e = lane; e < nelem; e += 64)
    {
        const int r  = (int)(e / k);
        int       c  = (int)(e - (int64_t)r*k);
        const int xy = r & 1;
        int col;
        if(P.Ncore_state && c < 2)
            col = m.i_state_intrinsics + xy + 2*c;
        else
        {
            if(P.Ncore_state) c -= 2;
            // splined models: the (order+1)^2 patch columns are data dependent
            // and rewritten by every evaluation; this is a placeholder
            if(c < P.Ndist_row)
                col = m.i_state_intrinsics + P.Ncore_state + c;
            else
            {
                c -= P.Ndist_row;
                if(has_ext && c < 6)
                    col = m.i_state_extrinsics + c;
                else
                {
                    if(has_ext) c -= 6;
                    if(P.do_optimize_frames && c < 6)
                        col = m.i_state_frame + c;
                    else
                    {
                        if(P.do_optimize_frames) c -= 6;
                        col = P.i_state_warp + c;
                    }
                }
            }
        }
        colidx[m.i_nnz0 + e] = col;
    }
}

////////////////////////////////////////////////////////////////////////////////
// 3. discrete points: one lane per observation (2 rows)
////////////////////////////////////////////////////////////////////////////////
template<int PROJ, int NDIST, bool WITH_J>
__global__ __launch_bounds__(64)
void point_kernel(DeviceProblem P, OpRef R)
{
    if(opref_skip(R)) return;
    const double* __restrict__ b  = opref_get(R).b;
    double*       __restrict__ x  = opref_get(R).x;
    double*       __restrict__ Jv = opref_get(R).Jv;
    const int iobs = blockIdx.x*blockDim.x + threadIdx.x;
    if(iobs >= P.Nobs_point) return;
    const PointObsMeta m = P.point_meta[iobs];
    const int k = m.nnz_per_row;

    const double* obs = P.point_pool + (size_t)iobs*3;
    const double  w   = obs[2];
    // note: <= here, < for boards. That is what the reference does
    // (mrcal.c:4918 vs :4706)
    const bool inlier = !(w <= 0.0);

    double* row[2] = { NULL, NULL };
    if(WITH_J)
    {
        row[0] = Jv + m.i_nnz0;
        row[1] = Jv + m.i_nnz0 + k;
    }

    if(!inlier)
    {
        x[m.i_meas0+0] = 0.0;
        x[m.i_meas0+1] = 0.0;
        if(WITH_J)
            for(int c=0;c<2*k;c++) row[0][c] = 0.0;
        return;
    }

    double intr[4 + NDIST];
#pragma unroll
    for(int i=0;i<4+NDIST;i++) intr[i] = get_intrinsic(P, b, m.icam_intrinsics, i);

    double pref[3];
    if(m.i_state_point >= 0)
        for(int i=0;i<3;i++) pref[i] = b[m.i_state_point + i] * SCALE_POSITION_POINT;
    else
        for(int i=0;i<3;i++) pref[i] = P.seed_points[3*m.i_point + i];

    // p = R(rc) pref + tc, or pref if the camera is at the reference
    double p[3];
    double dp_drc[3][3], dp_dpt[3][3];
    const bool at_ref = (m.icam_extrinsics < 0);
    if(at_ref)
    {
        for(int i=0;i<3;i++) p[i] = pref[i];
        for(int i=0;i<3;i++) for(int l=0;l<3;l++) { dp_drc[i][l] = 0.0; dp_dpt[i][l] = (i==l) ? 1.0 : 0.0; }
    }
    else
    {
        double rt_cam[6];
        get_rt_cam_ref(rt_cam, P, b, m.icam_extrinsics);
        Dual<6> rc[3], xx[3], y[3];
        for(int i=0;i<3;i++)
        {
            rc[i] = Dual<6>::variable(rt_cam[i], i);
            xx[i] = Dual<6>::variable(pref[i],   3+i);
        }
        rotate_point_r_dual<6>(y, rc, xx, false);
        for(int i=0;i<3;i++)
        {
            p[i] = y[i].x + rt_cam[3+i];
            for(int l=0;l<3;l++) { dp_drc[i][l] = y[i].d[l]; dp_dpt[i][l] = y[i].d[3+l]; }
        }
    }

    double q[2], dq_dp[2][3], dq_dk[2][NDIST > 0 ? NDIST : 1];
    project_lens<PROJ,NDIST,WITH_J>(q, dq_dp, dq_dk, p, intr, P.cfg);

    x[m.i_meas0+0] = (q[0] - obs[0])*w;
    x[m.i_meas0+1] = (q[1] - obs[1])*w;

    if(!WITH_J) return;

    const bool has_ext = P.do_optimize_extrinsics && !at_ref;
    for(int xy=0;xy<2;xy++)
    {
        int c = 0;
        if(P.Ncore_state)
        {
            row[xy][0] = (q[xy] - intr[2+xy])/intr[xy] * w * SCALE_INTRINSICS_FOCAL_LENGTH;
            row[xy][1] = w * SCALE_INTRINSICS_CENTER_PIXEL;
            c = 2;
        }
        if(NDIST > 0 && P.Ndist_state)
        {
            for(int i=0;i<NDIST;i++) row[xy][c+i] = dq_dk[xy][i] * w * SCALE_DISTORTION;
            c += NDIST;
        }
        if(has_ext)
        {
            for(int l=0;l<3;l++)
            {
                const double g = dq_dp[xy][0]*dp_drc[0][l] + dq_dp[xy][1]*dp_drc[1][l] + dq_dp[xy][2]*dp_drc[2][l];
                row[xy][c+l]   = g * w * SCALE_ROTATION_CAMERA;
                row[xy][c+3+l] = dq_dp[xy][l] * w * SCALE_TRANSLATION_CAMERA;
            }
            c += 6;
        }
        if(m.i_state_point >= 0)
        {
            for(int l=0;l<3;l++)
            {
                const double g = dq_dp[xy][0]*dp_dpt[0][l] + dq_dp[xy][1]*dp_dpt[1][l] + dq_dp[xy][2]*dp_dpt[2][l];
                row[xy][c+l] = g * w * SCALE_POSITION_POINT;
            }
            c += 3;
        }
    }
}

__global__ __launch_bounds__(64)
void point_structure_kernel(DeviceProblem P, int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx)
{
    const int iobs = blockIdx.x*blockDim.x + threadIdx.x;
    if(iobs >= P.Nobs_point) return;
    const PointObsMeta m = P.point_meta[iobs];
    const int k = m.nnz_per_row;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;
    for(int xy=0;xy<2;xy++)
    {
        rowptr[m.i_meas0 + xy] = (int32_t)(m.i_nnz0 + xy*k);
        int32_t* ci = colidx + m.i_nnz0 + xy*k;
        int c = 0;
        if(P.Ncore_state)
        {
            ci[c++] = m.i_state_intrinsics + xy;
            ci[c++] = m.i_state_intrinsics + xy + 2;
        }
        // splined models reach here only as outliers-or-not with the SAME
        // count: the first (order+1)^2 distortion columns for outliers,
        // the real patch otherwise; the patch is data dependent and is
        // rewritten by the spline kernels
        const int ndist_cols = k - c - (has_ext ? 6 : 0) - (m.i_state_point >= 0 ? 3 : 0);
        for(int i=0;i<ndist_cols;i++) ci[c++] = m.i_state_intrinsics + P.Ncore_state + i;
        if(has_ext)
            for(int i=0;i<6;i++) ci[c++] = m.i_state_extrinsics + i;
        if(m.i_state_point >= 0)
            for(int i=0;i<3;i++) ci[c++] = m.i_state_point + i;
    }
}

////////////////////////////////////////////////////////////////////////////////
// 4. regularization (parametric models): one lane per row
////////////////////////////////////////////////////////////////////////////////
// Rows, in order: [distortions of cam0..camN] [centre pixel x,y of cam0..camN]
// [unity_cam01]. Reference: mrcal.c:5795-5954
template<bool WITH_J, bool WITH_STRUCTURE, class BV>
__device__ __forceinline__
void regularization_row_at(const DeviceProblem& P, const BV& b, double* __restrict__ x, double* __restrict__ Jv,
                           int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx, const int i)
{
    const int Ndist_rows   = P.do_apply_regularization ? P.Ncameras_intrinsics*P.Ndist_state : 0;
    const int Ncenter_rows = (P.do_apply_regularization && P.Ncore_state) ? P.Ncameras_intrinsics*2 : 0;
    const int Nrows        = Ndist_rows + Ncenter_rows + (P.has_unity_cam01 ? 1 : 0);

    if(i == 0 && WITH_STRUCTURE)
        rowptr[P.i_meas_regularization + Nrows] =
            (int32_t)(P.i_nnz_regularization + Ndist_rows + Ncenter_rows + (P.has_unity_cam01 ? 3 : 0));
    if(i >= Nrows) return;

    const double nominal_pixel_error = 0.1;
    const int     imeas = P.i_meas_regularization + i;
    const int64_t innz  = P.i_nnz_regularization  + i;
    if(WITH_STRUCTURE) rowptr[imeas] = (int32_t)innz;

    if(i < Ndist_rows)
    {
        const int icam = i / P.Ndist_state;
        const int j    = i - icam*P.Ndist_state;
        double scale = nominal_pixel_error / 1.0;
        // the denominator coefficients of the rational OpenCV models are
        // pulled towards 0 harder
        if(P.lens_type >= MRCAL_LENSMODEL_OPENCV8 && P.lens_type <= MRCAL_LENSMODEL_OPENCV12 &&
           5 <= j && j <= 7)
            scale *= 5.0;
        x[imeas] = scale * get_intrinsic(P, b, icam, P.Ncore + j);
        if(WITH_J)         Jv[innz]     = scale * SCALE_DISTORTION;
        if(WITH_STRUCTURE) colidx[innz] = P.i_state_intrinsics + icam*P.Nintr_state + P.Ncore_state + j;
        return;
    }
    if(i < Ndist_rows + Ncenter_rows)
    {
        const int ii   = i - Ndist_rows;
        const int icam = ii >> 1;
        const int xy   = ii & 1;
        // camera 0's width sets the scale for every camera
        const double scale  = nominal_pixel_error / (P.imager_width_cam0 * 0.1);
        const double target = 0.5 * (double)(P.imagersizes[2*icam + xy] - 1);
        x[imeas] = scale * (get_intrinsic(P, b, icam, 2+xy) - target);
        if(WITH_J)         Jv[innz]     = scale * SCALE_INTRINSICS_CENTER_PIXEL;
        if(WITH_STRUCTURE) colidx[innz] = P.i_state_intrinsics + icam*P.Nintr_state + 2 + xy;
        return;
    }
    // unity_cam01: pull |t_cam0| to 1
    {
        const double scale = nominal_pixel_error / (1.0 * 0.01);
        double rt[6];
        get_rt_cam_ref(rt, P, b, 0);
        x[imeas] = scale * (rt[3]*rt[3] + rt[4]*rt[4] + rt[5]*rt[5] - 1.0);
        for(int l=0;l<3;l++)
        {
            if(WITH_J)         Jv[innz+l]     = scale * SCALE_TRANSLATION_CAMERA * 2.0 * rt[3+l];
            if(WITH_STRUCTURE) colidx[innz+l] = P.i_state_extrinsics + 3 + l;
        }
    }
}

template<bool WITH_J, bool WITH_STRUCTURE>
__global__ __launch_bounds__(64)
void regularization_kernel(DeviceProblem P, OpRef R, int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx)
{
    if(opref_skip(R)) return;
    regularization_row<WITH_J,WITH_STRUCTURE>(P, R, rowptr, colidx, blockIdx.x*blockDim.x + threadIdx.x);
}

// the prologue launch; with B.choose the trial step's choice rides in it (board_prologue_kernel<true>)
void launch_prologue(const DeviceProblem& P, const EvalBuffers& B, int nblocks_obs, int nblocks_unpack, int nblocks_zero,
                            int nblocks_reg, bool with_jacobian, hipStream_t stream)
{
    const int reg_mode = nblocks_reg > 0 ? (with_jacobian ? 1 : 0) : -1;
    const int n = nblocks_obs + nblocks_unpack + nblocks_zero + nblocks_reg;
    if(B.choose != NULL)
    {
        const int nblocks_choose = (B.choose->nd.Nstate + PRO_T - 1)/PRO_T;
        hipLaunchKernelGGL(board_prologue_kernel<true>, dim3(n + nblocks_choose), dim3(PRO_T), 0, stream,
                           P, B, nblocks_unpack, nblocks_zero, reg_mode, nblocks_reg, *B.choose);
    }
    else
        hipLaunchKernelGGL(board_prologue_kernel<false>, dim3(n), dim3(PRO_T), 0, stream,
                           P, B, nblocks_unpack, nblocks_zero, reg_mode, nblocks_reg, ChooseArgs());
}
////////////////////////////////////////////////////////////////////////////////
// 7. triangulated points: one lane per pair of observations (1 row)
////////////////////////////////////////////////////////////////////////////////
// Row = [6 columns of camera 0's rt, if it is not the reference]
//       [6 columns of camera 1's rt, ...], in that order (mrcal.c:5383-5506).
// A pair with an outlier observation keeps its columns, x = 0, values 0
// (round 6) WITH_J: TWO lanes a pair, in different workgroups - the even ones take the partials with respect to camera
// 0's pose, the odd ones camera 1's (triangulation.hpp: duals that carry the partials they can have). 67 000 pairs are
// 1045 waves at a lane each: a wave to a SIMD, nothing to hide its own latencies behind, 25.8 us at BASELINE configuration
// 5, its longest launch; the halves are 2090 waves of half the partials each
// (the body, for the launches that carry it: triangulated_kernel, and board_tri_kernel beside the board observations)
template<bool WITH_J>
__device__ __forceinline__ void triangulated_pairs(const DeviceProblem& P, const OpRef& R, const int block)
{
    const int half = WITH_J ? (block & 1) : 0;
    const int ip = (WITH_J ? (block >> 1) : block)*64 + (int)threadIdx.x;
    if(ip >= P.Npairs_tri) return;
    const TriPairMeta m = P.tri_meta[ip];
    const double* __restrict__ b  = opref_get(R).b;
    double*       __restrict__ x  = opref_get(R).x;
    double*       __restrict__ Jv = opref_get(R).Jv;
    const int n0 = (m.i_state_extrinsics0 >= 0) ? 6 : 0;
    const int n1 = (m.i_state_extrinsics1 >= 0) ? 6 : 0;
    if(P.tri_outlier[m.i0] || P.tri_outlier[m.i1])
    {
        if(half == 0) x[m.i_meas] = 0.0;
        if(WITH_J)
        {
            if(half == 0) for(int c=0;c<n0;c++)     Jv[m.i_nnz0 + c] = 0.0;
            else          for(int c=n0;c<n0+n1;c++) Jv[m.i_nnz0 + c] = 0.0;
        }
        return;
    }
    double rt0[6], rt1[6];
    if(m.icam_extrinsics0 >= 0) get_rt_cam_ref(rt0, P, b, m.icam_extrinsics0);
    if(m.icam_extrinsics1 >= 0) get_rt_cam_ref(rt1, P, b, m.icam_extrinsics1);
    const double* v0 = P.tri_px + 3*(size_t)m.i0;
    const double* v1 = P.tri_px + 3*(size_t)m.i1;
    const double v0l[3] = { v0[0], v0[1], v0[2] }, v1l[3] = { v1[0], v1[1], v1[2] };
    const double* p0 = (m.icam_extrinsics0 >= 0) ? rt0 : NULL;
    const double* p1 = (m.icam_extrinsics1 >= 0) ? rt1 : NULL;
    if(!WITH_J)
    {
        x[m.i_meas] = tri_pair_error<0>(v0l, v1l, p0, p1, NULL).x;
        return;
    }
    double de[12];
    if(half == 0)
    {
        // (a camera at the reference has no columns and its half carries six zero partials through the error function: one
        //  piece of code for every lane of the wave, whichever kind of pair it holds)
        x[m.i_meas] = tri_pair_error_partials<true, false>(de, v0l, v1l, p0, p1, NULL);
        if(n0)
        {
            for(int i=0;i<3;i++) Jv[m.i_nnz0 + i]     = de[i]   * SCALE_ROTATION_CAMERA;
            for(int i=0;i<3;i++) Jv[m.i_nnz0 + 3 + i] = de[3+i] * SCALE_TRANSLATION_CAMERA;
        }
    }
    else
    {
        (void)tri_pair_error_partials<false, true>(de, v0l, v1l, p0, p1, NULL);
        if(n1 == 0) return;
        for(int i=0;i<3;i++) Jv[m.i_nnz0 + n0 + i]     = de[6+i] * SCALE_ROTATION_CAMERA;
        for(int i=0;i<3;i++) Jv[m.i_nnz0 + n0 + 3 + i] = de[9+i] * SCALE_TRANSLATION_CAMERA;
    }
}
template<bool WITH_J, bool WITH_STRUCTURE>
__global__ __launch_bounds__(64)
void triangulated_kernel(DeviceProblem P, OpRef R, int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx)
{
    if(opref_skip(R)) return;
    if(WITH_STRUCTURE)
    {
        const int ip = (int)blockIdx.x*blockDim.x + threadIdx.x;
        if(ip >= P.Npairs_tri) return;
        const TriPairMeta m = P.tri_meta[ip];
        rowptr[m.i_meas] = (int32_t)m.i_nnz0;
        int c = 0;
        if(m.i_state_extrinsics0 >= 0) { for(int i=0;i<6;i++) colidx[m.i_nnz0 + c + i] = m.i_state_extrinsics0 + i; c += 6; }
        if(m.i_state_extrinsics1 >= 0) { for(int i=0;i<6;i++) colidx[m.i_nnz0 + c + i] = m.i_state_extrinsics1 + i; }
        return;
    }
    else
        triangulated_pairs<WITH_J>(P, R, (int)blockIdx.x);
}
// (round 6) The board observations and the triangulated pairs of a structure-from-motion problem in ONE launch: 1600
// observations are 1600 waves, 67 000 pairs 2090, for 2048 places either way - the two launches one after the other
// each left the chip half empty for 22 + 24 us (BASELINE configuration 5). Neither reads what the other writes. The
// pairs' workgroups - the longer waves - come first. For the board kernel's general variant (the intrinsics of such a
// problem are locked, as a rule); a problem that optimizes everything keeps the two launches
template<int PROJ, int NDIST, bool STORE_J>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
void board_tri_kernel(DeviceProblem P, OpRef R, const double* __restrict__ joint, double* __restrict__ gram, int ntri_blocks)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if((int)blockIdx.x < ntri_blocks)
    {
        if(!opref_skip(R)) triangulated_pairs<true>(P, R, (int)blockIdx.x);
        return;
    }
    board_observation<PROJ,NDIST,true,true,false,STORE_J>(P, R, joint, gram, (int)blockIdx.x - ntri_blocks, lds);
}
static int triangulated_blocks_with_jacobian(const DeviceProblem& P) { return 2*((P.Npairs_tri + 63)/64); }
void launch_triangulated(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian, hipStream_t stream)
{
    if(P.Npairs_tri <= 0) return;
    const dim3 grid((P.Npairs_tri + 63)/64), block(64);
    if(with_jacobian)
        hipLaunchKernelGGL((triangulated_kernel<true,false>),  dim3(triangulated_blocks_with_jacobian(P)), block, 0, stream, P, B.R, (int32_t*)NULL, (int32_t*)NULL);
    else
        hipLaunchKernelGGL((triangulated_kernel<false,false>), grid, block, 0, stream, P, B.R, (int32_t*)NULL, (int32_t*)NULL);
}

////////////////////////////////////////////////////////////////////////////////
// launchers
////////////////////////////////////////////////////////////////////////////////
// the board kernel's variant for problems that optimize everything (the ablation probes keep the general kernel)
static bool board_allopt(const DeviceProblem& P, int ndist)
{
    return ((16 + ndist) & 1) == 0 && P.Ncore_state && (ndist == 0 || P.Ndist_state) && P.do_optimize_extrinsics &&
           P.do_optimize_frames && P.has_warp_state && P.has_warp_seed && !ABLATE(P, ~0);
}
static int lens_ndist(int lens_type)
{
    switch(lens_type)
    {
    case MRCAL_LENSMODEL_OPENCV4:  return 4;
    case MRCAL_LENSMODEL_OPENCV5:  return 5;
    case MRCAL_LENSMODEL_OPENCV8:  return 8;
    case MRCAL_LENSMODEL_OPENCV12: return 12;
    case MRCAL_LENSMODEL_CAHVOR:   return 5;
    case MRCAL_LENSMODEL_CAHVORE:  return 8;
    default:                       return 0;
    }
}
// the triangulated pairs ride in the board kernel's launch (board_tri_kernel) when both are there and the Jacobian and
// the Grams are asked for
bool board_launch_takes_triangulated(const DeviceProblem& P)
{
    return P.Nobs_board > 0 && P.Npairs_tri > 0 && P.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC &&
           !board_allopt(P, lens_ndist(P.lens_type));
}
template<int PROJ, int NDIST>
static void launch_eval_t(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                          int lds_bytes, hipStream_t stream,
                          hipEvent_t ev_j0, hipEvent_t ev_j1, int parts)
{
    constexpr bool kfull_even = ((16 + NDIST) & 1) == 0;
    const bool allopt = board_allopt(P, NDIST);
    const bool tri_rides = with_jacobian && B.gram != NULL && board_launch_takes_triangulated(P);
    if(P.Nobs_board > 0 && (parts & EVAL_PART_PROLOGUE))
    {
        const int nblocks_obs    = prologue_obs_blocks(P.Nobs_board);
        const int nblocks_unpack = (P.Ncameras_intrinsics*P.Nintrinsics + 2 + PRO_T - 1)/PRO_T;
        const int nblocks_zero   = (B.zero_total > 0) ? PROLOGUE_ZERO_BLOCKS(B.zero_total) : 0;
        const int Nreg_rows      = P.Nmeas - P.i_meas_regularization;
        const int nblocks_reg    = (Nreg_rows + PRO_T - 1)/PRO_T;
        launch_prologue(P, B, nblocks_obs, nblocks_unpack, nblocks_zero, nblocks_reg, with_jacobian, stream);
    }
    if(P.Nobs_board > 0 && (parts & EVAL_PART_BOARD))
    {
        if(ev_j0) hipEventRecord(ev_j0, stream);
        if(with_jacobian && B.gram != NULL && allopt)
        {
            if constexpr (kfull_even)
            {
                if(B.store_jacobian)
                    hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,true,true>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                                       P, B.R, B.joint, B.gram);
                else
                    hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,true,true,false>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                                       P, B.R, B.joint, B.gram);
            }
        }
        else if(tri_rides)
        {
            const int ntri = triangulated_blocks_with_jacobian(P);
            if(B.store_jacobian)
                hipLaunchKernelGGL((board_tri_kernel<PROJ,NDIST,true>),  dim3(ntri + P.Nobs_board), dim3(64), lds_bytes, stream, P, B.R, B.joint, B.gram, ntri);
            else
                hipLaunchKernelGGL((board_tri_kernel<PROJ,NDIST,false>), dim3(ntri + P.Nobs_board), dim3(64), lds_bytes, stream, P, B.R, B.joint, B.gram, ntri);
        }
        else if(with_jacobian && B.gram != NULL && !B.store_jacobian)
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,true,false,false>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.R, B.joint, B.gram);
        else if(with_jacobian && B.gram != NULL)
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,true>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.R, B.joint, B.gram);
        else if(with_jacobian)
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,false>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.R, B.joint, (double*)NULL);
        else
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,false,false>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.R, B.joint, (double*)NULL);
        if(ev_j1) hipEventRecord(ev_j1, stream);
    }
    if(!(parts & EVAL_PART_REST)) return;
    if(P.Nobs_point > 0)
    {
        if(with_jacobian)
            hipLaunchKernelGGL((point_kernel<PROJ,NDIST,true>), dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream,
                               P, B.R);
        else
            hipLaunchKernelGGL((point_kernel<PROJ,NDIST,false>), dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream,
                               P, B.R);
    }
    if(!tri_rides) launch_triangulated(P, B, with_jacobian, stream);
    const int Nreg = P.Nmeas - P.i_meas_regularization;
    // (with board observations the regularization rows were written by the prologue launch)
    if(Nreg > 0 && P.Nobs_board <= 0)
    {
        if(with_jacobian)
            hipLaunchKernelGGL((regularization_kernel<true,false>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.R, (int32_t*)NULL, (int32_t*)NULL);
        else
            hipLaunchKernelGGL((regularization_kernel<false,false>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.R, (int32_t*)NULL, (int32_t*)NULL);
    }
}

bool prologue_takes_choose(const DeviceProblem& P)
{
    // (every evaluation of a problem with boards starts with the prologue launch, the splined models' too)
    return P.Nobs_board > 0;
}
bool lens_supported(int lens_type)
{
    switch(lens_type)
    {
    case MRCAL_LENSMODEL_PINHOLE:
    case MRCAL_LENSMODEL_STEREOGRAPHIC:
    case MRCAL_LENSMODEL_LONLAT:
    case MRCAL_LENSMODEL_LATLON:
    case MRCAL_LENSMODEL_OPENCV4:
    case MRCAL_LENSMODEL_OPENCV5:
    case MRCAL_LENSMODEL_OPENCV8:
    case MRCAL_LENSMODEL_OPENCV12:
    case MRCAL_LENSMODEL_CAHVOR:
    case MRCAL_LENSMODEL_CAHVORE:
    case MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC:
        return true;
    default:
        return false;
    }
}

hipError_t launch_evaluate(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                           int lds_bytes, hipStream_t stream,
                           hipEvent_t ev_j0, hipEvent_t ev_j1, int parts)
{
    switch(P.lens_type)
    {
    case MRCAL_LENSMODEL_PINHOLE:       launch_eval_t<PROJ_OPENCV,        0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_STEREOGRAPHIC: launch_eval_t<PROJ_STEREOGRAPHIC, 0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_LONLAT:        launch_eval_t<PROJ_LONLAT,        0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_LATLON:        launch_eval_t<PROJ_LATLON,        0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_OPENCV4:       launch_eval_t<PROJ_OPENCV,        4 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_OPENCV5:       launch_eval_t<PROJ_OPENCV,        5 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_OPENCV8:       launch_eval_t<PROJ_OPENCV,        8 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_OPENCV12:      launch_eval_t<PROJ_OPENCV,        12>(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_CAHVOR:        launch_eval_t<PROJ_CAHVOR,        5 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_CAHVORE:       launch_eval_t<PROJ_CAHVORE,       8 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1,parts); break;
    case MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC: launch_eval_splined(P,B,with_jacobian,stream,ev_j0,ev_j1,parts); break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_structure(const DeviceProblem& P, const EvalBuffers& B, hipStream_t stream)
{
    if(P.Nobs_board > 0)
        hipLaunchKernelGGL(board_structure_kernel, dim3(P.Nobs_board), dim3(64), 0, stream,
                           P, B.Jp, B.Ji);
    if(P.Nobs_point > 0)
        hipLaunchKernelGGL(point_structure_kernel, dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream,
                           P, B.Jp, B.Ji);
    if(P.Npairs_tri > 0)
        hipLaunchKernelGGL((triangulated_kernel<false,true>), dim3((P.Npairs_tri + 63)/64), dim3(64), 0, stream,
                           P, B.R, B.Jp, B.Ji);
    const int Nreg = P.Nmeas - P.i_meas_regularization;
    // also writes the terminating rowptr[Nmeas] when there are regularization
    // rows; the host writes it otherwise
    if(Nreg > 0)
    {
        if(P.lens_type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
            launch_structure_splined_regularization(P, B, Nreg, stream);
        else
            hipLaunchKernelGGL((regularization_kernel<false,true>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.R, B.Jp, B.Ji);
    }
    return hipGetLastError();
}

} // namespace mrcal_amd
