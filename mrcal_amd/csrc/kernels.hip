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
//      one lane per Jacobian row, 64 rows (32 corners) per pass. Each lane
//      projects its corner and forms its CSR row in registers; rows go to an
//      LDS tile (odd row stride => conflict-free ds_write_b64) and the tile is
//      then streamed to HBM as one contiguous, 16-byte-per-lane coalesced run:
//      consecutive CSR rows of one observation are adjacent in J's value
//      array, so the whole 2*W*H x k block is ONE contiguous HBM extent.
//   3. point_kernel / regularization_kernel: one lane per row pair / row.
//      These are a few hundred rows; they are launch-latency, not bandwidth.
//
// The CSR structure (rowptr, colidx) never changes between evaluations, so it
// is produced once by the *_structure kernels at problem creation.
#include <hip/hip_runtime.h>
#include "problem.hpp"
#include "device_math.hpp"
#include "lens_models.hpp"
#include "kernels.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// state access: packed state b[] (if the block is being optimized) or seeds
////////////////////////////////////////////////////////////////////////////////
__device__ __forceinline__
double get_intrinsic(const DeviceProblem& P, const double* __restrict__ b, int icam, int i)
{
    if(i < P.Ncore)
    {
        if(P.Ncore_state)
            return b[P.i_state_intrinsics + icam*P.Nintr_state + i] *
                ((i < 2) ? SCALE_INTRINSICS_FOCAL_LENGTH : SCALE_INTRINSICS_CENTER_PIXEL);
        return P.seed_intrinsics[icam*P.Nintrinsics + i];
    }
    if(P.Ndist_state)
        return b[P.i_state_intrinsics + icam*P.Nintr_state + P.Ncore_state + (i - P.Ncore)] * SCALE_DISTORTION;
    return P.seed_intrinsics[icam*P.Nintrinsics + i];
}
__device__ __forceinline__
void get_rt_cam_ref(double* rt, const DeviceProblem& P, const double* __restrict__ b, int icam_extrinsics)
{
    if(P.do_optimize_extrinsics)
    {
        const double* s = &b[P.i_state_extrinsics + 6*icam_extrinsics];
        for(int i=0;i<3;i++) rt[i]   = s[i]   * SCALE_ROTATION_CAMERA;
        for(int i=0;i<3;i++) rt[3+i] = s[3+i] * SCALE_TRANSLATION_CAMERA;
    }
    else
        for(int i=0;i<6;i++) rt[i] = P.seed_rt_cam_ref[6*icam_extrinsics + i];
}
__device__ __forceinline__
void get_rt_ref_frame(double* rt, const DeviceProblem& P, const double* __restrict__ b, int iframe)
{
    if(P.do_optimize_frames)
    {
        const double* s = &b[P.i_state_frames + 6*iframe];
        for(int i=0;i<3;i++) rt[i]   = s[i]   * SCALE_ROTATION_FRAME;
        for(int i=0;i<3;i++) rt[3+i] = s[3+i] * SCALE_TRANSLATION_FRAME;
    }
    else
        for(int i=0;i<6;i++) rt[i] = P.seed_rt_ref_frame[6*iframe + i];
}
__device__ __forceinline__
void get_warp(double* w, const DeviceProblem& P, const double* __restrict__ b)
{
    if(P.has_warp_state)
    {
        w[0] = b[P.i_state_warp+0] * SCALE_CALOBJECT_WARP;
        w[1] = b[P.i_state_warp+1] * SCALE_CALOBJECT_WARP;
    }
    else
    {
        w[0] = P.seed_warp[0];
        w[1] = P.seed_warp[1];
    }
}

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

__global__ __launch_bounds__(64)
void board_prologue_kernel(DeviceProblem P, const double* __restrict__ b, double* __restrict__ joint)
{
    const int iobs = blockIdx.x*blockDim.x + threadIdx.x;
    if(iobs >= P.Nobs_board) return;
    const BoardObsMeta m = P.board_meta[iobs];

    double rt_frame[6], rt_cam[6];
    get_rt_ref_frame(rt_frame, P, b, m.iframe);
    double rec[JOINT_STRIDE];
    if(m.icam_extrinsics >= 0)
    {
        get_rt_cam_ref(rt_cam, P, b, m.icam_extrinsics);
        joint_pose_record(rec, rt_cam, rt_frame);
    }
    else
        joint_pose_record(rec, NULL, rt_frame);
    double* out = joint + (size_t)iobs*JOINT_STRIDE;
    for(int i=0;i<JOINT_STRIDE;i++) out[i] = rec[i];
}

////////////////////////////////////////////////////////////////////////////////
// 2. board kernel
////////////////////////////////////////////////////////////////////////////////
//
// One wavefront per board observation, ONE LANE PER JACOBIAN ROW, 64 rows (32
// corners x {qx,qy}) per pass. A lane projects its corner, evaluates its own
// image row of the projection gradient, and forms the k values of its CSR row
// in registers.
//
// LDS tile of one pass: 64 rows x ks doubles (13.8 KB at 24 nonzeros per row:
// 11 waves per CU fit in the 160 KB LDS, which is what lets the stores of one
// wave overlap the arithmetic and the MFMAs of the others). A row's columns are
// in STATE order:
//
//   [fx fy cx cy]   if the core is optimized. An x row holds (dq/dfx, 0, w, 0),
//                   a y row (0, dq/dfy, 0, w): each CSR row carries only its
//                   own 2 core columns, the tile carries all 4 so that a tile
//                   column means the same state variable in every row
//   [distortions]   Ndist_state
//   [r_cam t_cam]   6, if this camera has extrinsics in the state
//   [r_frame t_frame] 6, if frames are optimized
//   [warp]          2, if the warp is optimized
//   [x]             the residual itself (only when the Gram is being formed)
//
// The row stride ks is odd, which makes the per-lane column writes
// (ds_write_b64, lane stride = one row) hit distinct banks.
//
// Copy-out: consecutive CSR rows of one observation are adjacent in J's value
// array, so the 64 x k tile is ONE contiguous HBM extent; it is streamed out 16
// bytes per lane, 1 KiB contiguous per wave instruction (measured: this store
// pattern reaches 5.9 TB/s, a row-per-lane direct store 3.4 TB/s;
// tools/exp/store_patterns.hip).
//
// Gram (WITH_GRAM): G = Tt T over the tile columns, accumulated over the
// passes of the observation with v_mfma_f64_16x16x4_f64. One k-step is 4 tile
// rows; lane l supplies T[4s + l/16][16b + l%16] for column block b, which is
// simultaneously the A operand (A[i=l%16][k=l/16]) of row block b and the B
// operand (B[k=l/16][j=l%16]) of column block b. Accumulator register v of
// lane l holds G[16bi + l/16 + 4v][16bj + l%16] (layout measured on gfx950,
// tools/mfma_f64_layout_probe.hip). The last column of G is Tt x = the
// observation's slice of Jt x, its corner is |x|^2.
typedef double double4_t __attribute__((ext_vector_type(4)));

template<int PROJ, int NDIST, bool WITH_J, bool WITH_GRAM>
__global__ __launch_bounds__(64)
void board_kernel(DeviceProblem P,
                  const double* __restrict__ b,
                  const double* __restrict__ joint,
                  double*       __restrict__ x,
                  double*       __restrict__ Jv,
                  double*       __restrict__ gram)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];

    const int iobs = blockIdx.x;
    const int lane = threadIdx.x;
    const BoardObsMeta m = P.board_meta[iobs];
    const double* __restrict__ jp = joint + (size_t)iobs*JOINT_STRIDE;

    const int  k       = m.nnz_per_row;
    const int  ncore   = P.Ncore_state;          // 0 or 4
    const int  kt      = k + (ncore ? 2 : 0);    // tile columns holding J
    const int  kx      = kt + (WITH_GRAM ? 1 : 0);
    const int  ks      = kx | 1;                 // odd LDS row stride
    const int  ks_alloc = P.board_tile_stride;   // the stride the LDS allocation was sized for (>= ks)
    const int  NPTS    = P.W*P.H;
    const int  NROWS   = 2*NPTS;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;

    // filled without a loop: see the note in lens_models.hpp
    double intr[4 + NDIST];
#define MRCAL_AMD_LOAD_INTR(i) if((i) < 4+NDIST) intr[(i) < 4+NDIST ? (i) : 0] = get_intrinsic(P, b, m.icam_intrinsics, (i))
    MRCAL_AMD_LOAD_INTR(0);  MRCAL_AMD_LOAD_INTR(1);  MRCAL_AMD_LOAD_INTR(2);  MRCAL_AMD_LOAD_INTR(3);
    MRCAL_AMD_LOAD_INTR(4);  MRCAL_AMD_LOAD_INTR(5);  MRCAL_AMD_LOAD_INTR(6);  MRCAL_AMD_LOAD_INTR(7);
    MRCAL_AMD_LOAD_INTR(8);  MRCAL_AMD_LOAD_INTR(9);  MRCAL_AMD_LOAD_INTR(10); MRCAL_AMD_LOAD_INTR(11);
    MRCAL_AMD_LOAD_INTR(12); MRCAL_AMD_LOAD_INTR(13); MRCAL_AMD_LOAD_INTR(14); MRCAL_AMD_LOAD_INTR(15);
#undef MRCAL_AMD_LOAD_INTR
    static_assert(4 + NDIST <= 16, "extend the list above");
    const double intr_fx = intr[0], intr_fy = intr[1], intr_cx = intr[2], intr_cy = intr[3];

    double warp[2] = {0.0, 0.0};
    if(P.has_warp_seed) get_warp(warp, P, b);

    // upper-triangular 16x16 tiles of G: up to 3 column blocks
    constexpr int NBMAX = GRAM_NB_MAX;
    constexpr int NTMAX = GRAM_NT_MAX;
    double4_t acc[WITH_GRAM ? NTMAX : 1];
    if(WITH_GRAM)
    {
#pragma unroll
        for(int t=0;t<NTMAX;t++) acc[t] = (double4_t){0.0,0.0,0.0,0.0};
    }
    const int NB = (kx + 15) >> 4;

    // copy-out stepping: element index e advances by 128 per iteration
    const int step_rows = 128 / k;
    const int step_cols = 128 - step_rows*k;

    const bool isy = (lane & 1) != 0;            // passes start at even rows

    // HW_REG_HW_ID[3:0]: this wave's slot in its SIMD
    const int order = (P.debug_ablate & 8) ? 0 : (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1);

    // The observation's pixels and weights (NPTS x 3 doubles) are staged in
    // LDS up front, behind the tile. After that this wave issues NO vector
    // loads: its stores are fire-and-forget, and nothing in the pass loop
    // waits on vmcnt. (A global load inside the loop would have to wait for
    // every store issued before it: vmcnt retires in order, and with a
    // data-dependent number of stores per pass the compiler can only wait for
    // vmcnt(0), which serializes the HBM write stream with the arithmetic.)
    double* __restrict__ obs_lds = tile + 64*ks_alloc;
    {
        const double* __restrict__ pool = P.board_pool + (size_t)iobs*NPTS*3;
        for(int i = lane; i < 3*NPTS; i += 64) obs_lds[i] = pool[i];
    }
    __builtin_amdgcn_wave_barrier();

    for(int row0 = 0; row0 < NROWS; row0 += 64)
    {
        const int r     = row0 + lane;
        const int nrows = (NROWS - row0 < 64) ? (NROWS - row0) : 64;

        if(r < NROWS)
        {
            const int pt = r >> 1;
            const int iy = pt / P.W;
            const int ix = pt - iy*P.W;
            const double bx = (double)ix * P.spacing;
            const double by = (double)iy * P.spacing;
            double bz = 0.0, dz_dw[2] = {0.0, 0.0};
            if(P.has_warp_seed)
            {
                // parabolic flex along each board axis, max deflection at the centre
                const double xr = (double)ix / (double)(P.W - 1);
                const double yr = (double)iy / (double)(P.H - 1);
                dz_dw[0] = 4.0*xr*(1.0 - xr);
                dz_dw[1] = 4.0*yr*(1.0 - yr);
                bz += warp[0]*dz_dw[0];
                bz += warp[1]*dz_dw[1];
            }

            double p[3];
#pragma unroll
            for(int i=0;i<3;i++)
                p[i] = jp[JOINT_R+3*i+0]*bx + jp[JOINT_R+3*i+1]*by + jp[JOINT_R+3*i+2]*bz + jp[JOINT_T+i];

            double q, dq_dp[3], dq_dk[NDIST > 0 ? NDIST : 1];
            if(P.debug_ablate & 4)
            {
                q = p[0]; dq_dp[0] = p[0]; dq_dp[1] = p[1]; dq_dp[2] = p[2];
#pragma unroll
                for(int i=0;i<NDIST;i++) dq_dk[i] = p[0] + (double)i;
            }
            else
            project_lens_row<PROJ,NDIST,WITH_J>(lane & 1, &q, dq_dp, dq_dk, p, intr, P.cfg);

            const double q_obs  = obs_lds[3*pt + (lane & 1)];
            const double w      = obs_lds[3*pt + 2];
            const bool   inlier = (w >= 0.0);

            const double err = inlier ? (q - q_obs)*w : 0.0;
            x[m.i_meas0 + r] = err;

            if(WITH_J)
            {
                double* __restrict__ row = tile + (size_t)lane*ks;
                // outliers keep their columns and get all-zero values
                const double ww = inlier ? w : 0.0;
                int c = 0;
                if(ncore)
                {
                    const double f  = isy ? intr_fy : intr_fx;
                    const double cc = isy ? intr_cy : intr_cx;
                    const double vf = inlier ? ((q - cc)/f) * w * SCALE_INTRINSICS_FOCAL_LENGTH : 0.0;
                    const double vc = ww * SCALE_INTRINSICS_CENTER_PIXEL;
                    row[0] = isy ? 0.0 : vf;
                    row[1] = isy ? vf  : 0.0;
                    row[2] = isy ? 0.0 : vc;
                    row[3] = isy ? vc  : 0.0;
                    c = 4;
                }
                if(NDIST > 0 && P.Ndist_state)
                {
#pragma unroll
                    for(int i=0;i<NDIST;i++)
                        row[c+i] = inlier ? dq_dk[i] * w * SCALE_DISTORTION : 0.0;
                    c += NDIST;
                }
                if(has_ext)
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
                        const double g = dq_dp[0]*dp[0] + dq_dp[1]*dp[1] + dq_dp[2]*dp[2];
                        row[c+l]   = inlier ? g * w * SCALE_ROTATION_CAMERA : 0.0;
                        row[c+3+l] = inlier ? dq_dp[l] * w * SCALE_TRANSLATION_CAMERA : 0.0;
                    }
                    c += 6;
                }
                if(P.do_optimize_frames)
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
                        const double gr = dq_dp[0]*dpr[0] + dq_dp[1]*dpr[1] + dq_dp[2]*dpr[2];
                        const double gt = dq_dp[0]*dpt[0] + dq_dp[1]*dpt[1] + dq_dp[2]*dpt[2];
                        row[c+l]   = inlier ? gr * w * SCALE_ROTATION_FRAME    : 0.0;
                        row[c+3+l] = inlier ? gt * w * SCALE_TRANSLATION_FRAME : 0.0;
                    }
                    c += 6;
                }
                if(P.has_warp_state)
                {
                    // dq/dwarp_i = (dq/dt . Rj[:,2]) dz/dwarp_i
                    const double d =
                        dq_dp[0]*jp[JOINT_R + 2] +
                        dq_dp[1]*jp[JOINT_R + 5] +
                        dq_dp[2]*jp[JOINT_R + 8];
                    row[c+0] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[0]) : 0.0;
                    row[c+1] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[1]) : 0.0;
                    c += 2;
                }
                if(WITH_GRAM)
                    row[c] = err;
            }
        }

        if(WITH_J)
        {
            // The workgroup IS one wavefront, and the LDS executes a wave's DS
            // instructions in order: the tile reads below see the writes above
            // without an s_barrier. __syncthreads() would also wait for this
            // wave's outstanding global stores (vmcnt(0)), i.e. serialize the
            // HBM write stream of a pass with the arithmetic of the next one
            __builtin_amdgcn_wave_barrier();
            // Two phases read the tile: the copy-out (vector-memory bound)
            // and the Gram (MFMA bound). Waves alternate the order by the
            // parity of their hardware wave slot, so that co-resident waves
            // of a SIMD lean on different pipes at the same time
            for(int phase = 0; phase < 2; phase++)
            {
                if((phase ^ order) == 0)
                {
                    // Stream the tile out. Output element e of this pass is CSR row
                    // e/k, column e%k. 16 bytes per lane per store, 1 KiB contiguous
                    // per wave instruction. nrows is even, so is the element count
                    const int nelem = nrows*k;
                    double* __restrict__ out = Jv + m.i_nnz0 + (size_t)row0*k;
                    int e  = 2*lane;
                    int r0 = e / k;
                    int c0 = e - r0*k;
                    if(!(P.debug_ablate & 1))
                    for(; e < nelem; e += 128)
                    {
                        int r1 = r0, c1 = c0 + 1;
                        if(c1 == k) { c1 = 0; r1++; }
                        // CSR column -> tile column: the row's own 2 core columns
                        // (f then c) sit at tile columns xy and 2+xy
                        const int xy0 = r0 & 1, xy1 = r1 & 1;
                        const int t0 = ncore ? ((c0 < 2) ? (2*c0 + xy0) : (c0 + 2)) : c0;
                        const int t1 = ncore ? ((c1 < 2) ? (2*c1 + xy1) : (c1 + 2)) : c1;
                        double2 v;
                        v.x = tile[r0*ks + t0];
                        v.y = tile[r1*ks + t1];
                        *reinterpret_cast<double2*>(&out[e]) = v;
                        c0 += step_cols; r0 += step_rows;
                        if(c0 >= k) { c0 -= k; r0++; }
                    }

                }
                else
                {
                    if(WITH_GRAM && !(P.debug_ablate & 2))
                    {
                        // G += Tt T over this pass. 4 tile rows per k-step; rows
                        // beyond nrows hold stale data and are masked out
                        const int nsteps = (nrows + 3) >> 2;
                        const int col    = lane & 15;
                        for(int s = 0; s < nsteps; s++)
                        {
                            const int  rr    = 4*s + (lane >> 4);
                            const bool valid = rr < nrows;
                            const double* __restrict__ trow = tile + rr*ks;
                            double a[NBMAX];
#pragma unroll
                            for(int bb=0;bb<NBMAX;bb++)
                            {
                                const int cc = 16*bb + col;
                                a[bb] = (bb < NB && valid && cc < kx) ? trow[cc] : 0.0;
                            }
                            int t = 0;
#pragma unroll
                            for(int bi=0;bi<NBMAX;bi++)
#pragma unroll
                                for(int bj=bi;bj<NBMAX;bj++,t++)
                                    if(bj < NB)
                                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    if(WITH_J && WITH_GRAM)
    {
        // accumulator layout, tile-major: gram[iobs][t][v][lane]. Only the
        // tiles in use are written
        double* __restrict__ g = gram + (size_t)iobs*GRAM_STRIDE;
        int t = 0;
#pragma unroll
        for(int bi=0;bi<NBMAX;bi++)
#pragma unroll
            for(int bj=bi;bj<NBMAX;bj++,t++)
                if(bj < NB)
                {
#pragma unroll
                    for(int v=0;v<4;v++)
                        g[(size_t)t*256 + v*64 + lane] = acc[t][v];
                }
    }
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
            if(c < P.Ndist_state)
                col = m.i_state_intrinsics + P.Ncore_state + c;
            else
            {
                c -= P.Ndist_state;
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
void point_kernel(DeviceProblem P,
                  const double* __restrict__ b,
                  double*       __restrict__ x,
                  double*       __restrict__ Jv)
{
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
template<bool WITH_J, bool WITH_STRUCTURE>
__global__ __launch_bounds__(64)
void regularization_kernel(DeviceProblem P,
                           const double* __restrict__ b,
                           double*       __restrict__ x,
                           double*       __restrict__ Jv,
                           int32_t*      __restrict__ rowptr,
                           int32_t*      __restrict__ colidx)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
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

////////////////////////////////////////////////////////////////////////////////
// launchers
////////////////////////////////////////////////////////////////////////////////
template<int PROJ, int NDIST>
static void launch_eval_t(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                          int lds_bytes, hipStream_t stream,
                          hipEvent_t ev_j0, hipEvent_t ev_j1)
{
    if(P.Nobs_board > 0)
    {
        hipLaunchKernelGGL(board_prologue_kernel, dim3((P.Nobs_board + 63)/64), dim3(64), 0, stream,
                           P, B.b, B.joint);
        if(ev_j0) hipEventRecord(ev_j0, stream);
        if(with_jacobian && B.gram != NULL)
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,true>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.b, B.joint, B.x, B.Jv, B.gram);
        else if(with_jacobian)
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,true,false>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.b, B.joint, B.x, B.Jv, (double*)NULL);
        else
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,false,false>), dim3(P.Nobs_board), dim3(64), lds_bytes, stream,
                               P, B.b, B.joint, B.x, B.Jv, (double*)NULL);
        if(ev_j1) hipEventRecord(ev_j1, stream);
    }
    if(P.Nobs_point > 0)
    {
        if(with_jacobian)
            hipLaunchKernelGGL((point_kernel<PROJ,NDIST,true>), dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream,
                               P, B.b, B.x, B.Jv);
        else
            hipLaunchKernelGGL((point_kernel<PROJ,NDIST,false>), dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream,
                               P, B.b, B.x, B.Jv);
    }
    const int Nreg = P.Nmeas - P.i_meas_regularization;
    if(Nreg > 0)
    {
        if(with_jacobian)
            hipLaunchKernelGGL((regularization_kernel<true,false>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.b, B.x, B.Jv, (int32_t*)NULL, (int32_t*)NULL);
        else
            hipLaunchKernelGGL((regularization_kernel<false,false>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.b, B.x, B.Jv, (int32_t*)NULL, (int32_t*)NULL);
    }
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
        return true;
    default:
        return false;
    }
}

hipError_t launch_evaluate(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                           int lds_bytes, hipStream_t stream,
                           hipEvent_t ev_j0, hipEvent_t ev_j1)
{
    switch(P.lens_type)
    {
    case MRCAL_LENSMODEL_PINHOLE:       launch_eval_t<PROJ_OPENCV,        0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_STEREOGRAPHIC: launch_eval_t<PROJ_STEREOGRAPHIC, 0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_LONLAT:        launch_eval_t<PROJ_LONLAT,        0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_LATLON:        launch_eval_t<PROJ_LATLON,        0 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_OPENCV4:       launch_eval_t<PROJ_OPENCV,        4 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_OPENCV5:       launch_eval_t<PROJ_OPENCV,        5 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_OPENCV8:       launch_eval_t<PROJ_OPENCV,        8 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_OPENCV12:      launch_eval_t<PROJ_OPENCV,        12>(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_CAHVOR:        launch_eval_t<PROJ_CAHVOR,        5 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
    case MRCAL_LENSMODEL_CAHVORE:       launch_eval_t<PROJ_CAHVORE,       8 >(P,B,with_jacobian,lds_bytes,stream,ev_j0,ev_j1); break;
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
    const int Nreg = P.Nmeas - P.i_meas_regularization;
    // also writes the terminating rowptr[Nmeas] when there are regularization
    // rows; the host writes it otherwise
    if(Nreg > 0)
        hipLaunchKernelGGL((regularization_kernel<false,true>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                           P, B.b, B.x, B.Jv, B.Jp, B.Ji);
    return hipGetLastError();
}

} // namespace mrcal_amd
