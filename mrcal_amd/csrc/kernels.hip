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
//      one lane per chessboard corner, 64 corners per pass. Each lane projects
//      its corner and forms its two Jacobian rows in registers; rows go to an
//      LDS tile (odd row stride => conflict-free ds_write_b64) and the tile is
//      then streamed to HBM as one contiguous, 16-byte-per-lane coalesced run:
//      consecutive CSR rows of one observation are adjacent in J's value
//      array, so the whole 2*W*H x k tile is ONE contiguous HBM extent.
//   3. point_kernel / regularization_kernel: one lane per row pair / row.
//      These are a few hundred rows; they are launch-latency, not bandwidth.
//
// The CSR structure (rowptr, colidx) never changes between evaluations, so it
// is produced once by the *_structure kernels at problem creation.
#include <hip/hip_runtime.h>
#include "problem.hpp"
#include "device_math.hpp"
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
// lens models: q, dq/dp (2x3), dq/d(distortion) (2 x NDIST)
////////////////////////////////////////////////////////////////////////////////

// OpenCV rational + tangential + thin-prism family; NDIST = 0 is a pinhole.
// The k[] slots beyond NDIST are compile-time zeros and fold away
template<int NDIST, bool WITH_GRAD>
__device__ __forceinline__
void project_opencv(double* q, double (*dq_dp)[3], double (*dq_dk)[NDIST > 0 ? NDIST : 1],
                    const double* p, const double* intr)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    double k[12];
#pragma unroll
    for(int i=0;i<12;i++) k[i] = (i < NDIST) ? intr[4+i] : 0.0;

    const double iz = 1.0/p[2];
    const double X  = p[0]*iz;
    const double Y  = p[1]*iz;
    const double r2 = X*X + Y*Y;
    const double r4 = r2*r2;
    const double r6 = r4*r2;
    const double a1 = 2.0*X*Y;
    const double a2 = r2 + 2.0*X*X;
    const double a3 = r2 + 2.0*Y*Y;
    const double num  = 1.0 + k[0]*r2 + k[1]*r4 + k[4]*r6;
    const double iden = 1.0/(1.0 + k[5]*r2 + k[6]*r4 + k[7]*r6);
    const double xd = X*num*iden + k[2]*a1 + k[3]*a2 + k[8] *r2 + k[9] *r4;
    const double yd = Y*num*iden + k[2]*a3 + k[3]*a1 + k[10]*r2 + k[11]*r4;
    q[0] = xd*fx + cx;
    q[1] = yd*fy + cy;

    if(!WITH_GRAD) return;

    const double dX[3] = { iz,  0.0, -X*iz };
    const double dY[3] = { 0.0, iz,  -Y*iz };
#pragma unroll
    for(int j=0;j<3;j++)
    {
        const double dr2   = 2.0*X*dX[j] + 2.0*Y*dY[j];
        const double dnum  = k[0]*dr2 + 2.0*k[1]*r2*dr2 + 3.0*k[4]*r4*dr2;
        const double diden = -iden*iden*(k[5]*dr2 + 2.0*k[6]*r2*dr2 + 3.0*k[7]*r4*dr2);
        const double da1   = 2.0*(X*dY[j] + Y*dX[j]);
        const double dxd   = dX[j]*num*iden + X*dnum*iden + X*num*diden +
            k[2]*da1 + k[3]*(dr2 + 4.0*X*dX[j]) + k[8]*dr2  + 2.0*r2*k[9]*dr2;
        const double dyd   = dY[j]*num*iden + Y*dnum*iden + Y*num*diden +
            k[2]*(dr2 + 4.0*Y*dY[j]) + k[3]*da1 + k[10]*dr2 + 2.0*r2*k[11]*dr2;
        dq_dp[0][j] = fx*dxd;
        dq_dp[1][j] = fy*dyd;
    }
    if(NDIST >= 4)
    {
        dq_dk[0][0] = fx*X*iden*r2;   dq_dk[1][0] = fy*(Y*iden*r2);
        dq_dk[0][1] = fx*X*iden*r4;   dq_dk[1][1] = fy*Y*iden*r4;
        dq_dk[0][2] = fx*a1;          dq_dk[1][2] = fy*a3;
        dq_dk[0][3] = fx*a2;          dq_dk[1][3] = fy*a1;
    }
    if(NDIST >= 5)
    {
        dq_dk[0][4] = fx*X*iden*r6;   dq_dk[1][4] = fy*Y*iden*r6;
    }
    if(NDIST >= 8)
    {
        const double t = num*(-iden)*iden;
        dq_dk[0][5] = fx*X*t*r2;      dq_dk[1][5] = fy*Y*t*r2;
        dq_dk[0][6] = fx*X*t*r4;      dq_dk[1][6] = fy*Y*t*r4;
        dq_dk[0][7] = fx*X*t*r6;      dq_dk[1][7] = fy*Y*t*r6;
    }
    if(NDIST >= 12)
    {
        dq_dk[0][8]  = fx*r2;         dq_dk[1][8]  = 0.0;
        dq_dk[0][9]  = fx*r4;         dq_dk[1][9]  = 0.0;
        dq_dk[0][10] = 0.0;           dq_dk[1][10] = fy*r2;
        dq_dk[0][11] = 0.0;           dq_dk[1][11] = fy*r4;
    }
}

// q = 2 p_xy/(|p| + p_z) f + c
template<bool WITH_GRAD>
__device__ __forceinline__
void project_stereographic(double* q, double (*dq_dp)[3], const double* p, const double* intr)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double mag   = sqrt(p[0]*p[0] + p[1]*p[1] + p[2]*p[2]);
    const double scale = 2.0/(mag + p[2]);
    if(WITH_GRAD)
    {
        const double A = -scale*scale/2.0;
        const double B = A/mag;
        dq_dp[0][0] = fx*(p[0]*(B*p[0]) + scale);
        dq_dp[0][1] = fx*(p[0]*(B*p[1]));
        dq_dp[0][2] = fx*(p[0]*(B*p[2] + A));
        dq_dp[1][0] = fy*(p[1]*(B*p[0]));
        dq_dp[1][1] = fy*(p[1]*(B*p[1]) + scale);
        dq_dp[1][2] = fy*(p[1]*(B*p[2] + A));
    }
    q[0] = p[0]*scale*fx + cx;
    q[1] = p[1]*scale*fy + cy;
}

// equirectangular: q = (atan2(px,pz), asin(py/|p|)) f + c
template<bool WITH_GRAD>
__device__ __forceinline__
void project_lonlat(double* q, double (*dq_dp)[3], const double* p, const double* intr)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double in2   = 1.0/(p[0]*p[0] + p[1]*p[1] + p[2]*p[2]);
    const double im    = sqrt(in2);
    const double in2xz = 1.0/(p[0]*p[0] + p[2]*p[2]);
    const double imxz  = sqrt(in2xz);
    if(WITH_GRAD)
    {
        dq_dp[0][0] =  fx*in2xz*p[2];
        dq_dp[0][1] =  0.0;
        dq_dp[0][2] = -fx*in2xz*p[0];
        dq_dp[1][0] = -fy*imxz*(p[1]*p[0]*in2);
        dq_dp[1][1] = -fy*imxz*(p[1]*p[1]*in2 - 1.0);
        dq_dp[1][2] = -fy*imxz*(p[1]*p[2]*in2);
    }
    q[0] = atan2(p[0], p[2])*fx + cx;
    q[1] = asin(p[1]*im)    *fy + cy;
}
// transverse equirectangular: lonlat with x and y swapped
template<bool WITH_GRAD>
__device__ __forceinline__
void project_latlon(double* q, double (*dq_dp)[3], const double* p, const double* intr)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double in2   = 1.0/(p[0]*p[0] + p[1]*p[1] + p[2]*p[2]);
    const double im    = sqrt(in2);
    const double in2yz = 1.0/(p[1]*p[1] + p[2]*p[2]);
    const double imyz  = sqrt(in2yz);
    if(WITH_GRAD)
    {
        dq_dp[0][0] = -fx*imyz*(p[0]*p[0]*in2 - 1.0);
        dq_dp[0][1] = -fx*imyz*(p[0]*p[1]*in2);
        dq_dp[0][2] = -fx*imyz*(p[0]*p[2]*in2);
        dq_dp[1][0] =  0.0;
        dq_dp[1][1] =  fy*in2yz*p[2];
        dq_dp[1][2] = -fy*in2yz*p[1];
    }
    q[0] = asin(p[0]*im)    *fx + cx;
    q[1] = atan2(p[1], p[2])*fy + cy;
}

// dispatch over the "simple" (closed-form, central) models
template<int PROJ, int NDIST, bool WITH_GRAD>
__device__ __forceinline__
void project_lens(double* q, double (*dq_dp)[3], double (*dq_dk)[NDIST > 0 ? NDIST : 1],
                  const double* p, const double* intr)
{
    if     (PROJ == PROJ_OPENCV)        project_opencv<NDIST,WITH_GRAD>(q, dq_dp, dq_dk, p, intr);
    else if(PROJ == PROJ_STEREOGRAPHIC) project_stereographic<WITH_GRAD>(q, dq_dp, p, intr);
    else if(PROJ == PROJ_LONLAT)        project_lonlat<WITH_GRAD>(q, dq_dp, p, intr);
    else                                project_latlon<WITH_GRAD>(q, dq_dp, p, intr);
}

////////////////////////////////////////////////////////////////////////////////
// 2. board kernel
////////////////////////////////////////////////////////////////////////////////
//
// LDS tile of one pass (64 corners = 128 rows). Row (xy,corner) lives at LDS
// row xy*64+corner; its columns are in STATE order:
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
// The row stride is odd, which makes the per-lane column writes
// (ds_write_b64, lane stride = one row) and the MFMA operand reads hit
// distinct banks.
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
    const int  NPTS    = P.W*P.H;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;

    double intr[4 + NDIST];
#pragma unroll
    for(int i=0;i<4+NDIST;i++) intr[i] = get_intrinsic(P, b, m.icam_intrinsics, i);

    double warp[2] = {0.0, 0.0};
    if(P.has_warp_seed) get_warp(warp, P, b);

    // upper-triangular 16x16 tiles of G: up to 3 column blocks
    constexpr int NBMAX = 3;
    constexpr int NTMAX = NBMAX*(NBMAX+1)/2;
    double4_t acc[WITH_GRAM ? NTMAX : 1];
    if(WITH_GRAM)
    {
#pragma unroll
        for(int t=0;t<NTMAX;t++) acc[t] = (double4_t){0.0,0.0,0.0,0.0};
    }
    const int NB = (kx + 15) >> 4;

    for(int chunk0 = 0; chunk0 < NPTS; chunk0 += 64)
    {
        const int pt   = chunk0 + lane;
        const int npts = (NPTS - chunk0 < 64) ? (NPTS - chunk0) : 64;

        if(pt < NPTS)
        {
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

            double q[2], dq_dp[2][3], dq_dk[2][NDIST > 0 ? NDIST : 1];
            project_lens<PROJ,NDIST,WITH_J>(q, dq_dp, dq_dk, p, intr);

            const double* __restrict__ obs = P.board_pool + ((size_t)iobs*NPTS + pt)*3;
            const double qx_obs = obs[0], qy_obs = obs[1], w = obs[2];
            const bool   inlier = (w >= 0.0);

            double2 err;
            err.x = inlier ? (q[0] - qx_obs)*w : 0.0;
            err.y = inlier ? (q[1] - qy_obs)*w : 0.0;
            *reinterpret_cast<double2*>(&x[m.i_meas0 + 2*pt]) = err;

            if(WITH_J)
            {
                double* __restrict__ row[2] = { tile + (size_t)lane*ks,
                                                tile + (size_t)(64 + lane)*ks };
                // outliers keep their columns and get all-zero values
                const double ww = inlier ? w : 0.0;
                int c = 0;
                if(ncore)
                {
#pragma unroll
                    for(int xy=0;xy<2;xy++)
                    {
                        const double dq_df = (q[xy] - intr[2+xy])/intr[xy];
                        row[xy][xy]       = inlier ? dq_df * w * SCALE_INTRINSICS_FOCAL_LENGTH : 0.0;
                        row[xy][1-xy]     = 0.0;
                        row[xy][2+xy]     = ww * SCALE_INTRINSICS_CENTER_PIXEL;
                        row[xy][2+(1-xy)] = 0.0;
                    }
                    c = 4;
                }
                if(NDIST > 0 && P.Ndist_state)
                {
#pragma unroll
                    for(int xy=0;xy<2;xy++)
#pragma unroll
                        for(int i=0;i<NDIST;i++)
                            row[xy][c+i] = inlier ? dq_dk[xy][i] * w * SCALE_DISTORTION : 0.0;
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
#pragma unroll
                        for(int xy=0;xy<2;xy++)
                        {
                            const double g = dq_dp[xy][0]*dp[0] + dq_dp[xy][1]*dp[1] + dq_dp[xy][2]*dp[2];
                            row[xy][c+l]   = inlier ? g * w * SCALE_ROTATION_CAMERA : 0.0;
                            row[xy][c+3+l] = inlier ? dq_dp[xy][l] * w * SCALE_TRANSLATION_CAMERA : 0.0;
                        }
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
#pragma unroll
                        for(int xy=0;xy<2;xy++)
                        {
                            const double gr = dq_dp[xy][0]*dpr[0] + dq_dp[xy][1]*dpr[1] + dq_dp[xy][2]*dpr[2];
                            const double gt = dq_dp[xy][0]*dpt[0] + dq_dp[xy][1]*dpt[1] + dq_dp[xy][2]*dpt[2];
                            row[xy][c+l]   = inlier ? gr * w * SCALE_ROTATION_FRAME    : 0.0;
                            row[xy][c+3+l] = inlier ? gt * w * SCALE_TRANSLATION_FRAME : 0.0;
                        }
                    }
                    c += 6;
                }
                if(P.has_warp_state)
                {
                    // dq/dwarp_i = (dq/dt . Rj[:,2]) dz/dwarp_i
#pragma unroll
                    for(int xy=0;xy<2;xy++)
                    {
                        const double d =
                            dq_dp[xy][0]*jp[JOINT_R + 2] +
                            dq_dp[xy][1]*jp[JOINT_R + 5] +
                            dq_dp[xy][2]*jp[JOINT_R + 8];
                        row[xy][c+0] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[0]) : 0.0;
                        row[xy][c+1] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[1]) : 0.0;
                    }
                    c += 2;
                }
                if(WITH_GRAM)
                {
                    row[0][c] = err.x;
                    row[1][c] = err.y;
                }
            }
        }

        if(WITH_J)
        {
            __syncthreads();
            // Stream the tile out. Output element e of this pass is CSR row
            // e/k, column e%k, with row = 2*point + xy. 16 bytes per lane per
            // store, 1 KiB contiguous per wave instruction
            const int nelem = 2*npts*k;
            double* __restrict__ out = Jv + m.i_nnz0 + (size_t)2*chunk0*k;
            for(int e = 2*lane; e < nelem; e += 128)
            {
                int r0 = e / k;
                int c0 = e - r0*k;
                int r1 = r0, c1 = c0 + 1;
                if(c1 == k) { c1 = 0; r1++; }
                // CSR column -> tile column: the row's own 2 core columns
                // (f then c) sit at tile columns xy and 2+xy
                const int xy0 = r0 & 1, xy1 = r1 & 1;
                const int t0 = ncore ? ((c0 < 2) ? (2*c0 + xy0) : (c0 + 2)) : c0;
                const int t1 = ncore ? ((c1 < 2) ? (2*c1 + xy1) : (c1 + 2)) : c1;
                double2 v;
                v.x = tile[(size_t)((xy0 << 6) + (r0 >> 1))*ks + t0];
                v.y = tile[(size_t)((xy1 << 6) + (r1 >> 1))*ks + t1];
                *reinterpret_cast<double2*>(&out[e]) = v;
            }

            if(WITH_GRAM)
            {
                // G += Tt T over this pass. 4 tile rows per k-step; rows of
                // corners beyond npts hold stale data and are masked out
                const int nsteps = (npts + 3) >> 2;
                const int col    = lane & 15;
                for(int plane = 0; plane < 2; plane++)
                    for(int s = 0; s < nsteps; s++)
                    {
                        const int  rr    = 4*s + (lane >> 4);
                        const bool valid = rr < npts;
                        const double* __restrict__ trow = tile + (size_t)((plane << 6) + rr)*ks;
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
            __syncthreads();
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
    project_lens<PROJ,NDIST,WITH_J>(q, dq_dp, dq_dk, p, intr);

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
            hipLaunchKernelGGL((board_kernel<PROJ,NDIST,false,false>), dim3(P.Nobs_board), dim3(64), 0, stream,
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
