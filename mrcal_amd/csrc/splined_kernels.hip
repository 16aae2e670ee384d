// The splined-stereographic models' evaluation kernels and their launches
// (round 6: cut out of kernels.hip; kernels_shared.hpp has what the two units share)
#include <hip/hip_runtime.h>
#include <utility>
#include <stdlib.h>
#include "problem.hpp"
#include "device_math.hpp"
#include "lens_models.hpp"
#include "kernels.hpp"
#include "kernels_shared.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// 5. SPLINED_STEREOGRAPHIC: its own kernels
////////////////////////////////////////////////////////////////////////////////
// The intrinsics columns of a row are the (order+1)^2 control points of one of
// the two spline surfaces around the projected point: WHICH state variables a
// row touches depends on the data, so colidx is rewritten by every evaluation
// and the per-observation Gram of the parametric models does not apply (the
// normal equations of these problems are assembled from the rows, observation
// by observation: assemble_splined_kernel, assembly_splined.hip). One lane per
// chessboard corner, rows written directly.
// Reference: mrcal.c:2075-2293 (projection), 4734-4760 (row layout)
// row stride of the staging tile of board_splined_kernel (doubles; entries per row <= 2 + 16 + 6 + 6 + 2): odd
#define SPLB_KT 33
template<bool WITH_J>
__global__ __launch_bounds__(64)
void board_splined_kernel(DeviceProblem P, OpRef R, const double* __restrict__ joint,
                          int32_t* __restrict__ colidx)
{
    if(opref_skip(R)) return;
    double* __restrict__ x  = opref_get(R).x;
    double* __restrict__ Jv = opref_get(R).Jv;
    const int NPTS = P.W*P.H;
    const int lane = threadIdx.x;
    // (the lanes past the last corner take part in the cooperative copy-out below: they repeat the last corner's
    //  arithmetic and store nothing)
    const int  gi_   = blockIdx.x*blockDim.x + threadIdx.x;
    const bool valid = gi_ < P.Nobs_board*NPTS;
    const int  gi    = valid ? gi_ : P.Nobs_board*NPTS - 1;
    const int iobs = gi / NPTS;
    const int pt   = gi - iobs*NPTS;
    const BoardObsMeta m = P.board_meta[iobs];
    const double* __restrict__ jp   = joint + (size_t)iobs*JOINT_STRIDE;
    const double* __restrict__ intr = P.unpacked + (size_t)m.icam_intrinsics*P.Nintrinsics;
    const double* __restrict__ wp   = P.unpacked + (size_t)P.Ncameras_intrinsics*P.Nintrinsics;
    const int  k       = m.nnz_per_row;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;

    const int iy = pt / P.W;
    const int ix = pt - iy*P.W;
    const double bx = (double)ix * P.spacing;
    const double by = (double)iy * P.spacing;
    double bz = 0.0, dz_dw[2] = {0.0, 0.0};
    if(P.has_warp_seed)
    {
        const double xr = (double)ix / (double)(P.W - 1);
        const double yr = (double)iy / (double)(P.H - 1);
        dz_dw[0] = 4.0*xr*(1.0 - xr);
        dz_dw[1] = 4.0*yr*(1.0 - yr);
        bz += wp[0]*dz_dw[0];
        bz += wp[1]*dz_dw[1];
    }
    double p[3];
    for(int i=0;i<3;i++)
        p[i] = jp[JOINT_R+3*i+0]*bx + jp[JOINT_R+3*i+1]*by + jp[JOINT_R+3*i+2]*bz + jp[JOINT_T+i];

    double q[2], dq_dp[2][3], dq_dfxy[2], cfx[4], cfy[4];
    int ivar0;
    project_splined<WITH_J>(q, dq_dp, dq_dfxy, &ivar0, cfx, cfy, p, intr, P.cfg);

    const double* __restrict__ obs = P.board_pool + ((size_t)iobs*NPTS + pt)*3;
    const double w = obs[2];
    const bool inlier = (w >= 0.0);
    if(valid)
    {
        x[m.i_meas0 + 2*pt + 0] = inlier ? (q[0] - obs[0])*w : 0.0;
        x[m.i_meas0 + 2*pt + 1] = inlier ? (q[1] - obs[1])*w : 0.0;
    }
    if(!WITH_J) return;

    // The box of control points under each observation, for the assembly of the normal equations (it used to find
    // the box itself, from the column indices: 25k of a workgroup's 140k cycles). The inliers' patches only: an
    // outlier's rows are zero. A wave's corners belong to one or two observations (more if a board has fewer than
    // 64 corners): one reduction and four integer atomics per observation and wave - any order, the same box
    int* __restrict__ box = opref_get(R).spl_box;
    if(box != NULL && P.Ndist_state)
    {
        const int  n1   = P.cfg.spline_order + 1;
        const int  knot = (ivar0 - 4) >> 1;
        const int  kiy  = knot / P.cfg.spline_Nx, kix = knot - kiy*P.cfg.spline_Nx;
        const bool ok   = valid && inlier;
        unsigned long long todo = __ballot(ok);
        while(todo)
        {
            const int  lead = __ffsll((long long)todo) - 1;
            const int  ob   = __shfl(iobs, lead);
            const bool in   = ok && iobs == ob;
            int x0 = in ? kix : 0x7fffffff, x1 = in ? kix + n1 - 1 : -1;
            int y0 = in ? kiy : 0x7fffffff, y1 = in ? kiy + n1 - 1 : -1;
            for(int off = 32; off > 0; off >>= 1)
            {
                x0 = min(x0, __shfl_xor(x0, off)); x1 = max(x1, __shfl_xor(x1, off));
                y0 = min(y0, __shfl_xor(y0, off)); y1 = max(y1, __shfl_xor(y1, off));
            }
            if(lane == lead)
            {
                atomicMin(&box[4*ob + 0], x0); atomicMax(&box[4*ob + 1], x1);
                atomicMin(&box[4*ob + 2], y0); atomicMax(&box[4*ob + 3], y1);
            }
            todo &= ~__ballot(in);
        }
    }

    // The rows go through LDS, 32 corners (64 rows) at a time, and leave as ONE contiguous stream: the rows of
    // consecutive corners - and of consecutive observations with the same number of entries per row - are adjacent
    // in the CSR arrays. (A lane storing its own rows, as this kernel did, is 64 eight-byte segments 250 bytes
    // apart per instruction: 62 us for 46 MB at BASELINE configuration 2.) Tile: the x rows of the 32 corners in
    // rows 0..31, their y rows in rows 32..63 (an odd row stride and one row per lane and pass of the loop below:
    // no bank conflicts); the values first, the column indices behind them. The columns of the spline's control
    // points are the only ones that depend on the data (the others were written at set-up)
    extern __shared__ double lds_spl[];
    double*  __restrict__ lv = lds_spl;
    int32_t* __restrict__ lc = (int32_t*)(lds_spl + 64*SPLB_KT);
    const long long dest0 = m.i_nnz0 + (long long)(2*pt)*k;          // this corner's first entry
    const int cs = P.Ncore_state ? 2 : 0;                             // the control points' columns: [cs, cs + ns)
    const int ns = P.Ndist_state ? (P.cfg.spline_order + 1)*(P.cfg.spline_order + 1) : 0;
    for(int h = 0; h < 2; h++)
    {
    const bool mine = valid && (lane >> 5) == h;
    if(mine)
    {

    const double ww = inlier ? w : 0.0;     // outliers: same columns, zero values
    const int n = P.cfg.spline_order + 1;
    for(int xy=0;xy<2;xy++)
    {
        double*  __restrict__ row = lv + ((lane & 31) + 32*xy)*SPLB_KT;
        int32_t* __restrict__ ci  = lc + ((lane & 31) + 32*xy)*SPLB_KT;
        int c = 0;
        if(P.Ncore_state)
        {
            row[c++] = inlier ? dq_dfxy[xy] * w * SCALE_INTRINSICS_FOCAL_LENGTH : 0.0;
            row[c++] = ww * SCALE_INTRINSICS_CENTER_PIXEL;
        }
        if(P.Ndist_state)
        {
            const int col0 = m.i_state_intrinsics + P.Ncore_state + (ivar0 - 4);
            for(int jy=0;jy<n;jy++)
                for(int jx=0;jx<n;jx++)
                {
                    ci[c]    = col0 + jy*2*P.cfg.spline_Nx + jx*2 + xy;
                    row[c++] = inlier ? cfx[jx]*cfy[jy]*intr[xy] * w * SCALE_DISTORTION : 0.0;
                }
        }
        if(has_ext)
        {
            for(int l=0;l<3;l++)
            {
                double dp[3];
                for(int i=0;i<3;i++)
                    dp[i] =
                        bx*jp[JOINT_MC + 0  + 3*i + l] +
                        by*jp[JOINT_MC + 9  + 3*i + l] +
                        bz*jp[JOINT_MC + 18 + 3*i + l] +
                        jp[JOINT_DTJ_DRC + 3*i + l];
                const double g = dq_dp[xy][0]*dp[0] + dq_dp[xy][1]*dp[1] + dq_dp[xy][2]*dp[2];
                row[c+l]   = inlier ? g * w * SCALE_ROTATION_CAMERA : 0.0;
                row[c+3+l] = inlier ? dq_dp[xy][l] * w * SCALE_TRANSLATION_CAMERA : 0.0;
            }
            c += 6;
        }
        if(P.do_optimize_frames)
        {
            for(int l=0;l<3;l++)
            {
                double dpr[3], dpt[3];
                for(int i=0;i<3;i++)
                {
                    dpr[i] =
                        bx*jp[JOINT_MF + 0  + 3*i + l] +
                        by*jp[JOINT_MF + 9  + 3*i + l] +
                        bz*jp[JOINT_MF + 18 + 3*i + l];
                    dpt[i] = jp[JOINT_DTJ_DTF + 3*i + l];
                }
                const double gr = dq_dp[xy][0]*dpr[0] + dq_dp[xy][1]*dpr[1] + dq_dp[xy][2]*dpr[2];
                const double gt = dq_dp[xy][0]*dpt[0] + dq_dp[xy][1]*dpt[1] + dq_dp[xy][2]*dpt[2];
                row[c+l]   = inlier ? gr * w * SCALE_ROTATION_FRAME    : 0.0;
                row[c+3+l] = inlier ? gt * w * SCALE_TRANSLATION_FRAME : 0.0;
            }
            c += 6;
        }
        if(P.has_warp_state)
        {
            const double d =
                dq_dp[xy][0]*jp[JOINT_R + 2] +
                dq_dp[xy][1]*jp[JOINT_R + 5] +
                dq_dp[xy][2]*jp[JOINT_R + 8];
            row[c+0] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[0]) : 0.0;
            row[c+1] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[1]) : 0.0;
        }
    }
    }   // mine
    __builtin_amdgcn_wave_barrier();
    // the half's rows are one extent if its corners have the same number of entries per row and follow each other
    // (always within an observation; across two of them unless one camera sits at the reference and the other does not)
    const unsigned long long live = __ballot(mine);
    if(live == 0ull) break;
    const int       lead = 32*h;
    const long long base = __shfl(dest0, lead);
    const int       kk   = __shfl(k, lead);
    const int       nrow = 2*__popcll(live);
    const bool uniform = __all(!mine || (k == kk && dest0 == base + (long long)(2*(lane & 31))*kk));
    if(uniform)
    {
        const int total = nrow*kk;
        // (r, c): CSR row of the half and entry of element e, kept by increments (one division here, none in the loop)
        const int dr = 64 / kk, dc = 64 - dr*kk;
        int r = lane / kk, c = lane - r*kk;
        for(int e = lane; e < total; e += 64, r += dr, c += dc)
        {
            if(c >= kk) { c -= kk; r++; }
            const int src = ((r >> 1) + 32*(r & 1))*SPLB_KT + c;
            Jv[base + e] = lv[src];          // (an ordinary store: the assembly reads these rows next)
            if(c >= cs && c < cs + ns) colidx[base + e] = lc[src];
        }
    }
    else if(mine)
        for(int xy=0;xy<2;xy++)
        {
            const int src = ((lane & 31) + 32*xy)*SPLB_KT;
            for(int c = 0; c < k; c++)
            {
                Jv[dest0 + (long long)xy*k + c] = lv[src + c];
                if(c >= cs && c < cs + ns) colidx[dest0 + (long long)xy*k + c] = lc[src + c];
            }
        }
    __builtin_amdgcn_wave_barrier();
    }   // h
}

// Round 6: the same rows by a lane per ROW (boards of 16 corners and more; board_splined_kernel<true> above stays for
// the smaller ones). With a lane per corner BASELINE configuration 2's 80 000 corners are 1250 waves on 1024 SIMDs, one
// generation, and the kernel's time IS a wave's life: 28.7 us, 46 % of it parked on memory (profiles/r06_config2_
// jacobian_kernel_pmc.txt: 194 vector loads a wave, most of them the pose record's entries, asked for one chain-rule
// term at a time). Here:
//   * a wave takes 64 consecutive ROWS of the board observations' CSR rows (row = 2 corner + coordinate): twice the waves
//     (2500: 2.4 a SIMD), each lane the projection's shared part + ONE surface (16 of the 32 control-point values, one
//     sum, one row's chain rule): project_splined_row(), the same bits as project_splined()'s coordinate
//   * the pose records of the (at most three) observations a wave's rows belong to are staged in LDS by two
//     coalesced loads each; the chain rule reads them as LDS broadcasts
//   * the rows leave through a 32-row LDS tile, half a wave at a time, as one contiguous stream (as above)
// LDS: 32 x 33 doubles of values + 32 x 16 column indices + 3 records = 12.5 KB: twelve waves a CU
#define SPLR_KT   33
#define SPLR_NREC 3
#define SPLR_LDS_BYTES (32*SPLR_KT*8 + 32*16*4 + SPLR_NREC*JOINT_REC*8)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
void board_splined_rows_kernel(DeviceProblem P, OpRef R, const double* __restrict__ joint,
                               int32_t* __restrict__ colidx)
{
    if(opref_skip(R)) return;
    double* __restrict__ x  = opref_get(R).x;
    double* __restrict__ Jv = opref_get(R).Jv;
    const int NPTS = P.W*P.H, RPO = 2*NPTS;                  // rows per observation
    const int lane = threadIdx.x;
    const long long nrows = (long long)P.Nobs_board*RPO;
    const long long g0    = (long long)blockIdx.x*64;
    const bool valid = g0 + lane < nrows;
    const long long gi = valid ? g0 + lane : nrows - 1;      // (the lanes past the end repeat the last row and store nothing)
    const int iobs = (int)(gi / RPO);
    const int r    = (int)(gi - (long long)iobs*RPO);
    const int pt   = r >> 1, xy = r & 1;

    extern __shared__ double lds_spl[];
    double*  __restrict__ lv  = lds_spl;                                   // [32][SPLR_KT]
    int32_t* __restrict__ lc  = (int32_t*)(lds_spl + 32*SPLR_KT);          // [32][16]
    double*  __restrict__ jpl = lds_spl + 32*SPLR_KT + 32*16/2;            // [SPLR_NREC][JOINT_REC]
    // the records of this wave's observations (wave-uniform range)
    const int obs0 = (int)(g0 / RPO);
    const long long glast = (g0 + 63 < nrows) ? g0 + 63 : nrows - 1;
    const int nrec = (int)(glast / RPO) - obs0 + 1;                        // <= SPLR_NREC for boards of >= 16 corners
    for(int q = 0; q < nrec; q++)
    {
        const double* __restrict__ src = joint + (size_t)(obs0 + q)*JOINT_STRIDE;
        jpl[q*JOINT_REC + lane] = src[lane];
        if(lane < JOINT_REC - 64) jpl[q*JOINT_REC + 64 + lane] = src[64 + lane];
    }
    const BoardObsMeta m = P.board_meta[iobs];
    const double* __restrict__ intr = P.unpacked + (size_t)m.icam_intrinsics*P.Nintrinsics;
    const double* __restrict__ wp   = P.unpacked + (size_t)P.Ncameras_intrinsics*P.Nintrinsics;
    const double* __restrict__ obs  = P.board_pool + ((size_t)iobs*NPTS + pt)*3;
    const double qobs = obs[xy], w = obs[2];
    const int  k       = m.nnz_per_row;
    const bool has_ext = P.do_optimize_extrinsics && m.icam_extrinsics >= 0;
    __builtin_amdgcn_wave_barrier();       // (one wave: the LDS is in order; the records are there)
    const double* __restrict__ jp = jpl + (iobs - obs0)*JOINT_REC;

    const int iy = pt / P.W;
    const int ix = pt - iy*P.W;
    const double bx = (double)ix * P.spacing;
    const double by = (double)iy * P.spacing;
    double bz = 0.0, dz_dw[2] = {0.0, 0.0};
    if(P.has_warp_seed)
    {
        const double xr = (double)ix / (double)(P.W - 1);
        const double yr = (double)iy / (double)(P.H - 1);
        dz_dw[0] = 4.0*xr*(1.0 - xr);
        dz_dw[1] = 4.0*yr*(1.0 - yr);
        bz += wp[0]*dz_dw[0];
        bz += wp[1]*dz_dw[1];
    }
    double p[3];
    for(int i=0;i<3;i++)
        p[i] = jp[JOINT_R+3*i+0]*bx + jp[JOINT_R+3*i+1]*by + jp[JOINT_R+3*i+2]*bz + jp[JOINT_T+i];

    double q, dq_dp[3], dq_df, cfx[4], cfy[4];
    int ivar0;
    project_splined_row<true>(xy, &q, dq_dp, &dq_df, &ivar0, cfx, cfy, p, intr, P.cfg);

    const bool inlier = (w >= 0.0);
    if(valid) x[m.i_meas0 + r] = inlier ? (q - qobs)*w : 0.0;

    // the observation's box of control points (as board_splined_kernel: both rows of a corner say the same)
    int* __restrict__ box = opref_get(R).spl_box;
    if(box != NULL && P.Ndist_state)
    {
        const int  n1   = P.cfg.spline_order + 1;
        const int  knot = (ivar0 - 4) >> 1;
        const int  kiy  = knot / P.cfg.spline_Nx, kix = knot - kiy*P.cfg.spline_Nx;
        const bool ok   = valid && inlier;
        unsigned long long todo = __ballot(ok);
        while(todo)
        {
            const int  lead = __ffsll((long long)todo) - 1;
            const int  ob   = __shfl(iobs, lead);
            const bool in   = ok && iobs == ob;
            int x0 = in ? kix : 0x7fffffff, x1 = in ? kix + n1 - 1 : -1;
            int y0 = in ? kiy : 0x7fffffff, y1 = in ? kiy + n1 - 1 : -1;
            for(int off = 32; off > 0; off >>= 1)
            {
                x0 = min(x0, __shfl_xor(x0, off)); x1 = max(x1, __shfl_xor(x1, off));
                y0 = min(y0, __shfl_xor(y0, off)); y1 = max(y1, __shfl_xor(y1, off));
            }
            if(lane == lead)
            {
                atomicMin(&box[4*ob + 0], x0); atomicMax(&box[4*ob + 1], x1);
                atomicMin(&box[4*ob + 2], y0); atomicMax(&box[4*ob + 3], y1);
            }
            todo &= ~__ballot(in);
        }
    }

    // this lane's row, entry by entry, in registers first (both halves compute; a half at a time goes through the tile)
    const long long dest0 = m.i_nnz0 + (long long)r*k;               // this row's first entry
    const int cs = P.Ncore_state ? 2 : 0;                             // the control points' columns: [cs, cs + ns)
    const int n  = P.cfg.spline_order + 1;
    const int ns = P.Ndist_state ? n*n : 0;
    const double ww = inlier ? w : 0.0;                               // outliers: same columns, zero values
    for(int h = 0; h < 2; h++)
    {
        const bool mine = valid && (lane >> 5) == h;
        if(mine)
        {
            double*  __restrict__ row = lv + (lane & 31)*SPLR_KT;
            int32_t* __restrict__ ci  = lc + (lane & 31)*16;
            int c = 0;
            if(P.Ncore_state)
            {
                row[c++] = inlier ? dq_df * w * SCALE_INTRINSICS_FOCAL_LENGTH : 0.0;
                row[c++] = ww * SCALE_INTRINSICS_CENTER_PIXEL;
            }
            if(P.Ndist_state)
            {
                const int col0 = m.i_state_intrinsics + P.Ncore_state + (ivar0 - 4);
                int e = 0;
                for(int jy=0;jy<n;jy++)
                    for(int jx=0;jx<n;jx++)
                    {
                        ci[e++]  = col0 + jy*2*P.cfg.spline_Nx + jx*2 + xy;
                        row[c++] = inlier ? cfx[jx]*cfy[jy]*intr[xy] * w * SCALE_DISTORTION : 0.0;
                    }
            }
            if(has_ext)
            {
                for(int l=0;l<3;l++)
                {
                    double dp[3];
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
                for(int l=0;l<3;l++)
                {
                    double dpr[3], dpt[3];
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
                const double d =
                    dq_dp[0]*jp[JOINT_R + 2] +
                    dq_dp[1]*jp[JOINT_R + 5] +
                    dq_dp[2]*jp[JOINT_R + 8];
                row[c+0] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[0]) : 0.0;
                row[c+1] = inlier ? (w*SCALE_CALOBJECT_WARP)*(d*dz_dw[1]) : 0.0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the half's rows are one extent if they have the same number of entries and follow each other in the CSR arrays
        // (always within an observation; across two unless one camera sits at the reference and the other does not)
        const unsigned long long live = __ballot(mine);
        if(live == 0ull) break;
        const int       lead = 32*h;
        const long long base = __shfl(dest0, lead);
        const int       kk   = __shfl(k, lead);
        const int       nrow = __popcll(live);
        const bool uniform = __all(!mine || (k == kk && dest0 == base + (long long)(lane & 31)*kk));
        if(uniform)
        {
            const int total = nrow*kk;
            const int dr = 64 / kk, dc = 64 - dr*kk;
            int rr = lane / kk, c = lane - rr*kk;
            for(int e = lane; e < total; e += 64, rr += dr, c += dc)
            {
                if(c >= kk) { c -= kk; rr++; }
                Jv[base + e] = lv[rr*SPLR_KT + c];          // (an ordinary store: the assembly reads these rows next)
                if(c >= cs && c < cs + ns) colidx[base + e] = lc[rr*16 + (c - cs)];
            }
        }
        else if(mine)
        {
            const int src = (lane & 31)*SPLR_KT;
            for(int c = 0; c < k; c++)
            {
                Jv[dest0 + c] = lv[src + c];
                if(c >= cs && c < cs + ns) colidx[dest0 + c] = lc[(lane & 31)*16 + (c - cs)];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// discrete points, splined model: one lane per observation (2 rows)
template<bool WITH_J>
__global__ __launch_bounds__(64)
void point_splined_kernel(DeviceProblem P, OpRef R, int32_t* __restrict__ colidx)
{
    if(opref_skip(R)) return;
    const double* __restrict__ b  = opref_get(R).b;
    double*       __restrict__ x  = opref_get(R).x;
    double*       __restrict__ Jv = opref_get(R).Jv;
    const int iobs = blockIdx.x*blockDim.x + threadIdx.x;
    if(iobs >= P.Nobs_point) return;
    const PointObsMeta m = P.point_meta[iobs];
    const int k = m.nnz_per_row;
    const int n = P.cfg.spline_order + 1;
    const double* obs = P.point_pool + (size_t)iobs*3;
    const double  w   = obs[2];
    const bool inlier = !(w <= 0.0);      // <= here, < for boards (mrcal.c:4918 vs :4706)
    const bool at_ref = (m.icam_extrinsics < 0);
    const bool has_ext = P.do_optimize_extrinsics && !at_ref;

    if(!inlier)
    {
        x[m.i_meas0+0] = 0.0;
        x[m.i_meas0+1] = 0.0;
        if(WITH_J)
            for(int xy=0;xy<2;xy++)
            {
                // "it doesn't matter which points I say I depend on": the first
                // (order+1)^2 control points (mrcal.c:4960-4972)
                double*  row = Jv     + m.i_nnz0 + xy*k;
                int32_t* ci  = colidx + m.i_nnz0 + xy*k;
                for(int c=0;c<k;c++) row[c] = 0.0;
                if(P.Ndist_state)
                {
                    const int c0 = P.Ncore_state ? 2 : 0;
                    for(int i=0;i<n*n;i++) ci[c0+i] = m.i_state_intrinsics + P.Ncore_state + i;
                }
            }
        return;
    }

    const double* __restrict__ intr = P.unpacked + (size_t)m.icam_intrinsics*P.Nintrinsics;
    double pref[3];
    if(m.i_state_point >= 0)
        for(int i=0;i<3;i++) pref[i] = b[m.i_state_point + i] * SCALE_POSITION_POINT;
    else
        for(int i=0;i<3;i++) pref[i] = P.seed_points[3*m.i_point + i];

    double p[3], dp_drc[3][3], dp_dpt[3][3];
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
    double q[2], dq_dp[2][3], dq_dfxy[2], cfx[4], cfy[4];
    int ivar0;
    project_splined<WITH_J>(q, dq_dp, dq_dfxy, &ivar0, cfx, cfy, p, intr, P.cfg);
    x[m.i_meas0+0] = (q[0] - obs[0])*w;
    x[m.i_meas0+1] = (q[1] - obs[1])*w;
    if(!WITH_J) return;
    for(int xy=0;xy<2;xy++)
    {
        double*  row = Jv     + m.i_nnz0 + xy*k;
        int32_t* ci  = colidx + m.i_nnz0 + xy*k;
        int c = 0;
        if(P.Ncore_state)
        {
            row[c++] = dq_dfxy[xy] * w * SCALE_INTRINSICS_FOCAL_LENGTH;
            row[c++] = w * SCALE_INTRINSICS_CENTER_PIXEL;
        }
        if(P.Ndist_state)
        {
            const int col0 = m.i_state_intrinsics + P.Ncore_state + (ivar0 - 4);
            for(int jy=0;jy<n;jy++)
                for(int jx=0;jx<n;jx++)
                {
                    ci[c]    = col0 + jy*2*P.cfg.spline_Nx + jx*2 + xy;
                    row[c++] = cfx[jx]*cfy[jy]*intr[xy] * w * SCALE_DISTORTION;
                }
        }
        if(has_ext)
        {
            for(int l=0;l<3;l++)
            {
                const double g = dq_dp[xy][0]*dp_drc[0][l] + dq_dp[xy][1]*dp_drc[1][l] + dq_dp[xy][2]*dp_drc[2][l];
                row[c+l]   = g * w * SCALE_ROTATION_CAMERA;
                row[c+3+l] = dq_dp[xy][l] * w * SCALE_TRANSLATION_CAMERA;
            }
            c += 6;
        }
        if(m.i_state_point >= 0)
            for(int l=0;l<3;l++)
            {
                const double g = dq_dp[xy][0]*dp_dpt[0][l] + dq_dp[xy][1]*dp_dpt[1][l] + dq_dp[xy][2]*dp_dpt[2][l];
                row[c+l] = g * w * SCALE_POSITION_POINT;
            }
    }
}

// Regularization rows of a splined model, in order: per camera and knot
// (iy-major) a radial and a tangential row, 2 nonzeros each; then the centre
// pixel rows (1 nonzero), then unity_cam01 (3). Reference: mrcal.c:5717-5785,
// 5854-5954. One lane per row
template<bool WITH_J, bool WITH_STRUCTURE>
__global__ __launch_bounds__(64)
void regularization_splined_kernel(DeviceProblem P, OpRef R,
                                   int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx)
{
    if(opref_skip(R)) return;
    const double* __restrict__ b  = opref_get(R).b;
    double*       __restrict__ x  = opref_get(R).x;
    double*       __restrict__ Jv = opref_get(R).Jv;
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int Nx = P.cfg.spline_Nx, Ny = P.cfg.spline_Ny;
    const int Nknot_rows   = (P.do_apply_regularization && P.Ndist_state) ? P.Ncameras_intrinsics*Nx*Ny*2 : 0;
    const int Ncenter_rows = (P.do_apply_regularization && P.Ncore_state) ? P.Ncameras_intrinsics*2 : 0;
    const int Nrows        = Nknot_rows + Ncenter_rows + (P.has_unity_cam01 ? 1 : 0);
    const int64_t nnz_knots = (int64_t)2*Nknot_rows;
    if(i == 0 && WITH_STRUCTURE)
        rowptr[P.i_meas_regularization + Nrows] =
            (int32_t)(P.i_nnz_regularization + nnz_knots + Ncenter_rows + (P.has_unity_cam01 ? 3 : 0));
    if(i >= Nrows) return;
    const double nominal_pixel_error = 0.1;
    const int imeas = P.i_meas_regularization + i;

    if(i < Nknot_rows)
    {
        const int64_t innz = P.i_nnz_regularization + (int64_t)2*i;
        if(WITH_STRUCTURE) rowptr[imeas] = (int32_t)innz;
        const int tangential = i & 1;
        const int iknot = i >> 1;
        const int icam  = iknot / (Nx*Ny);
        const int kk    = iknot - icam*Nx*Ny;
        const int iy = kk / Nx, ix = kk - iy*Nx;
        const double scale = nominal_pixel_error / 10.0;
        // direction from the centre of the knot grid to this knot
        double ux = (double)(2*ix - Nx + 1), uy = (double)(2*iy - Ny + 1);
        bool anisotropic = true;
        if(2*ix == Nx - 1 && 2*iy == Ny - 1) { ux = 1.0; anisotropic = false; }
        else
        {
            const double mag = sqrt(ux*ux + uy*uy);
            ux /= mag; uy /= mag;
        }
        const int ivar = 2*kk;
        const double d0 = get_intrinsic(P, b, icam, P.Ncore + ivar + 0);
        const double d1 = get_intrinsic(P, b, icam, P.Ncore + ivar + 1);
        const int col = P.i_state_intrinsics + icam*P.Nintr_state + P.Ncore_state + ivar;
        if(!tangential)
        {
            x[imeas] = scale*(d0*ux + d1*uy);
            if(WITH_J) { Jv[innz] = scale*ux*SCALE_DISTORTION; Jv[innz+1] = scale*uy*SCALE_DISTORTION; }
        }
        else
        {
            const double se = scale*(anisotropic ? 10. : 1.);
            x[imeas] = se*(d0*uy - d1*ux);
            if(WITH_J) { Jv[innz] = se*uy*SCALE_DISTORTION; Jv[innz+1] = -se*ux*SCALE_DISTORTION; }
        }
        if(WITH_STRUCTURE) { colidx[innz] = col; colidx[innz+1] = col+1; }
        return;
    }
    if(i < Nknot_rows + Ncenter_rows)
    {
        const int ii = i - Nknot_rows;
        const int64_t innz = P.i_nnz_regularization + nnz_knots + ii;
        if(WITH_STRUCTURE) rowptr[imeas] = (int32_t)innz;
        const int icam = ii >> 1, xy = ii & 1;
        const double scale  = nominal_pixel_error / (P.imager_width_cam0 * 0.1);
        const double target = 0.5 * (double)(P.imagersizes[2*icam + xy] - 1);
        x[imeas] = scale * (get_intrinsic(P, b, icam, 2+xy) - target);
        if(WITH_J)         Jv[innz]     = scale * SCALE_INTRINSICS_CENTER_PIXEL;
        if(WITH_STRUCTURE) colidx[innz] = P.i_state_intrinsics + icam*P.Nintr_state + 2 + xy;
        return;
    }
    {
        const int64_t innz = P.i_nnz_regularization + nnz_knots + Ncenter_rows;
        if(WITH_STRUCTURE) rowptr[imeas] = (int32_t)innz;
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

void launch_eval_splined(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                                hipStream_t stream, hipEvent_t ev_j0, hipEvent_t ev_j1, int parts)
{
    if(P.Nobs_board > 0 && (parts & EVAL_PART_PROLOGUE))
    {
        const int nblocks_obs    = prologue_obs_blocks(P.Nobs_board);
        const int nblocks_unpack = (P.Ncameras_intrinsics*P.Nintrinsics + 2 + PRO_T - 1)/PRO_T;
        const int nblocks_zero   = (B.zero_total > 0) ? PROLOGUE_ZERO_BLOCKS(B.zero_total) : 0;
        const int Nreg_rows      = 0;      // the splined regularization has its own kernel
        const int nblocks_reg    = (Nreg_rows + PRO_T - 1)/PRO_T;
        launch_prologue(P, B, nblocks_obs, nblocks_unpack, nblocks_zero, nblocks_reg, with_jacobian, stream);
    }
    if(P.Nobs_board > 0 && (parts & EVAL_PART_BOARD))
    {
        if(ev_j0) hipEventRecord(ev_j0, stream);
        const int n = P.Nobs_board*P.W*P.H;
        // (round 6) a lane per row where a wave's 64 rows belong to at most SPLR_NREC observations
        if(with_jacobian && P.W*P.H >= 16)
            hipLaunchKernelGGL(board_splined_rows_kernel, dim3((int)(((long long)2*n + 63)/64)), dim3(64), SPLR_LDS_BYTES, stream, P, B.R, B.joint, B.Ji);
        else if(with_jacobian)
            hipLaunchKernelGGL((board_splined_kernel<true>),  dim3((n+63)/64), dim3(64), 64*SPLB_KT*(sizeof(double) + sizeof(int32_t)), stream, P, B.R, B.joint, B.Ji);
        else
            hipLaunchKernelGGL((board_splined_kernel<false>), dim3((n+63)/64), dim3(64), 0, stream, P, B.R, B.joint, B.Ji);
        if(ev_j1) hipEventRecord(ev_j1, stream);
    }
    if(!(parts & EVAL_PART_REST)) return;
    if(P.Nobs_point > 0)
    {
        if(P.Nobs_board <= 0)   // the unpacked intrinsics are the prologue kernel's job
        {
            EvalBuffers Bu = B;
            Bu.zero_total = 0; Bu.choose = NULL;
            const int nblocks_unpack = (P.Ncameras_intrinsics*P.Nintrinsics + 2 + PRO_T - 1)/PRO_T;
            launch_prologue(P, Bu, 0, nblocks_unpack, 0, 0, false, stream);
        }
        if(with_jacobian)
            hipLaunchKernelGGL((point_splined_kernel<true>),  dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream, P, B.R, B.Ji);
        else
            hipLaunchKernelGGL((point_splined_kernel<false>), dim3((P.Nobs_point + 63)/64), dim3(64), 0, stream, P, B.R, B.Ji);
    }
    launch_triangulated(P, B, with_jacobian, stream);
    const int Nreg = P.Nmeas - P.i_meas_regularization;
    if(Nreg > 0)
    {
        if(with_jacobian)
            hipLaunchKernelGGL((regularization_splined_kernel<true,false>),  dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.R, (int32_t*)NULL, (int32_t*)NULL);
        else
            hipLaunchKernelGGL((regularization_splined_kernel<false,false>), dim3((Nreg + 63)/64), dim3(64), 0, stream,
                               P, B.R, (int32_t*)NULL, (int32_t*)NULL);
    }
}

void launch_structure_splined_regularization(const DeviceProblem& P, const EvalBuffers& B, int Nreg, hipStream_t stream)
{
    hipLaunchKernelGGL((regularization_splined_kernel<false,true>), dim3((Nreg + 63)/64), dim3(64), 0, stream, P, B.R, B.Jp, B.Ji);
}

} // namespace mrcal_amd
