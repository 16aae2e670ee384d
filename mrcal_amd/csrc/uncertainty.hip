// The cross-reprojection uncertainty's K matrix: drt_ref_refperturbed/db_packed (rrp) or
// drt_cam_camperturbed/db_packed (ccp) from the Jacobian of a solved calibration.
//
// Reference: _mrcal_drt_cross_reprojection__dbpacked(), uncertainty.c:798-1541 (declared mrcal.h:613-669;
// Python name mrcal.drt_cross_reprojection__dbpacked(), mrcal-pywrap.c:2012-2110, 2156-2161); the derivation is the
// comment at uncertainty.c:22-128:
//
//   K_packed = -inv(Jcross^T Jcross) Jcross^T J_packed[frames, points, calobject_warp (, extrinsics)]
//
// where a measurement's row of Jcross is its packed gradient with respect to ONE pose (or point) it depends
// on - the frame's rt_ref_frame, the point, or with ccp the camera's rt_cam_ref - pushed through the
// derivative of that pose with respect to a tiny perturbing transform:
//
//   Jcross_row = j_this^T Dinv T,     T = d compose_rt(rt_tiny, rt_this)/d rt_tiny = [ M         0 ]
//                                                                                    [ -skew(t)  I ]
//   M = d compose_r(r_tiny, r)/d r_tiny at r_tiny = 0 (poseutils.c:1003-1072), Dinv = the unpacking scales.
//
// So with S = sum j_this j_this^T and X = sum j_this j_other^T over the rows that share a pose,
//   Jcross^T J_packed_this  += T^T Dinv S        Jcross^T J_packed_other += T^T Dinv X
//   Jcross^T Jcross         += T^T Dinv S Dinv T
// and for a point p (no rotation of its own): T_p = [ -skew(p)  I ] (3 x 6).
//
// What is heavy is S and X: sums over every row of J (37 M values at the metric's size). They are formed here
// on the GPU from the CSR Jacobian - a wavefront per board observation, a lane per measurement row, the
// products summed across the wave in a fixed order - into one small record per observation. The per-pose 6x6
// algebra, the sequential accumulation over the observations and the final 6x6 Cholesky are O(Nobservations)
// and run on the host in the reference's order (consecutive observations of one pose are summed before T is
// applied, as the reference's row loop does), so that the result is reproducible and within rounding of the
// reference's.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "layout.hpp"
#include "host_state.hpp"
#include "problem_object.hpp"
#include "../../include/mrcal_amd.h"

using namespace mrcal_amd;

#define HIP_TRY(expr, onfail)                                           \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if(_e != hipSuccess)                                            \
        {                                                               \
            set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            onfail;                                                     \
        }                                                               \
    } while(0)

namespace {

// where the blocks of the state are, and which camera is asked for
struct DrtDims
{
    int i_intr0, N_intr, N_intr_percam, N_intr_row;
    int i_ext0,  N_ext;
    int i_frm0,  N_frm;
    int i_pt0,   N_pt;
    int i_cw0,   N_cw;
    int icam;          // >= 0: ccp for this camera; < 0: rrp
};

enum { DRT_NONE = 0, DRT_E, DRT_F, DRT_P, DRT_SKIP };
enum { DRT_BAD_STRUCTURE = 1, DRT_BAD_FRAME_AND_POINT = 2, DRT_BAD_NO_INTRINSICS = 4, DRT_BAD_MIXED = 8 };

// sums of one observation (a board: 2 W H rows; a point: 2 rows)
//   s[0..21)   S: upper triangle, row-major, of sum j_this j_this^T   (6x6; a point: 3x3 in s[0..6))
//   s[21..57)  X: sum j_this j_other^T, other = the frame (6x6) or the point (6x3, in s[21..39))   [E only]
//   s[57..69)  sum j_this j_warp^T (6x2)                                                           [E, F]
struct DrtRec
{
    int32_t mode;                     // DRT_*
    int32_t i_this;                   // state index of the pose / point the sums are about
    int32_t i_frame, i_point;         // E: the frame or point of these rows (state index, < 0: none)
    int32_t has_cw, icam_here, bad, _pad;
    double  s[70];
};

struct RowMap
{
    int ival_ext, ival_frm, ival_pt, ival_cw, icam_here, bad;
};

// the walk of uncertainty.c:1099-1172 over the columns of one row
__device__ __forceinline__
RowMap map_row(const DrtDims& d, const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, int r)
{
    RowMap m = { -1, -1, -1, -1, -1, 0 };
    int ival = Jp[r];
    const int iend = Jp[r+1];
    do
    {
        if(!(ival < iend)) break;
        int c = Ji[ival];
        if(d.N_intr > 0 && d.i_intr0 <= c && c < d.i_intr0 + d.N_intr)
        {
            m.icam_here = (c - d.i_intr0)/d.N_intr_percam;
            ival += d.N_intr_row;
            if(!(ival < iend)) break;
            c = Ji[ival];
        }
        if(d.N_ext > 0 && d.i_ext0 <= c && c < d.i_ext0 + d.N_ext)
        {
            m.ival_ext = ival; ival += 6;
            if(!(ival < iend)) break;
            c = Ji[ival];
        }
        if(d.N_frm > 0 && d.i_frm0 <= c && c < d.i_frm0 + d.N_frm)
        {
            m.ival_frm = ival; ival += 6;
            if(!(ival < iend)) break;
            c = Ji[ival];
        }
        if(d.N_pt > 0 && d.i_pt0 <= c && c < d.i_pt0 + d.N_pt)
        {
            m.ival_pt = ival; ival += 3;
            if(!(ival < iend)) break;
            c = Ji[ival];
        }
        if(d.N_cw > 0 && d.i_cw0 <= c && c < d.i_cw0 + d.N_cw)
        {
            m.ival_cw = ival; ival += d.N_cw;
        }
    } while(false);
    if(ival != iend)                          m.bad |= DRT_BAD_STRUCTURE;
    if(m.ival_frm >= 0 && m.ival_pt >= 0)     m.bad |= DRT_BAD_FRAME_AND_POINT;
    if(d.icam >= 0 && m.icam_here < 0)        m.bad |= DRT_BAD_NO_INTRINSICS;
    return m;
}

__device__ __forceinline__ int row_mode(const DrtDims& d, const RowMap& m)
{
    if(d.icam >= 0 && m.icam_here != d.icam) return DRT_SKIP;
    if(d.icam >= 0 && m.ival_ext >= 0)       return DRT_E;
    if(m.ival_frm >= 0)                      return DRT_F;
    if(m.ival_pt  >= 0)                      return DRT_P;
    return DRT_NONE;
}

// the sum over the wavefront, every lane ends up with it; the order of the additions is fixed
__device__ __forceinline__ double wave_sum_f64(double v)
{
    for(int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wavefront per board observation: rows [rows_per_obs*obs, +rows_per_obs)
__global__ __launch_bounds__(64)
void drt_board_rows_kernel(DrtDims d, int Nobs, int rows_per_obs,
                           const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, const double* __restrict__ Jx,
                           DrtRec* __restrict__ rec)
{
    const int obs = blockIdx.x, lane = threadIdx.x;
    if(obs >= Nobs) return;
    const int r0 = obs*rows_per_obs;
    // the observation's structure is that of its first row; every row must agree with it
    const RowMap m0   = map_row(d, Jp, Ji, r0);
    const int    mode = row_mode(d, m0);
    const int    p00  = Jp[r0];
    const int    o_ext = m0.ival_ext - p00, o_frm = m0.ival_frm - p00, o_pt = m0.ival_pt - p00, o_cw = m0.ival_cw - p00;
    const int    i_ext = m0.ival_ext >= 0 ? Ji[m0.ival_ext] : -1;
    const int    i_frm = m0.ival_frm >= 0 ? Ji[m0.ival_frm] : -1;
    const int    i_pt  = m0.ival_pt  >= 0 ? Ji[m0.ival_pt]  : -1;
    int bad = m0.bad;

    double acc[69];
#pragma unroll
    for(int i = 0; i < 69; i++) acc[i] = 0.0;

    for(int rr = lane; rr < rows_per_obs; rr += 64)
    {
        const int r = r0 + rr;
        const RowMap m = map_row(d, Jp, Ji, r);
        const int p0 = Jp[r];
        bad |= m.bad;
        if(m.icam_here != m0.icam_here ||
           (m.ival_ext >= 0 ? m.ival_ext - p0 : -1) != (m0.ival_ext >= 0 ? o_ext : -1) ||
           (m.ival_frm >= 0 ? m.ival_frm - p0 : -1) != (m0.ival_frm >= 0 ? o_frm : -1) ||
           (m.ival_pt  >= 0 ? m.ival_pt  - p0 : -1) != (m0.ival_pt  >= 0 ? o_pt  : -1) ||
           (m.ival_cw  >= 0 ? m.ival_cw  - p0 : -1) != (m0.ival_cw  >= 0 ? o_cw  : -1) ||
           (m.ival_ext >= 0 && Ji[m.ival_ext] != i_ext) ||
           (m.ival_frm >= 0 && Ji[m.ival_frm] != i_frm) ||
           (m.ival_pt  >= 0 && Ji[m.ival_pt]  != i_pt))
            bad |= DRT_BAD_MIXED;
        if(bad || mode == DRT_NONE || mode == DRT_SKIP) continue;

        double jt[6], jo[6], jw[2] = {0.0, 0.0};
        const int nthis = (mode == DRT_P) ? 3 : 6;
        const double* __restrict__ pthis = Jx + p0 + (mode == DRT_E ? o_ext : mode == DRT_F ? o_frm : o_pt);
#pragma unroll
        for(int i = 0; i < 6; i++) jt[i] = (i < nthis) ? pthis[i] : 0.0;
        int nother = 0;
        if(mode == DRT_E)
        {
            if(m0.ival_frm >= 0)     { nother = 6; for(int i = 0; i < 6; i++) jo[i] = Jx[p0 + o_frm + i]; }
            else if(m0.ival_pt >= 0) { nother = 3; for(int i = 0; i < 3; i++) jo[i] = Jx[p0 + o_pt  + i]; }
        }
        if(mode != DRT_P && m0.ival_cw >= 0) { jw[0] = Jx[p0 + o_cw]; jw[1] = Jx[p0 + o_cw + 1]; }

        {
            int k = 0;
#pragma unroll
            for(int i = 0; i < 6; i++)
#pragma unroll
                for(int j = i; j < 6; j++, k++) acc[k] += jt[i]*jt[j];
        }
        if(nother == 6)
        {
#pragma unroll
            for(int i = 0; i < 6; i++)
#pragma unroll
                for(int j = 0; j < 6; j++) acc[21 + 6*i + j] += jt[i]*jo[j];
        }
        else if(nother == 3)
        {
#pragma unroll
            for(int i = 0; i < 6; i++)
#pragma unroll
                for(int j = 0; j < 3; j++) acc[21 + 3*i + j] += jt[i]*jo[j];
        }
#pragma unroll
        for(int i = 0; i < 6; i++) { acc[57 + 2*i] += jt[i]*jw[0]; acc[57 + 2*i + 1] += jt[i]*jw[1]; }
    }
    for(int o = 32; o >= 1; o >>= 1) bad |= __shfl_xor(bad, o);
    DrtRec* __restrict__ out = rec + obs;
    // (a point record keeps its 3x3 in s[0..6): the upper triangle of the 6x6 with zero rows 3..5 is
    //  s[0,1,2, 6,7, 11]; repacked by the host)
#pragma unroll
    for(int i = 0; i < 69; i++)
    {
        const double v = wave_sum_f64(acc[i]);
        if(lane == 0) out->s[i] = v;
    }
    if(lane == 0)
    {
        out->mode = bad ? DRT_NONE : mode;
        out->i_this = (mode == DRT_E) ? i_ext : (mode == DRT_F) ? i_frm : (mode == DRT_P) ? i_pt : -1;
        out->i_frame = i_frm; out->i_point = i_pt;
        out->has_cw = m0.ival_cw >= 0; out->icam_here = m0.icam_here; out->bad = bad; out->_pad = 0;
        out->s[69] = 0.0;
    }
}

// discrete points: one lane per observation (its two rows)
__global__ __launch_bounds__(64)
void drt_point_rows_kernel(DrtDims d, int Nobs, int row0,
                           const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, const double* __restrict__ Jx,
                           DrtRec* __restrict__ rec)
{
    const int obs = blockIdx.x*64 + threadIdx.x;
    if(obs >= Nobs) return;
    DrtRec R;
    memset(&R, 0, sizeof(R));
    const int r0 = row0 + 2*obs;
    const RowMap m0 = map_row(d, Jp, Ji, r0), m1 = map_row(d, Jp, Ji, r0 + 1);
    const int mode = row_mode(d, m0);
    int bad = m0.bad | m1.bad;
    const int pa = Jp[r0], pb = Jp[r0+1];
    if(m0.icam_here != m1.icam_here ||
       (m0.ival_ext >= 0) != (m1.ival_ext >= 0) || (m0.ival_frm >= 0) != (m1.ival_frm >= 0) ||
       (m0.ival_pt  >= 0) != (m1.ival_pt  >= 0) || (m0.ival_cw  >= 0) != (m1.ival_cw  >= 0) ||
       (m0.ival_ext >= 0 && (m0.ival_ext - pa != m1.ival_ext - pb || Ji[m0.ival_ext] != Ji[m1.ival_ext])) ||
       (m0.ival_frm >= 0 && (m0.ival_frm - pa != m1.ival_frm - pb || Ji[m0.ival_frm] != Ji[m1.ival_frm])) ||
       (m0.ival_pt  >= 0 && (m0.ival_pt  - pa != m1.ival_pt  - pb || Ji[m0.ival_pt]  != Ji[m1.ival_pt])))
        bad |= DRT_BAD_MIXED;
    R.i_frame = m0.ival_frm >= 0 ? Ji[m0.ival_frm] : -1;
    R.i_point = m0.ival_pt  >= 0 ? Ji[m0.ival_pt]  : -1;
    R.has_cw = m0.ival_cw >= 0; R.icam_here = m0.icam_here; R.bad = bad;
    R.mode = bad ? DRT_NONE : mode;
    R.i_this = -1;
    if(!bad && (mode == DRT_E || mode == DRT_F || mode == DRT_P))
    {
        const int ithis0 = (mode == DRT_E) ? m0.ival_ext : (mode == DRT_F) ? m0.ival_frm : m0.ival_pt;
        R.i_this = Ji[ithis0];
        const int nthis = (mode == DRT_P) ? 3 : 6;
        for(int row = 0; row < 2; row++)
        {
            const RowMap& m = row ? m1 : m0;
            const double* __restrict__ pthis = Jx + ((mode == DRT_E) ? m.ival_ext : (mode == DRT_F) ? m.ival_frm : m.ival_pt);
            double jt[6];
            for(int i = 0; i < 6; i++) jt[i] = (i < nthis) ? pthis[i] : 0.0;
            int k = 0;
            for(int i = 0; i < 6; i++) for(int j = i; j < 6; j++, k++) R.s[k] += jt[i]*jt[j];
            if(mode == DRT_E && m.ival_frm >= 0)
                for(int i = 0; i < 6; i++) for(int j = 0; j < 6; j++) R.s[21 + 6*i + j] += jt[i]*Jx[m.ival_frm + j];
            else if(mode == DRT_E && m.ival_pt >= 0)
                for(int i = 0; i < 6; i++) for(int j = 0; j < 3; j++) R.s[21 + 3*i + j] += jt[i]*Jx[m.ival_pt + j];
            if(mode != DRT_P && m.ival_cw >= 0)
                for(int i = 0; i < 6; i++) { R.s[57 + 2*i] += jt[i]*Jx[m.ival_cw]; R.s[57 + 2*i + 1] += jt[i]*Jx[m.ival_cw + 1]; }
        }
    }
    rec[obs] = R;
}

// ---------------------------------------------------------------------------------------------------------
// host: the per-pose 6x6 algebra

inline void skew(double K[9], const double t[3])
{
    K[0] = 0;     K[1] = -t[2]; K[2] = t[1];
    K[3] = t[2];  K[4] = 0;     K[5] = -t[0];
    K[6] = -t[1]; K[7] = t[0];  K[8] = 0;
}

// M = d compose_r(r0, r)/d r0 at r0 = 0: mrcal_compose_r_tinyr0_gradientr0(), poseutils.c:1003-1072:
//   M = B/tanB I - (B/tanB - 1)/(4 B^2) r r^T - skew(r)/2,   B = |r|/2;   |r| < 2e-8: I
void compose_r_tinyr0_gradient(double M[9], const double r[3])
{
    const double n2 = r[0]*r[0] + r[1]*r[1] + r[2]*r[2];
    if(n2 < 2e-8*2e-8)
    {
        for(int i = 0; i < 9; i++) M[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double B = sqrt(n2)/2.0, c = B/tan(B);
    double K[9];
    skew(K, r);
    for(int i = 0; i < 3; i++)
        for(int j = 0; j < 3; j++)
            M[3*i + j] = -r[i]*r[j]*(c - 1.0)/(4.0*B*B) + (i == j ? c : 0.0) - K[3*i + j]/2.0;
}

struct KOut
{
    double* K; int stride0;     // (6, N) with this many elements between rows; NULL: not asked for
    int i0;                     // state index of its first column
};

struct Accumulator
{
    KOut e, f, p, cw;
    double JtJ[36];             // Jcross^T Jcross, full symmetric
    const double* b_packed;

    // a pose block: this = an rt (extrinsics or a frame), S its 6x6 sum, X_other (6 x nother) and X_cw (6 x 2) the
    // cross sums; what they are added into
    void flush_rt(const KOut& Kthis, int i_this, const double* Supper, double SCALE_R, double SCALE_T,
                  const KOut* Kother, int i_other, int nother, const double* Xother, const double* Xcw, bool have_cw)
    {
        double S[36];
        for(int i = 0, k = 0; i < 6; i++) for(int j = i; j < 6; j++, k++) S[6*i + j] = S[6*j + i] = Supper[k];
        double r[3], t[3], M[9], Kt[9];
        for(int i = 0; i < 3; i++) { r[i] = b_packed[i_this + i]*SCALE_R; t[i] = b_packed[i_this + 3 + i]*SCALE_T; }
        compose_r_tinyr0_gradient(M, r);
        skew(Kt, t);
        // Tt = [ M^T  skew(t) ; 0  I ]; Dinv = diag(1/SCALE_R x3, 1/SCALE_T x3)
        double Tt[36] = {};
        for(int i = 0; i < 3; i++) for(int j = 0; j < 3; j++) { Tt[6*i + j] = M[3*j + i]; Tt[6*i + 3 + j] = Kt[3*i + j]; }
        for(int i = 0; i < 3; i++) Tt[6*(3+i) + 3 + i] = 1.0;
        double TtD[36];     // T^T Dinv
        for(int i = 0; i < 6; i++) for(int j = 0; j < 6; j++) TtD[6*i + j] = Tt[6*i + j]/(j < 3 ? SCALE_R : SCALE_T);
        auto apply = [&](const double* X, int n, const KOut& out, int i_col)     // out[:, i_col - i0 ...] += T^T Dinv X
        {
            if(out.K == NULL) return;
            for(int i = 0; i < 6; i++)
                for(int j = 0; j < n; j++)
                {
                    double v = 0.0;
                    for(int k = 0; k < 6; k++) v += TtD[6*i + k]*X[n*k + j];
                    out.K[(size_t)out.stride0*i + (i_col - out.i0) + j] += v;
                }
        };
        double P[36];       // T^T Dinv S
        for(int i = 0; i < 6; i++)
            for(int j = 0; j < 6; j++)
            {
                double v = 0.0;
                for(int k = 0; k < 6; k++) v += TtD[6*i + k]*S[6*k + j];
                P[6*i + j] = v;
            }
        if(Kthis.K)
            for(int i = 0; i < 6; i++) for(int j = 0; j < 6; j++)
                Kthis.K[(size_t)Kthis.stride0*i + (i_this - Kthis.i0) + j] += P[6*i + j];
        if(Kother && nother > 0) apply(Xother, nother, *Kother, i_other);
        if(have_cw) apply(Xcw, 2, cw, cw.i0);
        // Jcross^T Jcross += P Dinv T = P (T^T Dinv)^T
        for(int i = 0; i < 6; i++)
            for(int j = i; j < 6; j++)
            {
                double v = 0.0;
                for(int k = 0; k < 6; k++) v += P[6*i + k]*TtD[6*j + k];
                JtJ[6*i + j] += v;
                if(j != i) JtJ[6*j + i] += v;
            }
    }
    // a point: T_p = [ -skew(p)  I ] (3 x 6)
    void flush_point(int i_this, const double* Supper3)
    {
        double S[9];
        for(int i = 0, k = 0; i < 3; i++) for(int j = i; j < 3; j++, k++) S[3*i + j] = S[3*j + i] = Supper3[k];
        double pt[3], Kp[9];
        for(int i = 0; i < 3; i++) pt[i] = b_packed[i_this + i]*SCALE_POSITION_POINT;
        skew(Kp, pt);
        double TtD[18];     // T_p^T / SCALE  (6 x 3): [ skew(p) ; I ]
        for(int i = 0; i < 3; i++) for(int j = 0; j < 3; j++)
        {
            TtD[3*i + j]     = Kp[3*i + j]/SCALE_POSITION_POINT;
            TtD[3*(3+i) + j] = (i == j ? 1.0 : 0.0)/SCALE_POSITION_POINT;
        }
        double P[18];       // T_p^T S / SCALE
        for(int i = 0; i < 6; i++) for(int j = 0; j < 3; j++)
        {
            double v = 0.0;
            for(int k = 0; k < 3; k++) v += TtD[3*i + k]*S[3*k + j];
            P[3*i + j] = v;
        }
        if(p.K)
            for(int i = 0; i < 6; i++) for(int j = 0; j < 3; j++)
                p.K[(size_t)p.stride0*i + (i_this - p.i0) + j] += P[3*i + j];
        for(int i = 0; i < 6; i++)
            for(int j = i; j < 6; j++)
            {
                double v = 0.0;
                for(int k = 0; k < 3; k++) v += P[3*i + k]*TtD[3*j + k];
                JtJ[6*i + j] += v;
                if(j != i) JtJ[6*j + i] += v;
            }
    }
};

bool cholesky6(double L[36], const double A[36])
{
    memset(L, 0, 36*sizeof(double));
    for(int j = 0; j < 6; j++)
    {
        double dd = A[6*j + j];
        for(int k = 0; k < j; k++) dd -= L[6*j + k]*L[6*j + k];
        if(!(dd > 0.0)) return false;
        dd = sqrt(dd);
        L[6*j + j] = dd;
        for(int i = j+1; i < 6; i++)
        {
            double s = A[6*i + j];
            for(int k = 0; k < j; k++) s -= L[6*i + k]*L[6*j + k];
            L[6*i + j] = s/dd;
        }
    }
    return true;
}
void cholesky6_solve(const double L[36], double x[6])
{
    for(int i = 0; i < 6; i++) { double s = x[i]; for(int k = 0; k < i; k++) s -= L[6*i + k]*x[k]; x[i] = s/L[6*i + i]; }
    for(int i = 5; i >= 0; i--) { double s = x[i]; for(int k = i+1; k < 6; k++) s -= L[6*k + i]*x[k]; x[i] = s/L[6*i + i]; }
}

bool init_K(KOut* o, double* K, int stride0_bytes, int stride1_bytes, int N, int i0, const char* what)
{
    o->K = K; o->i0 = i0; o->stride0 = 0;
    if(stride0_bytes <= 0) stride0_bytes = N*(int)sizeof(double);
    if(stride1_bytes <= 0) stride1_bytes = (int)sizeof(double);
    o->stride0 = stride0_bytes/(int)sizeof(double);
    if(K == NULL) return true;
    if(o->stride0*(int)sizeof(double) != stride0_bytes)
    {
        set_error("Currently the implementation assumes that %s_stride0 is a multiple of sizeof(double): got %d", what, stride0_bytes);
        return false;
    }
    if(stride1_bytes != (int)sizeof(double))
    {
        set_error("Currently the implementation assumes that Kpacked has densely-stored rows: %s_stride1 must be sizeof(double). Instead I got %d", what, stride1_bytes);
        return false;
    }
    for(int i = 0; i < 6; i++) memset(&K[(size_t)o->stride0*i], 0, (size_t)N*sizeof(double));
    return true;
}

// J on the device (rowptr, colidx, values of the CSR Jacobian), b_packed on the host
bool drt_cross_reprojection_device(double* Kpackede,  int Ke_s0, int Ke_s1,
                                   double* Kpackedf,  int Kf_s0, int Kf_s1,
                                   double* Kpackedp,  int Kp_s0, int Kp_s1,
                                   double* Kpackedcw, int Kcw_s0, int Kcw_s1,
                                   int icam_intrinsics, const double* b_packed,
                                   const int32_t* d_Jp, const int32_t* d_Ji, const double* d_Jx,
                                   const int32_t* row0_cols, int row0_len,     // host copy of the first row's columns
                                   const Layout& L, hipStream_t stream)
{
    const bool ccp = icam_intrinsics >= 0;
    DrtDims d;
    d.i_intr0 = L.i_state_intrinsics; d.N_intr = L.Nstate_intrinsics;
    d.N_intr_percam = L.dims.Ncameras_intrinsics > 0 ? L.Nstate_intrinsics/L.dims.Ncameras_intrinsics : 0;
    d.i_ext0 = L.i_state_extrinsics; d.N_ext = L.Nstate_extrinsics;
    d.i_frm0 = L.i_state_frames;     d.N_frm = L.Nstate_frames;
    d.i_pt0  = L.i_state_points;     d.N_pt  = L.Nstate_points;
    d.i_cw0  = L.i_state_warp;       d.N_cw  = L.Nstate_warp;
    d.icam   = ccp ? icam_intrinsics : -1;
    // the intrinsics columns a row carries, counted on the first row like uncertainty.c:762-786
    d.N_intr_row = 0;
    while(d.N_intr_row < row0_len && d.N_intr > 0 &&
          d.i_intr0 <= row0_cols[d.N_intr_row] && row0_cols[d.N_intr_row] < d.i_intr0 + d.N_intr)
        d.N_intr_row++;
    if(d.N_intr_percam <= 0) d.N_intr_percam = 1;

    // (uncertainty.c:943-950: so a problem with boards, a warp AND optimized discrete points is refused)
    if(L.i_state_frames >= 0 && L.i_state_warp >= 0 && L.i_state_warp != L.i_state_frames + L.Nstate_frames)
    {
        set_error("I assume that the calobject_warp state variables follow the frame state variables immediately");
        return false;
    }
    if(L.i_state_frames < 0 && L.i_state_extrinsics < 0)
    {
        set_error("Cross-reprojection uncertainty requires either the extrinsics or the frames/points to be optimized. Otherwise the direct method looking at the intrinsics subset of J works fine");
        return false;
    }

    Accumulator A;
    memset(A.JtJ, 0, sizeof(A.JtJ));
    A.b_packed = b_packed;
    if(!init_K(&A.e,  Kpackede,  Ke_s0,  Ke_s1,  L.Nstate_extrinsics, L.i_state_extrinsics, "Kpackede")  ||
       !init_K(&A.f,  Kpackedf,  Kf_s0,  Kf_s1,  L.Nstate_frames,     L.i_state_frames,     "Kpackedf")  ||
       !init_K(&A.p,  Kpackedp,  Kp_s0,  Kp_s1,  L.Nstate_points,     L.i_state_points,     "Kpackedp")  ||
       !init_K(&A.cw, Kpackedcw, Kcw_s0, Kcw_s1, L.Nstate_warp,       L.i_state_warp,       "Kpackedcw"))
        return false;

    const int Nb = L.dims.Nobservations_board > 0 ? L.dims.Nobservations_board : 0, Np = L.dims.Nobservations_point;
    const int rows_per_obs = 2*L.dims.object_width_n*L.dims.object_height_n;
    const int Nrec = Nb + Np;
    std::vector<DrtRec> rec((size_t)(Nrec > 0 ? Nrec : 1));
    if(Nrec > 0)
    {
        DrtRec* d_rec = NULL;
        HIP_TRY(hipMalloc((void**)&d_rec, (size_t)Nrec*sizeof(DrtRec)), return false);
        bool ok = true;
        if(Nb > 0)
        {
            hipLaunchKernelGGL(drt_board_rows_kernel, dim3(Nb), dim3(64), 0, stream, d, Nb, rows_per_obs, d_Jp, d_Ji, d_Jx, d_rec);
            HIP_TRY(hipGetLastError(), ok = false);
        }
        if(ok && Np > 0)
        {
            hipLaunchKernelGGL(drt_point_rows_kernel, dim3((Np + 63)/64), dim3(64), 0, stream, d, Np, L.i_meas_points, d_Jp, d_Ji, d_Jx, d_rec + Nb);
            HIP_TRY(hipGetLastError(), ok = false);
        }
        if(ok) HIP_TRY(hipMemcpyAsync(rec.data(), d_rec, (size_t)Nrec*sizeof(DrtRec), hipMemcpyDeviceToHost, stream), ok = false);
        if(ok) HIP_TRY(hipStreamSynchronize(stream), ok = false);
        (void)hipFree(d_rec);
        if(!ok) return false;
    }

    // the sequential pass of uncertainty.c:1088-1485 at observation granularity: consecutive observations that
    // accumulate into the same pose are summed first
    struct { int mode, i_this, i_frame, i_point, has_cw; double s[69]; } cur;
    cur.mode = DRT_NONE;
    auto flush = [&]()
    {
        if(cur.mode == DRT_E)
        {
            const bool fr = cur.i_frame >= 0, pt = !fr && cur.i_point >= 0;
            A.flush_rt(A.e, cur.i_this, &cur.s[0], SCALE_ROTATION_CAMERA, SCALE_TRANSLATION_CAMERA,
                       fr ? &A.f : (pt ? &A.p : NULL), fr ? cur.i_frame : cur.i_point, fr ? 6 : (pt ? 3 : 0), &cur.s[21],
                       &cur.s[57], cur.has_cw != 0);
        }
        else if(cur.mode == DRT_F)
            A.flush_rt(A.f, cur.i_this, &cur.s[0], SCALE_ROTATION_FRAME, SCALE_TRANSLATION_FRAME,
                       NULL, -1, 0, NULL, &cur.s[57], cur.has_cw != 0);
        else if(cur.mode == DRT_P)
        {
            // the 3x3 upper triangle out of the 6x6 one (rows/columns 3..5 are zero): entries (0,0..2),(1,1..2),(2,2)
            const double S3[6] = { cur.s[0], cur.s[1], cur.s[2], cur.s[6], cur.s[7], cur.s[11] };
            A.flush_point(cur.i_this, S3);
        }
        cur.mode = DRT_NONE;
    };
    for(int i = 0; i < Nrec; i++)
    {
        const DrtRec& R = rec[i];
        const int imeas = (i < Nb) ? i*rows_per_obs : L.i_meas_points + 2*(i - Nb);
        if(R.bad & DRT_BAD_FRAME_AND_POINT)
        {
            set_error("ERROR: both points and frames exist in this measuremnet. This is not supported");
            return false;
        }
        if(R.bad & (DRT_BAD_STRUCTURE | DRT_BAD_MIXED))
        {
            set_error("ERROR: unexpected jacobian structure (measurement %d)", imeas);
            return false;
        }
        if(!R.has_cw && Kpackedcw != NULL)
        {
            set_error("Unexpected jacobian structure. There's no calobject_warp gradient in measurement %d, but the user asked for it", imeas);
            return false;
        }
        if(R.has_cw && Kpackedcw == NULL)
        {
            set_error("Unexpected jacobian structure. There's a calobject_warp gradient in measurement %d, but the user didn't ask for it", imeas);
            return false;
        }
        if(ccp)
        {
            if(R.icam_here < 0)
            {
                set_error("ERROR: I was asked to report the uncertainty for a given icam_intrinsics, but saw a measurement with unknown icam_intrinsics. The intrinsics are probably fixed, and this implementation can't handle that. Please fix it");
                return false;
            }
            if(R.icam_here != icam_intrinsics) continue;      // another camera's measurement
        }
        if(R.mode == DRT_E && Kpackede == NULL) { set_error("Kpackede is needed: the measurements of camera %d carry extrinsics", icam_intrinsics); return false; }
        if(R.mode == DRT_F && Kpackedf == NULL) { set_error("Kpackedf is needed: measurement %d carries a frame gradient", imeas); return false; }
        if((R.mode == DRT_P || (R.mode == DRT_E && R.i_frame < 0 && R.i_point >= 0)) && Kpackedp == NULL)
        { set_error("Kpackedp is needed: measurement %d carries a point gradient", imeas); return false; }
        if(R.mode == DRT_E && R.i_frame >= 0 && Kpackedf == NULL) { set_error("Kpackedf is needed: measurement %d carries a frame gradient", imeas); return false; }

        const bool same = R.mode == cur.mode && R.mode != DRT_NONE && R.i_this == cur.i_this &&
            (R.mode != DRT_E || (R.i_frame == cur.i_frame && R.i_point == cur.i_point));
        if(!same)
        {
            flush();
            cur.mode = R.mode; cur.i_this = R.i_this; cur.i_frame = R.i_frame; cur.i_point = R.i_point; cur.has_cw = R.has_cw;
            memset(cur.s, 0, sizeof(cur.s));
        }
        if(R.mode != DRT_NONE)
            for(int k = 0; k < 69; k++) cur.s[k] += R.s[k];
    }
    flush();

    double Lc[36];
    if(!cholesky6(Lc, A.JtJ))
    {
        set_error("Singular Jcross_t Jcross!");
        return false;
    }
    auto finalize = [&](const KOut& o, int N)
    {
        if(o.K == NULL) return;
        for(int j = 0; j < N; j++)
        {
            double x[6];
            for(int i = 0; i < 6; i++) x[i] = o.K[(size_t)o.stride0*i + j];
            cholesky6_solve(Lc, x);
            for(int i = 0; i < 6; i++) o.K[(size_t)o.stride0*i + j] = -x[i];
        }
    };
    finalize(A.e,  L.Nstate_extrinsics);
    finalize(A.f,  L.Nstate_frames);
    finalize(A.p,  L.Nstate_points);
    finalize(A.cw, L.Nstate_warp);
    return true;
}

} // namespace

extern "C" {

// reference: mrcal.h:613-669, uncertainty.c:798-1541. Host pointers; Jt as mrcal_optimizer_callback() fills it
bool _mrcal_drt_cross_reprojection__dbpacked(double* Kpackede,  int Kpackede_stride0,  int Kpackede_stride1,
                                             double* Kpackedf,  int Kpackedf_stride0,  int Kpackedf_stride1,
                                             double* Kpackedp,  int Kpackedp_stride0,  int Kpackedp_stride1,
                                             double* Kpackedcw, int Kpackedcw_stride0, int Kpackedcw_stride1,
                                             const int icam_intrinsics,
                                             const double* b_packed, int buffer_size_b_packed,
                                             struct cholmod_sparse_struct* Jt,
                                             int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                                             int Npoints, int Npoints_fixed,
                                             int Nobservations_board, int Nobservations_point,
                                             const mrcal_lensmodel_t* lensmodel,
                                             mrcal_problem_selections_t problem_selections,
                                             int calibration_object_width_n, int calibration_object_height_n)
{
    last_error_string().clear();
    if(mrcal_amd_device_count() <= 0) { set_error("no HIP device is visible: libmrcal_amd has no CPU fallback"); return false; }
    Dims dm;
    memset(&dm, 0, sizeof(dm));
    dm.Ncameras_intrinsics = Ncameras_intrinsics; dm.Ncameras_extrinsics = Ncameras_extrinsics; dm.Nframes = Nframes;
    dm.Npoints = Npoints; dm.Npoints_fixed = Npoints_fixed;
    dm.Nobservations_board = Nobservations_board; dm.Nobservations_point = Nobservations_point;
    dm.object_width_n = calibration_object_width_n; dm.object_height_n = calibration_object_height_n;
    const Layout L = make_layout(dm, effective_selections(problem_selections, *lensmodel, Nobservations_board), *lensmodel, NULL, 0);
    if(L.i_state_warp >= 0 && L.Nstate_warp != 2)
    {
        set_error("I assume that the calobject_warp has exactly 2 state variables");
        return false;
    }
    if(buffer_size_b_packed != L.Nstate*(int)sizeof(double))
    {
        set_error("The buffer b_packed has the wrong size. Needed exactly %d bytes, but got %d bytes", L.Nstate*(int)sizeof(double), buffer_size_b_packed);
        return false;
    }
    if(Jt == NULL || L.Nstate != (int)Jt->nrow)
    {
        set_error("Inconsistent inputs. I have Nstate=%d, but Jt->nrow=%d. Giving up", L.Nstate, Jt ? (int)Jt->nrow : -1);
        return false;
    }
    const int Nmeas_obs = L.Nmeas_boards + L.Nmeas_points;
    if((int)Jt->ncol < Nmeas_obs)
    {
        set_error("Inconsistent inputs. The observations take %d measurements, but Jt->ncol=%d", Nmeas_obs, (int)Jt->ncol);
        return false;
    }
    // The kernels walk the caller's arrays (map_row: Ji[Jp[r] .. Jp[r+1])): 32-bit indices, real doubles, rowptr
    // non-decreasing from 0 and inside nzmax, every column inside the state (as mrcal_amd_csr_Jt_x() checks its CSR)
    if(Jt->itype != 0 /* CHOLMOD_INT */ || Jt->xtype != 1 /* CHOLMOD_REAL */ || Jt->dtype != 0 /* CHOLMOD_DOUBLE */)
    {
        set_error("Jt must hold 32-bit indices and real doubles (itype %d, xtype %d, dtype %d)", Jt->itype, Jt->xtype, Jt->dtype);
        return false;
    }
    const int32_t* Jp = (const int32_t*)Jt->p;
    const int32_t* Ji = (const int32_t*)Jt->i;
    const double*  Jx = (const double*) Jt->x;
    if(Jp == NULL || Ji == NULL || Jx == NULL || Jp[0] != 0) { set_error("malformed Jt: its column pointers must start at 0"); return false; }
    for(int r = 0; r < Nmeas_obs; r++)
        if(Jp[r+1] < Jp[r]) { set_error("malformed Jt: the column pointers decrease at measurement %d", r); return false; }
    const int64_t nnz = Nmeas_obs > 0 ? Jp[Nmeas_obs] : 0;
    if(nnz > (int64_t)Jt->nzmax) { set_error("malformed Jt: %lld entries in the observations' rows, nzmax %lld", (long long)nnz, (long long)Jt->nzmax); return false; }
    {
        unsigned bad = 0;
        for(int64_t p = 0; p < nnz; p++) bad |= (unsigned)((unsigned)Ji[p] >= (unsigned)L.Nstate);
        if(bad) { set_error("malformed Jt: a state index is outside [0,%d)", L.Nstate); return false; }
    }
    int32_t *d_Jp = NULL, *d_Ji = NULL; double* d_Jx = NULL;
    bool ok = true;
    HIP_TRY(hipMalloc((void**)&d_Jp, (size_t)(Nmeas_obs + 1)*sizeof(int32_t)), return false);
    HIP_TRY(hipMalloc((void**)&d_Ji, (size_t)(nnz > 0 ? nnz : 1)*sizeof(int32_t)), ok = false);
    if(ok) HIP_TRY(hipMalloc((void**)&d_Jx, (size_t)(nnz > 0 ? nnz : 1)*sizeof(double)), ok = false);
    if(ok) HIP_TRY(hipMemcpy(d_Jp, Jp, (size_t)(Nmeas_obs + 1)*sizeof(int32_t), hipMemcpyHostToDevice), ok = false);
    if(ok && nnz) HIP_TRY(hipMemcpy(d_Ji, Ji, (size_t)nnz*sizeof(int32_t), hipMemcpyHostToDevice), ok = false);
    if(ok && nnz) HIP_TRY(hipMemcpy(d_Jx, Jx, (size_t)nnz*sizeof(double), hipMemcpyHostToDevice), ok = false);
    if(ok)
        ok = drt_cross_reprojection_device(Kpackede, Kpackede_stride0, Kpackede_stride1, Kpackedf, Kpackedf_stride0, Kpackedf_stride1,
                                           Kpackedp, Kpackedp_stride0, Kpackedp_stride1, Kpackedcw, Kpackedcw_stride0, Kpackedcw_stride1,
                                           icam_intrinsics, b_packed, d_Jp, d_Ji, d_Jx,
                                           Ji, Nmeas_obs > 0 ? (Jp[1] - Jp[0]) : 0, L, NULL);
    (void)hipFree(d_Jp); (void)hipFree(d_Ji); (void)hipFree(d_Jx);
    return ok;
}

// The resident tier: K (6, Nstate) row-major, zero outside the blocks the reference's wrapper fills
// (mrcal-pywrap.c:2045-2110), from the Jacobian the problem holds at its current operating point (evaluated here)
bool mrcal_amd_problem_drt_cross_reprojection(mrcal_amd_problem_t* P, int icam_intrinsics, double* K)
{
    last_error_string().clear();
    if((int)P->board_sel.size() != P->L.dims.Nobservations_board)
    {
        set_error("drt_cross_reprojection: this problem is a shard (it holds a part of the rows)");
        return false;
    }
    if(icam_intrinsics >= P->L.dims.Ncameras_intrinsics)
    {
        set_error("icam_intrinsics MUST be <0 (if unused) or in [0,Ncameras_intrinsics-1]. got %d NOT in [0,%d]", icam_intrinsics, P->L.dims.Ncameras_intrinsics-1);
        return false;
    }
    const Layout& L = P->L;
    const int Nstate = L.Nstate;
    if(!mrcal_amd_problem_evaluate(P, true, true)) return false;
    std::vector<double> b((size_t)(Nstate > 0 ? Nstate : 1));
    if(!mrcal_amd_problem_get_b_packed(P, b.data())) return false;
    memset(K, 0, (size_t)6*Nstate*sizeof(double));
    int32_t p01[2] = {0, 0};
    std::vector<int32_t> row0;
    if(L.Nmeas_boards + L.Nmeas_points > 0)
    {
        HIP_TRY(hipMemcpy(p01, P->d_Jp, 2*sizeof(int32_t), hipMemcpyDeviceToHost), return false);
        row0.resize((size_t)(p01[1] - p01[0] > 0 ? p01[1] - p01[0] : 1));
        if(p01[1] > p01[0]) HIP_TRY(hipMemcpy(row0.data(), P->d_Ji + p01[0], (size_t)(p01[1] - p01[0])*sizeof(int32_t), hipMemcpyDeviceToHost), return false);
    }
    const int s0 = Nstate*(int)sizeof(double), s1 = (int)sizeof(double);
    return drt_cross_reprojection_device(L.i_state_extrinsics >= 0 ? K + L.i_state_extrinsics : NULL, s0, s1,
                                         L.i_state_frames     >= 0 ? K + L.i_state_frames     : NULL, s0, s1,
                                         L.i_state_points     >= 0 ? K + L.i_state_points     : NULL, s0, s1,
                                         L.i_state_warp       >= 0 ? K + L.i_state_warp       : NULL, s0, s1,
                                         icam_intrinsics, b.data(), P->d_Jp, P->d_Ji, P->op[P->icur].Jv,
                                         row0.data(), p01[1] - p01[0], L, P->stream);
}

} // extern "C"
