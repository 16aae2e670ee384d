// Stand-alone projection and unprojection of N points (mrcal_project() / mrcal_unproject() and their batch forms): the
// solver's own lens-model device functions, a lane per point. (Round 6: cut out of kernels.hip, which holds the
// evaluation kernels of the optimization)
#include <hip/hip_runtime.h>
#include <utility>
#include <stdlib.h>
#include "problem.hpp"
#include "device_math.hpp"
#include "lens_models.hpp"
#include "kernels.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// 6. stand-alone projection of N points (mrcal_project(), mrcal.c:2867-3069)
////////////////////////////////////////////////////////////////////////////////
// One lane per point; the same device functions as the solver's kernels.
// dq_dp (N,2,3) and dq_dintrinsics (N,2,Nintrinsics) may be NULL
template<int PROJ, int NDIST>
__global__ __launch_bounds__(64)
void project_points_kernel(LensConfig cfg, int N, int Nintrinsics,
                           const double* __restrict__ p, const double* __restrict__ intr_in,
                           double* __restrict__ q, double* __restrict__ dq_dp, double* __restrict__ dq_di)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    double intr[4 + NDIST];
#pragma unroll
    for(int k=0;k<4+NDIST;k++) intr[k] = intr_in[k];
    const double pp[3] = { p[3*i], p[3*i+1], p[3*i+2] };
    double qq[2], g[2][3], gk[2][NDIST > 0 ? NDIST : 1];
    project_lens<PROJ,NDIST,true>(qq, g, gk, pp, intr, cfg);
    q[2*i] = qq[0]; q[2*i+1] = qq[1];
    if(dq_dp != NULL)
        for(int xy=0;xy<2;xy++) for(int l=0;l<3;l++) dq_dp[6*i + 3*xy + l] = g[xy][l];
    if(dq_di != NULL)
        for(int xy=0;xy<2;xy++)
        {
            double* __restrict__ row = dq_di + ((size_t)2*i + xy)*Nintrinsics;
            row[xy]     = (qq[xy] - intr[2+xy])/intr[xy];
            row[1-xy]   = 0.0;
            row[2+xy]   = 1.0;
            row[3-xy]   = 0.0;
            for(int k=0;k<NDIST;k++) row[4+k] = gk[xy][k];
        }
}
__global__ __launch_bounds__(64)
void project_points_splined_kernel(LensConfig cfg, int N, int Nintrinsics,
                                   const double* __restrict__ p, const double* __restrict__ intr,
                                   double* __restrict__ q, double* __restrict__ dq_dp, double* __restrict__ dq_di)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    const double pp[3] = { p[3*i], p[3*i+1], p[3*i+2] };
    double qq[2], g[2][3], dfxy[2], cfx[4], cfy[4];
    int ivar0;
    project_splined<true>(qq, g, dfxy, &ivar0, cfx, cfy, pp, intr, cfg);
    q[2*i] = qq[0]; q[2*i+1] = qq[1];
    if(dq_dp != NULL)
        for(int xy=0;xy<2;xy++) for(int l=0;l<3;l++) dq_dp[6*i + 3*xy + l] = g[xy][l];
    if(dq_di != NULL)
    {
        // the caller zeroed dq_di; the sparse gradient lands in its patch
        const int n = cfg.spline_order + 1;
        for(int xy=0;xy<2;xy++)
        {
            double* __restrict__ row = dq_di + ((size_t)2*i + xy)*Nintrinsics;
            row[xy]   = dfxy[xy];
            row[2+xy] = 1.0;
            for(int jy=0;jy<n;jy++)
                for(int jx=0;jx<n;jx++)
                    row[ivar0 + jy*2*cfg.spline_Nx + 2*jx + xy] = cfx[jx]*cfy[jy]*intr[xy];
        }
    }
}

hipError_t launch_project_points(int lens_type, const LensConfig& cfg, int N, int Nintrinsics,
                                 const double* p, const double* intr,
                                 double* q, double* dq_dp, double* dq_di, hipStream_t stream)
{
    if(N <= 0) return hipSuccess;
    const dim3 grid((N + 63)/64), block(64);
#define MRCAL_AMD_PROJECT(PROJ, ND) hipLaunchKernelGGL((project_points_kernel<PROJ,ND>), grid, block, 0, stream, cfg, N, Nintrinsics, p, intr, q, dq_dp, dq_di)
    switch(lens_type)
    {
    case MRCAL_LENSMODEL_PINHOLE:       MRCAL_AMD_PROJECT(PROJ_OPENCV,        0 ); break;
    case MRCAL_LENSMODEL_STEREOGRAPHIC: MRCAL_AMD_PROJECT(PROJ_STEREOGRAPHIC, 0 ); break;
    case MRCAL_LENSMODEL_LONLAT:        MRCAL_AMD_PROJECT(PROJ_LONLAT,        0 ); break;
    case MRCAL_LENSMODEL_LATLON:        MRCAL_AMD_PROJECT(PROJ_LATLON,        0 ); break;
    case MRCAL_LENSMODEL_OPENCV4:       MRCAL_AMD_PROJECT(PROJ_OPENCV,        4 ); break;
    case MRCAL_LENSMODEL_OPENCV5:       MRCAL_AMD_PROJECT(PROJ_OPENCV,        5 ); break;
    case MRCAL_LENSMODEL_OPENCV8:       MRCAL_AMD_PROJECT(PROJ_OPENCV,        8 ); break;
    case MRCAL_LENSMODEL_OPENCV12:      MRCAL_AMD_PROJECT(PROJ_OPENCV,        12); break;
    case MRCAL_LENSMODEL_CAHVOR:        MRCAL_AMD_PROJECT(PROJ_CAHVOR,        5 ); break;
    case MRCAL_LENSMODEL_CAHVORE:       MRCAL_AMD_PROJECT(PROJ_CAHVORE,       8 ); break;
    case MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC:
        hipLaunchKernelGGL(project_points_splined_kernel, grid, block, 0, stream, cfg, N, Nintrinsics, p, intr, q, dq_dp, dq_di);
        break;
    default: return hipErrorInvalidValue;
    }
#undef MRCAL_AMD_PROJECT
    return hipGetLastError();
}

////////////////////////////////////////////////////////////////////////////////
// 6b. stand-alone unprojection, with gradients (mrcal.c:3082-3286; the
//     gradients as mrcal/projections.py:330-395 derives them from project()'s)
////////////////////////////////////////////////////////////////////////////////
// One lane per pixel. The models without a closed-form inverse are inverted in
// the 2 stereographic coordinates u of the observation vector (the model's own
// fx,fy,cx,cy define the mapping), from the pinhole unprojection as the seed:
// Newton on q(v(u)) - q with the reference's acceptance test (|q(u)-q|^2/2 <=
// 1e-4, else NaN). v comes out as (s, 1 - |s|^2/4), s = (u - c)/f.
template<class PROJECT>
__device__ __forceinline__
void unproject_newton(double* __restrict__ vout, const double qx, const double qy, const double* __restrict__ intr,
                      bool behind_camera_ok, PROJECT&& project)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    double u[2];
    {
        const double p0 = (qx - cx)/fx, p1 = (qy - cy)/fy;
        const double sc = 2.0/(sqrt(p0*p0 + p1*p1 + 1.0) + 1.0);
        u[0] = p0*sc*fx + cx;
        u[1] = p1*sc*fy + cy;
    }
    double norm2x = 1e300;
    for(int it=0; it<100; it++)
    {
        const double sx = (u[0] - cx)/fx, sy = (u[1] - cy)/fy;
        const double v[3] = { sx, sy, 1.0 - (sx*sx + sy*sy)/4.0 };
        const double dv_du[3][2] = { {1.0/fx, 0.0}, {0.0, 1.0/fy}, {-sx/2.0/fx, -sy/2.0/fy} };
        double qh[2], dq_dv[2][3];
        project(qh, dq_dv, v);
        const double x0 = qh[0] - qx, x1 = qh[1] - qy;
        norm2x = x0*x0 + x1*x1;
        double J[2][2];
        for(int a=0;a<2;a++) for(int b=0;b<2;b++)
            J[a][b] = dq_dv[a][0]*dv_du[0][b] + dq_dv[a][1]*dv_du[1][b] + dq_dv[a][2]*dv_du[2][b];
        const double det = J[0][0]*J[1][1] - J[0][1]*J[1][0];
        if(!(fabs(det) > 0.0)) break;
        const double du0 = -( J[1][1]*x0 - J[0][1]*x1)/det;
        const double du1 = -(-J[1][0]*x0 + J[0][0]*x1)/det;
        u[0] += du0; u[1] += du1;
        if(du0*du0 + du1*du1 < 1e-24) break;
    }
    if(!(norm2x/2.0 <= 1e-4))
    {
        vout[0] = vout[1] = __longlong_as_double(0x7ff8000000000000ll);
        vout[2] = 0.0;
        return;
    }
    const double sx = (u[0] - cx)/fx, sy = (u[1] - cy)/fy;
    double v[3] = { sx, sy, 1.0 - (sx*sx + sy*sy)/4.0 };
    if(!behind_camera_ok && v[2] < 0.0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
    vout[0] = v[0]; vout[1] = v[1]; vout[2] = v[2];
}
template<int PROJ, int NDIST>
__global__ __launch_bounds__(64)
void unproject_points_kernel(LensConfig cfg, int N, const double* __restrict__ q, const double* __restrict__ intr_in,
                             int behind_camera_ok, double* __restrict__ v)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    double intr[4 + NDIST];
#pragma unroll
    for(int k=0;k<4+NDIST;k++) intr[k] = intr_in[k];
    unproject_newton(&v[3*i], q[2*i], q[2*i+1], intr, behind_camera_ok != 0,
                     [&](double* qh, double (*g)[3], const double* vv)
                     {
                         double gk[2][NDIST > 0 ? NDIST : 1];
                         project_lens<PROJ,NDIST,true>(qh, g, gk, vv, intr, cfg);
                     });
}
__global__ __launch_bounds__(64)
void unproject_points_splined_kernel(LensConfig cfg, int N, const double* __restrict__ q, const double* __restrict__ intr,
                                     double* __restrict__ v)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    unproject_newton(&v[3*i], q[2*i], q[2*i+1], intr, true,
                     [&](double* qh, double (*g)[3], const double* vv)
                     {
                         double dfxy[2], cfx[4], cfy[4]; int ivar0;
                         project_splined<true>(qh, g, dfxy, &ivar0, cfx, cfy, vv, intr, cfg);
                     });
}
// The models with a closed-form inverse, and their gradients as
// mrcal.unproject() reports them (mrcal/projections.py:252-312): dv/dq from the
// formula; dv/df = (c - q)/f dv/dq, dv/dc = -dv/dq
__global__ __launch_bounds__(64)
void unproject_closed_form_kernel(int lens_type, int N, const double* __restrict__ q, const double* __restrict__ intr,
                                  double* __restrict__ v, double* __restrict__ dv_dq, double* __restrict__ dv_di)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double ux = (q[2*i] - cx)/fx, uy = (q[2*i+1] - cy)/fy;
    double vv[3], g[3][2];          // g = dv/du, u = (q - c)/f
    if(lens_type == MRCAL_LENSMODEL_PINHOLE)
    {
        vv[0] = ux; vv[1] = uy; vv[2] = 1.0;
        g[0][0] = 1; g[0][1] = 0; g[1][0] = 0; g[1][1] = 1; g[2][0] = 0; g[2][1] = 0;
    }
    else if(lens_type == MRCAL_LENSMODEL_STEREOGRAPHIC)
    {
        vv[0] = ux; vv[1] = uy; vv[2] = 1.0 - (ux*ux + uy*uy)/4.0;
        g[0][0] = 1; g[0][1] = 0; g[1][0] = 0; g[1][1] = 1; g[2][0] = -ux/2.0; g[2][1] = -uy/2.0;
    }
    else if(lens_type == MRCAL_LENSMODEL_LONLAT)
    {
        // q = (lon, lat) f + c
        const double sl = sin(ux), cl = cos(ux), sa = sin(uy), ca = cos(uy);
        vv[0] = ca*sl; vv[1] = sa; vv[2] = ca*cl;
        g[0][0] = ca*cl;  g[0][1] = -sa*sl;
        g[1][0] = 0.0;    g[1][1] = ca;
        g[2][0] = -ca*sl; g[2][1] = -sa*cl;
    }
    else
    {
        // q = (lat, lon) f + c
        const double sa = sin(ux), ca = cos(ux), sl = sin(uy), cl = cos(uy);
        vv[0] = sa; vv[1] = ca*sl; vv[2] = ca*cl;
        g[0][0] = ca;     g[0][1] = 0.0;
        g[1][0] = -sa*sl; g[1][1] = ca*cl;
        g[2][0] = -sa*cl; g[2][1] = -ca*sl;
    }
    for(int k=0;k<3;k++) v[3*i+k] = vv[k];
    if(dv_dq != NULL)
        for(int k=0;k<3;k++) { dv_dq[6*i + 2*k] = g[k][0]/fx; dv_dq[6*i + 2*k + 1] = g[k][1]/fy; }
    if(dv_di != NULL)
        for(int k=0;k<3;k++)
        {
            double* __restrict__ row = dv_di + ((size_t)3*i + k)*4;
            const double dx = g[k][0]/fx, dy = g[k][1]/fy;
            row[0] = (cx - q[2*i])/fx*dx;   row[1] = (cy - q[2*i+1])/fy*dy;
            row[2] = -dx;                   row[3] = -dy;
        }
}
// v (any length) -> the stereographic representative v = (u, 1 - |u|^2/4) of the same direction; with the
// gradients of project() AT THAT v (dq_dv (2x3), dq_di (2xNi)): dv/dq = dv/du inv(dq/du), dv/di = -dv/dq dq/di
// (q is held constant). Two passes around the launch of project(): pass 0 re-expresses v; pass 1 combines
__global__ __launch_bounds__(64)
void unproject_restereo_kernel(int N, double* __restrict__ v)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    const double a = v[3*i], b = v[3*i+1], c = v[3*i+2];
    const double mag = sqrt(a*a + b*b + c*c);
    const double sc  = 2.0/(mag + c);
    const double u0 = a*sc, u1 = b*sc;
    v[3*i] = u0; v[3*i+1] = u1; v[3*i+2] = 1.0 - (u0*u0 + u1*u1)/4.0;
}
__global__ __launch_bounds__(64)
void unproject_combine_kernel(int N, int Ni, const double* __restrict__ v, const double* __restrict__ dq_dv,
                              const double* __restrict__ dq_di, double* __restrict__ dv_dq, double* __restrict__ dv_di)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    const double u0 = v[3*i], u1 = v[3*i+1];
    const double dv_du[3][2] = { {1.0, 0.0}, {0.0, 1.0}, {-u0/2.0, -u1/2.0} };
    double J[2][2];
    for(int a=0;a<2;a++) for(int b=0;b<2;b++)
        J[a][b] = dq_dv[6*i+3*a]*dv_du[0][b] + dq_dv[6*i+3*a+1]*dv_du[1][b] + dq_dv[6*i+3*a+2]*dv_du[2][b];
    const double det = J[0][0]*J[1][1] - J[0][1]*J[1][0];
    const double Ji[2][2] = { { J[1][1]/det, -J[0][1]/det }, { -J[1][0]/det, J[0][0]/det } };
    double g[3][2];
    for(int k=0;k<3;k++) for(int b=0;b<2;b++) g[k][b] = dv_du[k][0]*Ji[0][b] + dv_du[k][1]*Ji[1][b];
    for(int k=0;k<3;k++) { dv_dq[6*i+2*k] = g[k][0]; dv_dq[6*i+2*k+1] = g[k][1]; }
    if(dv_di != NULL)
        for(int k=0;k<3;k++)
        {
            double* __restrict__ row = dv_di + ((size_t)3*i + k)*Ni;
            const double* __restrict__ r0 = dq_di + ((size_t)2*i)*Ni;
            const double* __restrict__ r1 = r0 + Ni;
            for(int j=0;j<Ni;j++) row[j] = -(g[k][0]*r0[j] + g[k][1]*r1[j]);
        }
}
// vn = v/|v|; dvn = (I - vn vn^T) dv / |v|   (mrcal/projections.py:214-243)
__global__ __launch_bounds__(64)
void unproject_normalize_kernel(int N, int Ni, double* __restrict__ v, double* __restrict__ dv_dq, double* __restrict__ dv_di)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= N) return;
    double vv[3] = { v[3*i], v[3*i+1], v[3*i+2] };
    const bool good = isfinite(vv[0]) && isfinite(vv[1]) && isfinite(vv[2]);
    if(!good) { v[3*i] = v[3*i+1] = v[3*i+2] = 0.0; }      // (the reference: not-finite vectors normalize to 0)
    const double mag = good ? sqrt(vv[0]*vv[0] + vv[1]*vv[1] + vv[2]*vv[2]) : 1.0;
    for(int k=0;k<3;k++) vv[k] = good ? vv[k]/mag : 0.0;
    if(good) for(int k=0;k<3;k++) v[3*i+k] = vv[k];
    auto fix = [&](double* __restrict__ d, int ncol)
    {
        for(int j=0;j<ncol;j++)
        {
            const double a0 = d[j]/mag, a1 = d[ncol + j]/mag, a2 = d[2*ncol + j]/mag;
            const double dt = vv[0]*a0 + vv[1]*a1 + vv[2]*a2;
            d[j] = a0 - vv[0]*dt; d[ncol + j] = a1 - vv[1]*dt; d[2*ncol + j] = a2 - vv[2]*dt;
        }
    };
    if(dv_dq != NULL) fix(dv_dq + (size_t)6*i, 2);
    if(dv_di != NULL) fix(dv_di + (size_t)3*i*Ni, Ni);
}

hipError_t launch_unproject_points(int lens_type, const LensConfig& cfg, int N, int Nintrinsics,
                                   const double* q, const double* intr,
                                   double* v, double* dv_dq, double* dv_di,
                                   double* scratch_q, double* scratch_dq_dv, double* scratch_dq_di,
                                   bool normalize, hipStream_t stream)
{
    if(N <= 0) return hipSuccess;
    const dim3 grid((N + 63)/64), block(64);
    const bool closed = lens_type == MRCAL_LENSMODEL_PINHOLE || lens_type == MRCAL_LENSMODEL_STEREOGRAPHIC ||
                        lens_type == MRCAL_LENSMODEL_LONLAT  || lens_type == MRCAL_LENSMODEL_LATLON;
    if(closed)
        hipLaunchKernelGGL(unproject_closed_form_kernel, grid, block, 0, stream, lens_type, N, q, intr, v, dv_dq, dv_di);
    else
    {
        // can_project_behind_camera (mrcal.c:255-288): none of the parametric models that come this way can: a solution with z < 0 is flipped, mrcal.c:3274
        const int behind_ok = 0;       // (the splined model, which can, has its own kernel below)
#define MRCAL_AMD_UNPROJECT(PROJ, ND) hipLaunchKernelGGL((unproject_points_kernel<PROJ,ND>), grid, block, 0, stream, cfg, N, q, intr, behind_ok, v)
        switch(lens_type)
        {
        case MRCAL_LENSMODEL_OPENCV4:  MRCAL_AMD_UNPROJECT(PROJ_OPENCV,  4 ); break;
        case MRCAL_LENSMODEL_OPENCV5:  MRCAL_AMD_UNPROJECT(PROJ_OPENCV,  5 ); break;
        case MRCAL_LENSMODEL_OPENCV8:  MRCAL_AMD_UNPROJECT(PROJ_OPENCV,  8 ); break;
        case MRCAL_LENSMODEL_OPENCV12: MRCAL_AMD_UNPROJECT(PROJ_OPENCV,  12); break;
        case MRCAL_LENSMODEL_CAHVOR:   MRCAL_AMD_UNPROJECT(PROJ_CAHVOR,  5 ); break;
        case MRCAL_LENSMODEL_CAHVORE:  MRCAL_AMD_UNPROJECT(PROJ_CAHVORE, 8 ); break;
        case MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC:
            hipLaunchKernelGGL(unproject_points_splined_kernel, grid, block, 0, stream, cfg, N, q, intr, v);
            break;
        default: return hipErrorInvalidValue;
        }
#undef MRCAL_AMD_UNPROJECT
        if(dv_dq != NULL)
        {
            // the gradients of project() at the stereographic representative of v
            hipLaunchKernelGGL(unproject_restereo_kernel, grid, block, 0, stream, N, v);
            if(lens_type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && dv_di != NULL)
                (void)hipMemsetAsync(scratch_dq_di, 0, (size_t)2*N*Nintrinsics*sizeof(double), stream);
            const hipError_t e = launch_project_points(lens_type, cfg, N, Nintrinsics, v, intr, scratch_q, scratch_dq_dv,
                                                       dv_di != NULL ? scratch_dq_di : NULL, stream);
            if(e != hipSuccess) return e;
            hipLaunchKernelGGL(unproject_combine_kernel, grid, block, 0, stream, N, Nintrinsics, v, scratch_dq_dv,
                               scratch_dq_di, dv_dq, dv_di);
        }
    }
    if(normalize)
        hipLaunchKernelGGL(unproject_normalize_kernel, grid, block, 0, stream, N, Nintrinsics, v, dv_dq, dv_di);
    return hipGetLastError();
}

} // namespace mrcal_amd
