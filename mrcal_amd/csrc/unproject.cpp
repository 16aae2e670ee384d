// mrcal_unproject(): pixel -> observation vector. HOST code.
//
// Reference: mrcal.c:3082-3286. This is a setup step of the solve, not part of
// its hot loop: the reference's wrapper calls it once per triangulated-point
// observation while marshalling the inputs (mrcal-pywrap.c:1388-1395), and so
// does ours. It stays on the host like in the reference, on top of the same
// lens-model source the kernels compile (lens_models.hpp is __host__
// __device__).
//
// Like the reference, the models without a closed-form inverse are inverted
// iteratively in the 2 stereographic coordinates u of the observation vector
// (the model's own fx,fy,cx,cy define the stereographic mapping), starting
// from the pinhole unprojection; the reference runs libdogleg's dense solver on
// this 2x2 problem, here it is a plain Newton iteration on the same residual
// q(u) - q with the same acceptance test (|q(u)-q|^2/2 <= 1e-4, else NaN).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <thread>
#include <vector>
#include <algorithm>
#include "layout.hpp"
#include "kernels.hpp"
#include "host_state.hpp"
#include "lens_models.hpp"
#include "../../include/mrcal_amd.h"

using namespace mrcal_amd;

namespace {

template<int PROJ, int NDIST>
bool project_grad(double* q, double (*dq_dp)[3], const double* v, const double* intr, const LensConfig& cfg)
{
    double gk[2][NDIST > 0 ? NDIST : 1];
    return project_lens<PROJ,NDIST,true>(q, dq_dp, gk, v, intr, cfg);
}
bool project_any(const mrcal_lensmodel_t& m, const LensConfig& cfg,
                 double* q, double (*dq_dp)[3], const double* v, const double* intr)
{
    switch(m.type)
    {
    case MRCAL_LENSMODEL_OPENCV4:  return project_grad<PROJ_OPENCV,4 >(q, dq_dp, v, intr, cfg);
    case MRCAL_LENSMODEL_OPENCV5:  return project_grad<PROJ_OPENCV,5 >(q, dq_dp, v, intr, cfg);
    case MRCAL_LENSMODEL_OPENCV8:  return project_grad<PROJ_OPENCV,8 >(q, dq_dp, v, intr, cfg);
    case MRCAL_LENSMODEL_OPENCV12: return project_grad<PROJ_OPENCV,12>(q, dq_dp, v, intr, cfg);
    case MRCAL_LENSMODEL_CAHVOR:   return project_grad<PROJ_CAHVOR,5 >(q, dq_dp, v, intr, cfg);
    case MRCAL_LENSMODEL_CAHVORE:  return project_grad<PROJ_CAHVORE,8>(q, dq_dp, v, intr, cfg);
    case MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC:
    {
        double dfxy[2], cx[4], cy[4]; int ivar0;
        project_splined<true>(q, dq_dp, dfxy, &ivar0, cx, cy, v, intr, cfg);
        return true;
    }
    default: return false;
    }
}
bool projects_behind_camera(mrcal_lensmodel_type_t t)
{
    // mrcal_lensmodel_metadata().can_project_behind_camera, mrcal.c:255-288 (CAHVORE: false; its solutions with
    // z < 0 are flipped like the other models', mrcal.c:3274)
    return t == MRCAL_LENSMODEL_STEREOGRAPHIC || t == MRCAL_LENSMODEL_LONLAT ||
           t == MRCAL_LENSMODEL_LATLON        || t == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC;
}

} // namespace

extern "C"
bool mrcal_unproject(mrcal_point3_t* out, const mrcal_point2_t* q, int N,
                     const mrcal_lensmodel_t* lensmodel, const double* intrinsics)
{
    last_error_string().clear();
    const double fx = intrinsics[0], fy = intrinsics[1], cx = intrinsics[2], cy = intrinsics[3];
    const mrcal_lensmodel_type_t t = lensmodel->type;

    if(t == MRCAL_LENSMODEL_PINHOLE || t == MRCAL_LENSMODEL_STEREOGRAPHIC ||
       t == MRCAL_LENSMODEL_LONLAT  || t == MRCAL_LENSMODEL_LATLON)
    {
        for(int i=0;i<N;i++)
        {
            const double ux = (q[i].x - cx)/fx, uy = (q[i].y - cy)/fy;
            if(t == MRCAL_LENSMODEL_PINHOLE)            { out[i].x = ux; out[i].y = uy; out[i].z = 1.0; }
            else if(t == MRCAL_LENSMODEL_STEREOGRAPHIC) { out[i].x = ux; out[i].y = uy; out[i].z = 1.0 - (ux*ux + uy*uy)/4.0; }
            else if(t == MRCAL_LENSMODEL_LONLAT)
            {
                // q = (lon, lat) f + c
                out[i].x = cos(uy)*sin(ux); out[i].y = sin(uy); out[i].z = cos(uy)*cos(ux);
            }
            else
            {
                // q = (lat, lon) f + c
                out[i].x = sin(ux); out[i].y = cos(ux)*sin(uy); out[i].z = cos(ux)*cos(uy);
            }
        }
        return true;
    }

    LensConfig cfg; memset(&cfg, 0, sizeof(cfg));
    if(t == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        cfg.spline_order = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.order;
        cfg.spline_Nx    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx;
        cfg.spline_Ny    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny;
        cfg.spline_segments_per_u =
            spline_segments_per_u(cfg.spline_order, cfg.spline_Nx,
                                  (double)lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.fov_x_deg);
    }
    else if(t == MRCAL_LENSMODEL_CAHVORE)
    {
        cfg.cahvore_linearity = lensmodel->LENSMODEL_CAHVORE__config.linearity;
        for(int i=9;i<12;i++)
            if(intrinsics[i] != 0.)
            {
                set_error("unproject() currently only works with a central projection. So I cannot unproject(CAHVORE,E!=0). Please set E=0 to centralize this model");
                return false;
            }
    }
    else if(!(lensmodel_is_opencv(t) || t == MRCAL_LENSMODEL_CAHVOR))
    {
        set_error("mrcal_unproject(): unknown lens model %d", (int)t);
        return false;
    }

    // (round 6) the points are independent: from 2048 of them on (the 57 000 observations of BASELINE configuration 4 are
    // 10 ms of one core - as long as that problem's whole solve on the device) a range each for up to 16 threads. The same
    // code per point: the same bits
    auto points = [&](int ipt0, int ipt1) {
    for(int ipt=ipt0; ipt<ipt1; ipt++)
    {
        // seed: the pinhole unprojection, in stereographic pixel coordinates
        double u[2];
        {
            const double p[3] = { (q[ipt].x - cx)/fx, (q[ipt].y - cy)/fy, 1.0 };
            const double s = 2.0/(sqrt(p[0]*p[0] + p[1]*p[1] + 1.0) + 1.0);
            u[0] = p[0]*s*fx + cx;
            u[1] = p[1]*s*fy + cy;
        }
        double norm2x = 1e300;
        for(int it=0; it<100; it++)
        {
            // v(u): stereographic unprojection, dv/du
            const double sx = (u[0] - cx)/fx, sy = (u[1] - cy)/fy;
            const double v[3] = { sx, sy, 1.0 - (sx*sx + sy*sy)/4.0 };
            const double dv_du[3][2] = { {1.0/fx, 0.0}, {0.0, 1.0/fy}, {-sx/2.0/fx, -sy/2.0/fy} };
            double qh[2], dq_dv[2][3];
            if(!project_any(*lensmodel, cfg, qh, dq_dv, v, intrinsics)) { norm2x = 1e300; break; }
            const double x0 = qh[0] - q[ipt].x, x1 = qh[1] - q[ipt].y;
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
            out[ipt].x = out[ipt].y = nan("");
            out[ipt].z = 0.0;
            continue;
        }
        const double sx = (u[0] - cx)/fx, sy = (u[1] - cy)/fy;
        out[ipt].x = sx; out[ipt].y = sy; out[ipt].z = 1.0 - (sx*sx + sy*sy)/4.0;
        if(!projects_behind_camera(t) && out[ipt].z < 0.0)
        {
            out[ipt].x *= -1.0; out[ipt].y *= -1.0; out[ipt].z *= -1.0;
        }
    }
    };
    int nthreads = 1;
    if(N >= 2048)
    {
        nthreads = (int)std::thread::hardware_concurrency();
        if(nthreads > 16)     nthreads = 16;
        if(nthreads > N/1024) nthreads = N/1024;
        if(nthreads < 1)      nthreads = 1;
    }
    if(nthreads == 1) points(0, N);
    else
    {
        std::vector<std::thread> th;
        const int per = (N + nthreads - 1)/nthreads;
        for(int k = 1; k < nthreads; k++)
            th.emplace_back(points, std::min(N, k*per), std::min(N, (k + 1)*per));
        points(0, std::min(N, per));
        for(std::thread& x : th) x.join();
    }
    return true;
}


// The batch form on the GPU, with the gradients mrcal.unproject(get_gradients=True)
// reports (mrcal/projections.py:112-395): host buffers in and out.
//   v (N,3); dv_dq (N,3,2) and dv_dintrinsics (N,3,Nintrinsics) may be NULL.
// Without gradients the vectors are what mrcal_unproject() returns; with them,
// like the reference's, the stereographic representative of the same direction
// (the gradients are those of THAT vector); normalize: unit vectors and the
// gradients of the unit vectors
extern "C"
bool mrcal_amd_unproject(mrcal_point3_t* v, double* dv_dq, double* dv_dintrinsics,
                         const mrcal_point2_t* q, int N,
                         const mrcal_lensmodel_t* lensmodel, const double* intrinsics, bool normalize)
{
    last_error_string().clear();
    if(mrcal_amd_device_count() <= 0)
    {
        set_error("no HIP device is visible: libmrcal_amd has no CPU fallback");
        return false;
    }
    if(N <= 0) return true;
    if(dv_dintrinsics != NULL && dv_dq == NULL)
    {
        set_error("mrcal_amd_unproject(): dv_dintrinsics needs dv_dq");
        return false;
    }
    const mrcal_lensmodel_type_t t = lensmodel->type;
    if(!lens_supported((int)t)) { set_error("mrcal_amd_unproject(): lens model %d is not supported", (int)t); return false; }
    const int Ni = lensmodel_num_params(*lensmodel);
    LensConfig cfg; memset(&cfg, 0, sizeof(cfg));
    if(t == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        cfg.spline_order = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.order;
        cfg.spline_Nx    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx;
        cfg.spline_Ny    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny;
        cfg.spline_segments_per_u =
            spline_segments_per_u(cfg.spline_order, cfg.spline_Nx,
                                  (double)lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.fov_x_deg);
    }
    else if(t == MRCAL_LENSMODEL_CAHVORE)
    {
        cfg.cahvore_linearity = lensmodel->LENSMODEL_CAHVORE__config.linearity;
        for(int i=9;i<12;i++)
            if(intrinsics[i] != 0.)
            {
                set_error("unproject() currently only works with a central projection. So I cannot unproject(CAHVORE,E!=0). Please set E=0 to centralize this model");
                return false;
            }
    }
    const bool closed = t == MRCAL_LENSMODEL_PINHOLE || t == MRCAL_LENSMODEL_STEREOGRAPHIC ||
                        t == MRCAL_LENSMODEL_LONLAT  || t == MRCAL_LENSMODEL_LATLON;
    const bool grads  = dv_dq != NULL;
    double *d_q = NULL, *d_i = NULL, *d_v = NULL, *d_gq = NULL, *d_gi = NULL, *s_q = NULL, *s_gv = NULL, *s_gi = NULL;
    bool ok = true;
#define TRY(expr) do { if(ok && (expr) != hipSuccess) { set_error("mrcal_amd_unproject(): %s failed", #expr); ok = false; } } while(0)
    TRY(hipMalloc((void**)&d_q, (size_t)2*N*sizeof(double)));
    TRY(hipMalloc((void**)&d_i, (size_t)Ni*sizeof(double)));
    TRY(hipMalloc((void**)&d_v, (size_t)3*N*sizeof(double)));
    if(grads)                  TRY(hipMalloc((void**)&d_gq, (size_t)6*N*sizeof(double)));
    if(dv_dintrinsics != NULL) TRY(hipMalloc((void**)&d_gi, (size_t)3*N*Ni*sizeof(double)));
    if(grads && !closed)
    {
        TRY(hipMalloc((void**)&s_q,  (size_t)2*N*sizeof(double)));
        TRY(hipMalloc((void**)&s_gv, (size_t)6*N*sizeof(double)));
        if(dv_dintrinsics != NULL) TRY(hipMalloc((void**)&s_gi, (size_t)2*N*Ni*sizeof(double)));
    }
    TRY(hipMemcpy(d_q, q, (size_t)2*N*sizeof(double), hipMemcpyHostToDevice));
    TRY(hipMemcpy(d_i, intrinsics, (size_t)Ni*sizeof(double), hipMemcpyHostToDevice));
    TRY(launch_unproject_points((int)t, cfg, N, Ni, d_q, d_i, d_v, d_gq, d_gi, s_q, s_gv, s_gi, normalize, NULL));
    TRY(hipMemcpy(v, d_v, (size_t)3*N*sizeof(double), hipMemcpyDeviceToHost));
    if(grads)                  TRY(hipMemcpy(dv_dq, d_gq, (size_t)6*N*sizeof(double), hipMemcpyDeviceToHost));
    if(dv_dintrinsics != NULL) TRY(hipMemcpy(dv_dintrinsics, d_gi, (size_t)3*N*Ni*sizeof(double), hipMemcpyDeviceToHost));
#undef TRY
    hipFree(d_q); hipFree(d_i); hipFree(d_v); hipFree(d_gq); hipFree(d_gi); hipFree(s_q); hipFree(s_gv); hipFree(s_gi);
    return ok;
}
