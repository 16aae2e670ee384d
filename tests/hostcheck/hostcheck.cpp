// TEST-ONLY host build of the __host__ __device__ math in
// mrcal_amd/csrc/lens_models.hpp and device_math.hpp, so that the CPU test
// suite (no GPU in the build container) can check the very source the kernels
// compile against the reference's own mrcal_project() (oracle/_ref).
// This is not a product path: nothing in mrcal_amd/ loads it.
//
//   hipcc -O2 -std=c++17 -fPIC -shared -o libhostcheck.so hostcheck.cpp      (host code only)
#include <string.h>
#include "../../mrcal_amd/csrc/lens_models.hpp"

using namespace mrcal_amd;

template<int PROJ, int NDIST>
static int run(double* q, double* dq_dp, double* dq_dk, const double* p, int N,
               const double* intr, const LensConfig& cfg)
{
    int nfail = 0;
    for(int i=0;i<N;i++)
    {
        double g[2][3], gk[2][NDIST > 0 ? NDIST : 1];
        if(!project_lens<PROJ,NDIST,true>(&q[2*i], g, gk, &p[3*i], intr, cfg)) nfail++;
        memcpy(&dq_dp[6*i], g, sizeof(g));
        for(int xy=0;xy<2;xy++)
            for(int k=0;k<NDIST;k++)
                dq_dk[(2*i + xy)*NDIST + k] = gk[xy][k];
    }
    return nfail;
}

extern "C" {

// model: the PROJ_* value; ndist: number of distortion parameters.
// q[N][2], dq_dp[N][2][3], dq_dk[N][2][ndist]. Returns the number of failed
// projections (CAHVORE), <0 for an unknown model
int hostcheck_project(int model, int ndist, double* q, double* dq_dp, double* dq_dk,
                      const double* p, int N, const double* intr, double cahvore_linearity)
{
    LensConfig cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.cahvore_linearity = cahvore_linearity;
    switch(model)
    {
    case PROJ_OPENCV:
        switch(ndist)
        {
        case 0:  return run<PROJ_OPENCV,0 >(q,dq_dp,dq_dk,p,N,intr,cfg);
        case 4:  return run<PROJ_OPENCV,4 >(q,dq_dp,dq_dk,p,N,intr,cfg);
        case 5:  return run<PROJ_OPENCV,5 >(q,dq_dp,dq_dk,p,N,intr,cfg);
        case 8:  return run<PROJ_OPENCV,8 >(q,dq_dp,dq_dk,p,N,intr,cfg);
        case 12: return run<PROJ_OPENCV,12>(q,dq_dp,dq_dk,p,N,intr,cfg);
        }
        return -1;
    case PROJ_STEREOGRAPHIC: return run<PROJ_STEREOGRAPHIC,0>(q,dq_dp,dq_dk,p,N,intr,cfg);
    case PROJ_LONLAT:        return run<PROJ_LONLAT,0>(q,dq_dp,dq_dk,p,N,intr,cfg);
    case PROJ_LATLON:        return run<PROJ_LATLON,0>(q,dq_dp,dq_dk,p,N,intr,cfg);
    case PROJ_CAHVOR:        return run<PROJ_CAHVOR,5>(q,dq_dp,dq_dk,p,N,intr,cfg);
    case PROJ_CAHVORE:       return run<PROJ_CAHVORE,8>(q,dq_dp,dq_dk,p,N,intr,cfg);
    }
    return -1;
}

// splined stereographic: q[N][2], dq_dp[N][2][3], dq_dfxy[N][2], ivar0[N], coef[N][8] (x then y basis values)
void hostcheck_project_splined(double* q, double* dq_dp, double* dq_dfxy, int* ivar0, double* coef,
                               const double* p, int N, const double* intr,
                               int order, int Nx, int Ny, double fov_x_deg)
{
    LensConfig cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.spline_order = order; cfg.spline_Nx = Nx; cfg.spline_Ny = Ny;
    cfg.spline_segments_per_u = spline_segments_per_u(order, Nx, fov_x_deg);
    for(int i=0;i<N;i++)
    {
        double g[2][3];
        project_splined<true>(&q[2*i], g, &dq_dfxy[2*i], &ivar0[i], &coef[8*i], &coef[8*i+4], &p[3*i], intr, cfg);
        memcpy(&dq_dp[6*i], g, sizeof(g));
    }
}

// the same by ROWS (round 6: board_splined_rows_kernel's lanes take one image coordinate each): project_splined_row() of
// coordinate k into the k-th halves of the same arrays; coef: the x then the y basis values as coordinate 0's call left them,
// coef_k1[N][8]: coordinate 1's (they must be the same)
void hostcheck_project_splined_rows(double* q, double* dq_dp, double* dq_dfxy, int* ivar0, double* coef, double* coef_k1,
                                    const double* p, int N, const double* intr,
                                    int order, int Nx, int Ny, double fov_x_deg)
{
    LensConfig cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.spline_order = order; cfg.spline_Nx = Nx; cfg.spline_Ny = Ny;
    cfg.spline_segments_per_u = spline_segments_per_u(order, Nx, fov_x_deg);
    for(int i=0;i<N;i++)
        for(int k=0;k<2;k++)
        {
            int iv;
            double* c = k ? &coef_k1[8*i] : &coef[8*i];
            project_splined_row<true>(k, &q[2*i + k], &dq_dp[6*i + 3*k], &dq_dfxy[2*i + k], &iv, c, c + 4, &p[3*i], intr, cfg);
            if(k == 0) ivar0[i] = iv; else if(iv != ivar0[i]) ivar0[i] = -1;
        }
}

// rotation composition and friends, for the poseutils known-answer tests
void hostcheck_compose_rt(double* rt_out, double* d_r_r0, double* d_r_r1, double* d_t_r0, double* d_t_t1,
                          const double* rt0, const double* rt1)
{
    Dual<6> r0[3], r1[3], r01[3];
    for(int i=0;i<3;i++) { r0[i] = Dual<6>::variable(rt0[i], i); r1[i] = Dual<6>::variable(rt1[i], 3+i); }
    compose_r_dual<6>(r01, r0, r1);
    Dual<6> t1[3], t01[3];
    for(int i=0;i<3;i++) t1[i] = Dual<6>::variable(rt1[3+i], 3+i);
    rotate_point_r_dual<6>(t01, r0, t1, false);
    for(int i=0;i<3;i++)
    {
        rt_out[i]   = r01[i].x;
        rt_out[3+i] = t01[i].x + rt0[3+i];
        for(int l=0;l<3;l++)
        {
            d_r_r0[3*i+l] = r01[i].d[l];   d_r_r1[3*i+l] = r01[i].d[3+l];
            d_t_r0[3*i+l] = t01[i].d[l];   d_t_t1[3*i+l] = t01[i].d[3+l];
        }
    }
}
void hostcheck_R_from_r(double* R, double* dR, const double* r) { R_from_r_with_grad(R, dR, r); }

} // extern "C"

#include "../../mrcal_amd/csrc/triangulation.hpp"
extern "C" {
// the triangulated-pair residual and its 12 derivatives (rt0 then rt1; a NULL rt
// = that camera is at the reference, its 6 derivatives are returned as 0)
double hostcheck_tri_pair_error(double* derr_drt0, double* derr_drt1, int* convergent,
                                const double* v0, const double* v1, const double* rt0, const double* rt1)
{
    bool conv = true;
    const Dual<12> e = tri_pair_error<12>(v0, v1, rt0, rt1, &conv);
    for(int i=0;i<6;i++) { derr_drt0[i] = e.d[i]; derr_drt1[i] = e.d[6+i]; }
    *convergent = conv ? 1 : 0;
    // the value-only instantiation must agree
    bool conv0 = true;
    const Dual<0> e0 = tri_pair_error<0>(v0, v1, rt0, rt1, &conv0);
    if(e0.x != e.x || conv0 != conv) return -12345.0;
    return e.x;
}
}
