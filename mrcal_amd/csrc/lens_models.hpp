// Lens models: camera-frame point p -> pixel q, with dq/dp (2x3) and
// dq/d(distortion parameters).
//
// Reference behaviour being reproduced (the math and its branch points, not
// the code):
//   OPENCV4/5/8/12, PINHOLE  opencv.c:50-152, mrcal.c:1436-1472
//   STEREOGRAPHIC            mrcal.c:1511-1573
//   LONLAT / LATLON          mrcal.c:1687-1743 / 1819-1858
//   CAHVOR                   mrcal.c:1067-1240
//   CAHVORE                  cahvore.cc:19-149 (Newton solve for theta carried
//                            through with forward-mode duals), mrcal.c:1242-1347
//   SPLINED_STEREOGRAPHIC    mrcal.c:884-1051 (B-spline sampling), 2075-2293
//
// Everything is __host__ __device__ so that the SAME source is unit-tested on
// the CPU against the reference's code (tests/hostcheck/) and runs in the
// kernels. There is no CPU product path: the host build exists only in tests.
#pragma once
#include "device_math.hpp"

namespace mrcal_amd {

enum
{
    PROJ_OPENCV = 0,      // NDIST = 0 (pinhole), 4, 5, 8, 12
    PROJ_STEREOGRAPHIC,
    PROJ_LONLAT,
    PROJ_LATLON,
    PROJ_CAHVOR,          // NDIST = 5
    PROJ_CAHVORE,         // NDIST = 8
    PROJ_SPLINED          // handled by its own entry points below
};

// model configuration that is not in the intrinsics vector
struct LensConfig
{
    double cahvore_linearity;
    int    spline_order, spline_Nx, spline_Ny;
    double spline_segments_per_u;
};

////////////////////////////////////////////////////////////////////////////////
// OpenCV rational + tangential + thin-prism family
////////////////////////////////////////////////////////////////////////////////
// Both image rows at once, for a lane that owns a corner (the board and point
// kernels). Same model, but the gradients go through the four partials of the
// distorted coordinates with respect to (X,Y) = (x/z, y/z) instead of three
// general directional derivatives per row:
//   ud_x = X g + k2 a1 + k3 a2 + k8 r2 + k9 r4       g = num/den,  a1 = 2XY,
//   ud_y = Y g + k3 a1 + k2 a3 + k10 r2 + k11 r4     a2 = r2 + 2X^2,  a3 = r2 + 2Y^2
//   d ud_x/dX = g + 2X (X g' + 3 k3 + k8 + 2 k9 r2) + 2 k2 Y      g' = dg/dr2 = (num' - g den')/den
//   d ud_x/dY =     2Y (X g' +   k3 + k8 + 2 k9 r2) + 2 k2 X
//   d ud_y/dX =     2X (Y g' +   k2 + k10 + 2 k11 r2) + 2 k3 Y
//   d ud_y/dY = g + 2Y (Y g' + 3 k2 + k10 + 2 k11 r2) + 2 k3 X
//   dq/dp = f (d ud/dX, d ud/dY, -(X d ud/dX + Y d ud/dY)) / z
// About half the arithmetic of evaluating the rows one by one with directional
// derivatives (180 -> 103 FP64 instructions at OPENCV8), in a kernel that is
// bound by FP64 issue.
// The k[] slots beyond NDIST are compile-time zeros and fold away.
// NOTE on the coding style: everything is a named scalar, never an element of a
// local array picked by a run-time index. A select between two array elements
// is turned by the compiler into a load from a selected ADDRESS if it gets to
// it before the array has been promoted to registers; the array then lives in
// scratch memory, and every access to it waits for vmcnt(0), i.e. for all the
// outstanding stores of the Jacobian (measured once: +20 us at the benchmark)
template<int NDIST, bool WITH_GRAD>
MRCAL_AMD_HD
void project_opencv_both(double* q, double (*dq_dp)[3], double (*dq_dk)[NDIST > 0 ? NDIST : 1],
                         const double* p, const double* intr)
{
    const double k0  = (NDIST > 0 ) ? intr[4 + 0 ] : 0.0;
    const double k1  = (NDIST > 1 ) ? intr[4 + 1 ] : 0.0;
    const double k2  = (NDIST > 2 ) ? intr[4 + 2 ] : 0.0;
    const double k3  = (NDIST > 3 ) ? intr[4 + 3 ] : 0.0;
    const double k4  = (NDIST > 4 ) ? intr[4 + 4 ] : 0.0;
    const double k5  = (NDIST > 5 ) ? intr[4 + 5 ] : 0.0;
    const double k6  = (NDIST > 6 ) ? intr[4 + 6 ] : 0.0;
    const double k7  = (NDIST > 7 ) ? intr[4 + 7 ] : 0.0;
    const double k8  = (NDIST > 8 ) ? intr[4 + 8 ] : 0.0;
    const double k9  = (NDIST > 9 ) ? intr[4 + 9 ] : 0.0;
    const double k10 = (NDIST > 10) ? intr[4 + 10] : 0.0;
    const double k11 = (NDIST > 11) ? intr[4 + 11] : 0.0;
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];

    const double iz = 1.0/p[2];
    const double X  = p[0]*iz;
    const double Y  = p[1]*iz;
    const double r2 = X*X + Y*Y;
    const double r4 = r2*r2;
    const double r6 = r4*r2;
    const double a1 = 2.0*X*Y;
    const double a2 = r2 + 2.0*X*X;
    const double a3 = r2 + 2.0*Y*Y;
    const double num  = 1.0 + k0*r2 + k1*r4 + k4*r6;
    const double iden = (NDIST > 5) ? 1.0/(1.0 + k5*r2 + k6*r4 + k7*r6) : 1.0;
    const double g    = num*iden;
    q[0] = (X*g + k2*a1 + k3*a2 + k8 *r2 + k9 *r4)*fx + cx;
    q[1] = (Y*g + k3*a1 + k2*a3 + k10*r2 + k11*r4)*fy + cy;

    if(!WITH_GRAD) return;

    const double dnum = k0 + 2.0*k1*r2 + 3.0*k4*r4;     // d/dr2
    const double dden = k5 + 2.0*k6*r2 + 3.0*k7*r4;
    const double dg   = (dnum - g*dden)*iden;
    const double Xdg  = X*dg, Ydg = Y*dg;
    const double ex   = k8  + 2.0*k9 *r2;
    const double ey   = k10 + 2.0*k11*r2;
    const double X2 = 2.0*X, Y2 = 2.0*Y;
    const double dxX = g + X2*(Xdg + 3.0*k3 + ex) + k2*Y2;
    const double dxY =     Y2*(Xdg +     k3 + ex) + k2*X2;
    const double dyX =     X2*(Ydg +     k2 + ey) + k3*Y2;
    const double dyY = g + Y2*(Ydg + 3.0*k2 + ey) + k3*X2;
    const double fxz = fx*iz, fyz = fy*iz;
    dq_dp[0][0] = fxz*dxX;
    dq_dp[0][1] = fxz*dxY;
    dq_dp[0][2] = -(X*dq_dp[0][0] + Y*dq_dp[0][1]);
    dq_dp[1][0] = fyz*dyX;
    dq_dp[1][1] = fyz*dyY;
    dq_dp[1][2] = -(X*dq_dp[1][0] + Y*dq_dp[1][1]);

    if constexpr (NDIST >= 4)
    {
        const double ux = fx*X*iden, uy = fy*Y*iden;
        dq_dk[0][0] = ux*r2;   dq_dk[1][0] = uy*r2;
        dq_dk[0][1] = ux*r4;   dq_dk[1][1] = uy*r4;
        dq_dk[0][2] = fx*a1;   dq_dk[1][2] = fy*a3;
        dq_dk[0][3] = fx*a2;   dq_dk[1][3] = fy*a1;
        if constexpr (NDIST >= 5) { dq_dk[0][4] = ux*r6;   dq_dk[1][4] = uy*r6; }
        if constexpr (NDIST >= 8)
        {
            const double tx = -ux*g, ty = -uy*g;        // f U num (-iden) iden
            dq_dk[0][5] = tx*r2;   dq_dk[1][5] = ty*r2;
            dq_dk[0][6] = tx*r4;   dq_dk[1][6] = ty*r4;
            dq_dk[0][7] = tx*r6;   dq_dk[1][7] = ty*r6;
        }
        if constexpr (NDIST >= 12)
        {
            dq_dk[0][8]  = fx*r2;  dq_dk[1][8]  = 0.0;
            dq_dk[0][9]  = fx*r4;  dq_dk[1][9]  = 0.0;
            dq_dk[0][10] = 0.0;    dq_dk[1][10] = fy*r2;
            dq_dk[0][11] = 0.0;    dq_dk[1][11] = fy*r4;
        }
    }
}

////////////////////////////////////////////////////////////////////////////////
// Closed-form central models without distortion parameters
////////////////////////////////////////////////////////////////////////////////
// q = 2 p_xy/(|p| + p_z) f + c
template<bool WITH_GRAD>
MRCAL_AMD_HD
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
MRCAL_AMD_HD
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
MRCAL_AMD_HD
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

////////////////////////////////////////////////////////////////////////////////
// "perturb the point, then pinhole" models: CAHVOR, CAHVORE
////////////////////////////////////////////////////////////////////////////////
// q = f pd_xy/pd_z + c given the perturbed point pd, dpd/dp (3x3, row-major)
// and dpd/dk (3 x NDIST, row-major)
template<int NDIST, bool WITH_GRAD>
MRCAL_AMD_HD
void pinhole_of_perturbed(double* q, double (*dq_dp)[3], double (*dq_dk)[NDIST],
                          const double* pd, const double* dpd_dp, const double* dpd_dk,
                          const double* intr)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double iz = 1.0/pd[2];
    q[0] = pd[0]*iz*fx + cx;
    q[1] = pd[1]*iz*fy + cy;
    if(!WITH_GRAD) return;
    const double gx[3] = { fx*iz, 0.0,   -fx*pd[0]*iz*iz };
    const double gy[3] = { 0.0,   fy*iz, -fy*pd[1]*iz*iz };
    for(int j=0;j<3;j++)
    {
        dq_dp[0][j] = gx[0]*dpd_dp[0*3+j] + gx[1]*dpd_dp[1*3+j] + gx[2]*dpd_dp[2*3+j];
        dq_dp[1][j] = gy[0]*dpd_dp[0*3+j] + gy[1]*dpd_dp[1*3+j] + gy[2]*dpd_dp[2*3+j];
    }
    for(int i=0;i<NDIST;i++)
    {
        const double dx = dpd_dk[0*NDIST+i], dy = dpd_dk[1*NDIST+i], dz = dpd_dk[2*NDIST+i];
        dq_dk[0][i] = fx*iz*(dx - pd[0]*iz*dz);
        dq_dk[1][i] = fy*iz*(dy - pd[1]*iz*dz);
    }
}

// CAHVOR. Distortions: (alpha, beta, r0, r1, r2). The optical axis is
// o = (sin a cos b, sin b, cos a cos b); with w = p.o, tau = |p|^2/w^2 - 1,
// mu = r0 + r1 tau + r2 tau^2:   pd = p + mu (p - w o)
template<bool WITH_GRAD>
MRCAL_AMD_HD
void project_cahvor(double* q, double (*dq_dp)[3], double (*dq_dk)[5],
                    const double* p, const double* intr)
{
    const double alpha = intr[4], beta = intr[5], r0 = intr[6], r1 = intr[7], r2 = intr[8];
    double sa, ca, sb, cb;
    sincos(alpha, &sa, &ca);
    sincos(beta,  &sb, &cb);
    const double o[3]   = {  sa*cb, sb,  ca*cb };
    const double doa[3] = {  ca*cb, 0.0, -sa*cb };   // do/dalpha
    const double dob[3] = { -sa*sb, cb,  -ca*sb };   // do/dbeta

    const double n2  = p[0]*p[0] + p[1]*p[1] + p[2]*p[2];
    const double w   = p[0]*o[0] + p[1]*o[1] + p[2]*o[2];
    const double iw  = 1.0/w;
    const double tau = n2*iw*iw - 1.0;
    const double mu  = r0 + tau*r1 + tau*tau*r2;

    double pd[3], lat[3]; // lat = p - w o: the component of p across the axis
    for(int i=0;i<3;i++)
    {
        lat[i] = p[i] - w*o[i];
        pd[i]  = p[i] + mu*lat[i];
    }
    if(!WITH_GRAD)
    {
        pinhole_of_perturbed<5,false>(q, NULL, NULL, pd, NULL, NULL, intr);
        return;
    }

    const double dmu_dtau = r1 + 2.0*tau*r2;
    const double dtau_dw  = -2.0*n2*iw*iw*iw;
    const double dwa = p[0]*doa[0] + p[1]*doa[1] + p[2]*doa[2];
    const double dwb = p[0]*dob[0] + p[1]*dob[1] + p[2]*dob[2];
    // mu's partials wrt (alpha, beta, r0, r1, r2)
    const double dmu_dk[5] = { dmu_dtau*dtau_dw*dwa, dmu_dtau*dtau_dw*dwb, 1.0, tau, tau*tau };

    double dpd_dp[9], dpd_dk[15];
    for(int i=0;i<3;i++)
    {
        for(int j=0;j<3;j++)
        {
            // dmu/dp_j = dmu/dtau (2 p_j/w^2 + dtau/dw o_j)
            const double dmu_dpj = dmu_dtau*(2.0*p[j]*iw*iw + dtau_dw*o[j]);
            dpd_dp[3*i+j] = ((i==j) ? (mu + 1.0) : 0.0) + lat[i]*dmu_dpj - mu*o[i]*o[j];
        }
        for(int kk=0;kk<5;kk++) dpd_dk[5*i+kk] = dmu_dk[kk]*lat[i];
        dpd_dk[5*i+0] -= mu*(dwa*o[i] + w*doa[i]);
        dpd_dk[5*i+1] -= mu*(dwb*o[i] + w*dob[i]);
    }
    pinhole_of_perturbed<5,true>(q, dq_dp, dq_dk, pd, dpd_dp, dpd_dk, intr);
}

// CAHVORE (noncentral). Distortions: (alpha, beta, r0, r1, r2, e0, e1, e2) +
// the configuration value "linearity". Returns false if the Newton iteration
// for theta does not converge or theta is out of bounds: the caller then
// zeroes q and all gradients (mrcal.c:2492-2513).
//
// The angle theta solves a scalar equation in (zeta, l, e0, e1, e2) by Newton
// iterations started at atan2(l, zeta); like the reference, the derivatives
// are carried THROUGH the iterations (forward mode), here with 5 independent
// variables (zeta, l, e0, e1, e2) and the chain rule to (p, alpha, beta)
// applied once afterwards, which is the same linear map
template<bool WITH_GRAD>
MRCAL_AMD_HD
bool project_cahvore(double* q, double (*dq_dp)[3], double (*dq_dk)[8],
                     const double* p, const double* intr, double linearity)
{
    // independent variables of the outer duals: 0..7 = the 8 distortions, 8..10 = p
    typedef Dual<WITH_GRAD ? 11 : 0> D;
    typedef Dual<WITH_GRAD ? 5  : 0> D5;

    D P[3];
    for(int i=0;i<3;i++) P[i] = D::variable(p[i], 8+i);
    const D alpha = D::variable(intr[4], 0), beta = D::variable(intr[5], 1);
    const D r0 = D::variable(intr[6], 2), r1 = D::variable(intr[7], 3), r2 = D::variable(intr[8], 4);

    D sa, ca, sb, cb;
    dsincos(alpha, &sa, &ca);
    dsincos(beta,  &sb, &cb);
    const D o[3] = { cb*sa, sb, cb*ca };

    const D zeta = P[0]*o[0] + P[1]*o[1] + P[2]*o[2];
    D ll[3];
    for(int i=0;i<3;i++) ll[i] = P[i] - o[i]*zeta;
    const D l = dsqrt(ll[0]*ll[0] + ll[1]*ll[1] + ll[2]*ll[2]);

    // Newton, in the reduced variables
    const D5 z5 = D5::variable(zeta.x, 0), l5 = D5::variable(l.x, 1);
    const D5 e0 = D5::variable(intr[9], 2), e1 = D5::variable(intr[10], 3), e2 = D5::variable(intr[11], 4);
    D5 th = datan2(l5, z5);
    int inewton;
    for(inewton = 100; inewton; inewton--)
    {
        D5 s, c;
        dsincos(th, &s, &c);
        const D5 th2 = th*th, th3 = th*th2, th4 = th*th3;
        const D5 poly = e0 + e1*th2 + e2*th4;
        const D5 upsilon =
            z5*c + l5*s
            + (c - 1.0)*poly
            - (th - s)*(e1*th*2.0 + e2*th3*4.0);
        const D5 dth = (z5*s - l5*c - (th - s)*poly)/upsilon;
        th = th - dth;
        if(fabs(dth.x) < 1e-8) break;
    }
    if(inewton == 0) return false;
    if(th.x*fabs(linearity) > M_PI/2.0) return false;

    D pd[3];
    if(th.x > 1e-8)
    {
        // theta in the outer variables
        D theta(th.x);
        if(WITH_GRAD)
        {
            for(int i=0;i<11;i++) theta.d[i] = th.d[0]*zeta.d[i] + th.d[1]*l.d[i];
            theta.d[5] += th.d[2];
            theta.d[6] += th.d[3];
            theta.d[7] += th.d[4];
        }
        D chi;
        if(linearity < -1e-15)      chi = dsin(theta*linearity)/linearity;
        else if(linearity > 1e-15)  chi = dtan(theta*linearity)/linearity;
        else                        chi = theta;
        const D chi2 = chi*chi, chi4 = chi2*chi2;
        const D zetap = l/chi;
        const D mu1   = r0 + r1*chi2 + r2*chi4 + 1.0;
        for(int i=0;i<3;i++) pd[i] = o[i]*zetap + ll[i]*mu1;
    }
    else
        for(int i=0;i<3;i++) pd[i] = P[i];

    const double pdv[3] = { pd[0].x, pd[1].x, pd[2].x };
    if(!WITH_GRAD)
    {
        pinhole_of_perturbed<8,false>(q, NULL, NULL, pdv, NULL, NULL, intr);
        return true;
    }
    double dpd_dp[9], dpd_dk[24];
    for(int i=0;i<3;i++)
    {
        for(int j=0;j<3;j++) dpd_dp[3*i+j] = pd[i].d[8+j];
        for(int j=0;j<8;j++) dpd_dk[8*i+j] = pd[i].d[j];
    }
    pinhole_of_perturbed<8,true>(q, dq_dp, dq_dk, pdv, dpd_dp, dpd_dk, intr);
    return true;
}

////////////////////////////////////////////////////////////////////////////////
// dispatch over the parametric models: both image rows
////////////////////////////////////////////////////////////////////////////////
template<int PROJ, int NDIST, bool WITH_GRAD>
MRCAL_AMD_HD
bool project_lens(double* q, double (*dq_dp)[3], double (*dq_dk)[NDIST > 0 ? NDIST : 1],
                  const double* p, const double* intr, const LensConfig& cfg)
{
    if(PROJ == PROJ_OPENCV)
        project_opencv_both<NDIST,WITH_GRAD>(q, dq_dp, dq_dk, p, intr);
    else if(PROJ == PROJ_STEREOGRAPHIC) project_stereographic<WITH_GRAD>(q, dq_dp, p, intr);
    else if(PROJ == PROJ_LONLAT)        project_lonlat<WITH_GRAD>(q, dq_dp, p, intr);
    else if(PROJ == PROJ_LATLON)        project_latlon<WITH_GRAD>(q, dq_dp, p, intr);
    else if(PROJ == PROJ_CAHVOR)        project_cahvor<WITH_GRAD>(q, dq_dp, (double(*)[5])dq_dk, p, intr);
    else if(PROJ == PROJ_CAHVORE)
    {
        if(!project_cahvore<WITH_GRAD>(q, dq_dp, (double(*)[8])dq_dk, p, intr, cfg.cahvore_linearity))
        {
            // The reference zeroes the result and (part of) the gradients and
            // leaves the rest uninitialized; we zero all of it
            q[0] = q[1] = 0.0;
            if(WITH_GRAD)
            {
                for(int i=0;i<3;i++) dq_dp[0][i] = dq_dp[1][i] = 0.0;
                for(int i=0;i<NDIST;i++) dq_dk[0][i] = dq_dk[1][i] = 0.0;
            }
            return false;
        }
    }
    return true;
}

////////////////////////////////////////////////////////////////////////////////
// SPLINED_STEREOGRAPHIC
////////////////////////////////////////////////////////////////////////////////
// q = (u + deltau(u)) f + c, u = stereographic(p), deltau = two interleaved
// B-spline surfaces over a Nx x Ny grid of control points (the "distortion"
// parameters, [Ny][Nx][2]).
//
// Outputs besides q:
//   dq_dp      2x3, through both u and deltau(u)
//   ivar0      index, in the full intrinsics vector, of the first control
//              point of the (order+1)^2 patch this point touches
//   coef_x/y   the order+1 B-spline basis values along x and y: the gradient
//              of q[xy] wrt control point (iy,ix,xy) of the patch is
//              coef_x[ix] coef_y[iy] f[xy]; wrt (iy,ix,1-xy) it is 0
//   dq_dfxy    u + deltau
MRCAL_AMD_HD void bspline_basis(int order, double* v, double* dv, double x)
{
    if(order == 3)
    {
        // uniform cubic B-spline segment, x in [0,1] between the 2nd and 3rd
        // control points
        const double x2 = x*x, x3 = x2*x;
        v[0] = (-x3 + 3*x2 - 3*x + 1)/6;
        v[1] = (3*x3/2 - 3*x2 + 2)/3;
        v[2] = (-3*x3 + 3*x2 + 3*x + 1)/6;
        v[3] = x3/6;
        dv[0] = -x2/2 + x - 1./2.;
        dv[1] = 3*x2/2 - 2*x;
        dv[2] = -3*x2/2 + x + 1./2.;
        dv[3] = x2/2;
    }
    else
    {
        // uniform quadratic B-spline segment, x in [-1/2,1/2] around the
        // middle control point
        const double x2 = x*x;
        v[0] = (4*x2 - 4*x + 1)/8;
        v[1] = (3 - 4*x2)/4;
        v[2] = (4*x2 + 4*x + 1)/8;
        v[3] = 0.0;
        dv[0] = x - 1./2.;
        dv[1] = -2.*x;
        dv[2] = x + 1./2.;
        dv[3] = 0.0;
    }
}

template<bool WITH_GRAD>
MRCAL_AMD_HD
void project_splined(double* q, double (*dq_dp)[3], double* dq_dfxy,
                     int* ivar0, double* coef_x /*[4]*/, double* coef_y /*[4]*/,
                     const double* p, const double* intr, const LensConfig& cfg)
{
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const int order = cfg.spline_order, Nx = cfg.spline_Nx, Ny = cfg.spline_Ny;
    const int n = order + 1;

    const double mag   = sqrt(p[0]*p[0] + p[1]*p[1] + p[2]*p[2]);
    const double scale = 2.0/(mag + p[2]);
    const double u[2]  = { p[0]*scale, p[1]*scale };

    const double ix = u[0]*cfg.spline_segments_per_u + (double)(Nx-1)/2.;
    const double iy = u[1]*cfg.spline_segments_per_u + (double)(Ny-1)/2.;
    // the segment; out of bounds clamps to the nearest valid one (the
    // polynomial is then extrapolated)
    int ix0, iy0;
    if(order == 3)
    {
        ix0 = (int)ix;          iy0 = (int)iy;
        ix0 = ix0 < 1 ? 1 : (ix0 > Nx-3 ? Nx-3 : ix0);
        iy0 = iy0 < 1 ? 1 : (iy0 > Ny-3 ? Ny-3 : iy0);
    }
    else
    {
        ix0 = (int)(ix + 0.5);  iy0 = (int)(iy + 0.5);
        ix0 = ix0 < 1 ? 1 : (ix0 > Nx-2 ? Nx-2 : ix0);
        iy0 = iy0 < 1 ? 1 : (iy0 > Ny-2 ? Ny-2 : iy0);
    }
    *ivar0 = 4 + 2*((iy0-1)*Nx + (ix0-1));

    double dcx[4], dcy[4];
    bspline_basis(order, coef_x, dcx, ix - ix0);
    bspline_basis(order, coef_y, dcy, iy - iy0);

    // deltau and its derivatives wrt the fractional grid position
    const double* ctrl = intr + *ivar0;
    double du[2] = {0,0}, du_dx[2] = {0,0}, du_dy[2] = {0,0};
    for(int k=0;k<2;k++)
    {
        double rowv[4], rowd[4];
        for(int jy=0;jy<n;jy++)
        {
            double v = 0.0, d = 0.0;
            for(int jx=0;jx<n;jx++)
            {
                const double c = ctrl[jy*2*Nx + 2*jx + k];
                v += coef_x[jx]*c;
                d += dcx[jx]*c;
            }
            rowv[jy] = v; rowd[jy] = d;
        }
        for(int jy=0;jy<n;jy++)
        {
            du[k]    += coef_y[jy]*rowv[jy];
            du_dx[k] += coef_y[jy]*rowd[jy];
            du_dy[k] += dcy[jy]   *rowv[jy];
        }
    }

    q[0] = (u[0] + du[0])*fx + cx;
    q[1] = (u[1] + du[1])*fy + cy;
    if(dq_dfxy) { dq_dfxy[0] = u[0] + du[0]; dq_dfxy[1] = u[1] + du[1]; }
    if(!WITH_GRAD) return;

    const double A = -scale*scale/2.0;
    const double B = A/mag;
    const double du_dp[2][3] = { { p[0]*(B*p[0]) + scale, p[0]*(B*p[1]),         p[0]*(B*p[2] + A) },
                                 { p[1]*(B*p[0]),         p[1]*(B*p[1]) + scale, p[1]*(B*p[2] + A) } };
    // grid position -> u
    const double s = cfg.spline_segments_per_u;
    for(int j=0;j<3;j++)
    {
        dq_dp[0][j] = fx*(du_dp[0][j]*(1.0 + du_dx[0]*s) + du_dy[0]*s*du_dp[1][j]);
        dq_dp[1][j] = fy*(du_dp[1][j]*(1.0 + du_dy[1]*s) + du_dx[1]*s*du_dp[0][j]);
    }
}

// ONE image coordinate of the same projection (round 6: board_splined_rows_kernel, a lane per Jacobian ROW): coordinate
// k = 0 (x) or 1 (y) touches surface k's control points only, so a row's lane reads 16 of the 32 values and sums one
// surface. The expressions are project_splined()'s own, in its order, for that k: the same bits in q[k], dq_dp[k][.],
// dq_dfxy[k] and the coefficients. qk, dqk_dp[3], dqk_dfk: of coordinate k
template<bool WITH_GRAD>
MRCAL_AMD_HD
void project_splined_row(const int k, double* qk, double* dqk_dp, double* dqk_dfk,
                         int* ivar0, double* coef_x /*[4]*/, double* coef_y /*[4]*/,
                         const double* p, const double* intr, const LensConfig& cfg)
{
    const double fk = intr[k], ck = intr[2 + k];
    const int order = cfg.spline_order, Nx = cfg.spline_Nx, Ny = cfg.spline_Ny;
    const int n = order + 1;

    const double mag   = sqrt(p[0]*p[0] + p[1]*p[1] + p[2]*p[2]);
    const double scale = 2.0/(mag + p[2]);
    const double u[2]  = { p[0]*scale, p[1]*scale };

    const double ix = u[0]*cfg.spline_segments_per_u + (double)(Nx-1)/2.;
    const double iy = u[1]*cfg.spline_segments_per_u + (double)(Ny-1)/2.;
    int ix0, iy0;
    if(order == 3)
    {
        ix0 = (int)ix;          iy0 = (int)iy;
        ix0 = ix0 < 1 ? 1 : (ix0 > Nx-3 ? Nx-3 : ix0);
        iy0 = iy0 < 1 ? 1 : (iy0 > Ny-3 ? Ny-3 : iy0);
    }
    else
    {
        ix0 = (int)(ix + 0.5);  iy0 = (int)(iy + 0.5);
        ix0 = ix0 < 1 ? 1 : (ix0 > Nx-2 ? Nx-2 : ix0);
        iy0 = iy0 < 1 ? 1 : (iy0 > Ny-2 ? Ny-2 : iy0);
    }
    *ivar0 = 4 + 2*((iy0-1)*Nx + (ix0-1));

    double dcx[4], dcy[4];
    bspline_basis(order, coef_x, dcx, ix - ix0);
    bspline_basis(order, coef_y, dcy, iy - iy0);

    const double* ctrl = intr + *ivar0 + k;
    // (all the surface's values asked for before any is used: 16 loads in flight)
    double c[4][4];
#pragma unroll
    for(int jy=0;jy<4;jy++)
#pragma unroll
        for(int jx=0;jx<4;jx++)
            c[jy][jx] = (jy < n && jx < n) ? ctrl[jy*2*Nx + 2*jx] : 0.0;
    double du = 0.0, du_dx = 0.0, du_dy = 0.0;
    {
        double rowv[4], rowd[4];
#pragma unroll
        for(int jy=0;jy<4;jy++)
        {
            double v = 0.0, d = 0.0;
#pragma unroll
            for(int jx=0;jx<4;jx++)
                if(jx < n)
                {
                    v += coef_x[jx]*c[jy][jx];
                    d += dcx[jx]*c[jy][jx];
                }
            rowv[jy] = v; rowd[jy] = d;
        }
#pragma unroll
        for(int jy=0;jy<4;jy++)
            if(jy < n)
            {
                du    += coef_y[jy]*rowv[jy];
                du_dx += coef_y[jy]*rowd[jy];
                du_dy += dcy[jy]   *rowv[jy];
            }
    }

    *qk = (u[k] + du)*fk + ck;
    if(dqk_dfk) *dqk_dfk = u[k] + du;
    if(!WITH_GRAD) return;

    const double A = -scale*scale/2.0;
    const double B = A/mag;
    const double du_dp[2][3] = { { p[0]*(B*p[0]) + scale, p[0]*(B*p[1]),         p[0]*(B*p[2] + A) },
                                 { p[1]*(B*p[0]),         p[1]*(B*p[1]) + scale, p[1]*(B*p[2] + A) } };
    const double s = cfg.spline_segments_per_u;
    // (coordinate 0: fx (du0_dp (1 + du_dx s) + du_dy s du1_dp); coordinate 1: fy (du1_dp (1 + du_dy s) + du_dx s du0_dp))
    const double own = k ? du_dy : du_dx, other = k ? du_dx : du_dy;
    for(int j=0;j<3;j++)
        dqk_dp[j] = fk*(du_dp[k][j]*(1.0 + own*s) + other*s*du_dp[1-k][j]);
}

// mrcal.c:1904-1952
MRCAL_AMD_HD double spline_segments_per_u(int order, int Nx, double fov_x_deg)
{
    const int    Nknots_margin = (order == 2) ? 1 : 2;
    const double th_edge_x = fov_x_deg/2. * M_PI/180.;
    const double u_edge_x  = tan(th_edge_x/2.)*2;
    return (Nx - 1 - Nknots_margin)/(u_edge_x*2.);
}

} // namespace mrcal_amd
