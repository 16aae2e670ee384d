// Device-side geometry: forward-mode dual numbers, Rodrigues rotations and
// their derivatives, rotation composition.
//
// These follow the MATH (incl. the numerical branch points) of the reference so
// that results agree to rounding:
//   - rotation of a point by a Rodrigues vector + gradients:
//       poseutils-uses-autodiff.cc:16-80   (th^2 < 1e-10 small-angle branch)
//   - composition of two Rodrigues vectors + gradients:
//       poseutils-uses-autodiff.cc:278-769 (half-angle formula of Altmann 1989
//       with the A~0, B~0, C~0, C~pi, cosC<0 branches)
//   - rotation matrix from a Rodrigues vector + 27 partials:
//       poseutils-opencv.c:42-155          (|r|^2 < DBL_EPSILON^2 branch)
//   - sin(x)/x with a flat |x|<1e-5 branch: _autodiff.hh:256-288
// The implementation is new: one small dual-number type specialised for
// register-resident use on a 64-wide wavefront, no strides, no runtime N.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>

namespace mrcal_amd {

#define MRCAL_AMD_HD __host__ __device__ __forceinline__

// value + N partial derivatives
template<int N>
struct Dual
{
    double x;
    double d[N > 0 ? N : 1];

    MRCAL_AMD_HD Dual() {}
    MRCAL_AMD_HD Dual(double v) : x(v) { for(int i=0; i<N; i++) d[i] = 0.0; }
    // variable number ivar of the N independent variables
    static MRCAL_AMD_HD Dual variable(double v, int ivar)
    {
        Dual r(v);
        if(ivar >= 0 && ivar < N) r.d[ivar] = 1.0;
        return r;
    }
};

template<int N> MRCAL_AMD_HD Dual<N> operator+(const Dual<N>& a, const Dual<N>& b)
{ Dual<N> r; r.x = a.x+b.x; for(int i=0;i<N;i++) r.d[i] = a.d[i]+b.d[i]; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator-(const Dual<N>& a, const Dual<N>& b)
{ Dual<N> r; r.x = a.x-b.x; for(int i=0;i<N;i++) r.d[i] = a.d[i]-b.d[i]; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator-(const Dual<N>& a)
{ Dual<N> r; r.x = -a.x; for(int i=0;i<N;i++) r.d[i] = -a.d[i]; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator*(const Dual<N>& a, const Dual<N>& b)
{ Dual<N> r; r.x = a.x*b.x; for(int i=0;i<N;i++) r.d[i] = a.d[i]*b.x + a.x*b.d[i]; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator*(const Dual<N>& a, double b)
{ Dual<N> r; r.x = a.x*b; for(int i=0;i<N;i++) r.d[i] = a.d[i]*b; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator*(double b, const Dual<N>& a) { return a*b; }
template<int N> MRCAL_AMD_HD Dual<N> operator+(const Dual<N>& a, double b)
{ Dual<N> r = a; r.x += b; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator-(const Dual<N>& a, double b)
{ Dual<N> r = a; r.x -= b; return r; }
template<int N> MRCAL_AMD_HD Dual<N> operator/(const Dual<N>& a, const Dual<N>& b)
{
    Dual<N> r;
    r.x = a.x/b.x;
    const double inv2 = 1.0/(b.x*b.x);
    for(int i=0;i<N;i++) r.d[i] = (a.d[i]*b.x - a.x*b.d[i]) * inv2;
    return r;
}
template<int N> MRCAL_AMD_HD Dual<N> operator/(const Dual<N>& a, double b) { return a*(1.0/b); }

template<int N> MRCAL_AMD_HD Dual<N> dsqrt(const Dual<N>& a)
{
    Dual<N> r;
    r.x = sqrt(a.x);
    const double k = 1.0/(2.0*r.x);
    for(int i=0;i<N;i++) r.d[i] = a.d[i]*k;
    return r;
}
template<int N> MRCAL_AMD_HD void dsincos(const Dual<N>& a, Dual<N>* s, Dual<N>* c)
{
    double sv, cv;
    sincos(a.x, &sv, &cv);
    s->x = sv; c->x = cv;
    for(int i=0;i<N;i++) { s->d[i] = cv*a.d[i]; c->d[i] = -sv*a.d[i]; }
}
template<int N> MRCAL_AMD_HD Dual<N> dtan(const Dual<N>& a)
{
    double sv, cv;
    sincos(a.x, &sv, &cv);
    Dual<N> r;
    r.x = sv/cv;
    const double k = 1.0/(cv*cv);
    for(int i=0;i<N;i++) r.d[i] = a.d[i]*k;
    return r;
}
// atan2(y,x): d = (dy x - y dx)/(x^2+y^2)   (_autodiff.hh:208-224)
template<int N> MRCAL_AMD_HD Dual<N> datan2(const Dual<N>& y, const Dual<N>& x)
{
    Dual<N> r;
    r.x = atan2(y.x, x.x);
    const double inv = 1.0/(y.x*y.x + x.x*x.x);
    for(int i=0;i<N;i++) r.d[i] = (y.d[i]*x.x - y.x*x.d[i]) * inv;
    return r;
}
template<int N> MRCAL_AMD_HD Dual<N> dsin(const Dual<N>& a)
{
    double sv, cv;
    sincos(a.x, &sv, &cv);
    Dual<N> r;
    r.x = sv;
    for(int i=0;i<N;i++) r.d[i] = cv*a.d[i];
    return r;
}
template<int N> MRCAL_AMD_HD Dual<N> dacos(const Dual<N>& a)
{
    Dual<N> r;
    r.x = acos(a.x);
    const double k = -1.0/sqrt(1.0 - a.x*a.x);
    for(int i=0;i<N;i++) r.d[i] = a.d[i]*k;
    return r;
}
// sin(x)/x given sin(x). Flat (value 1, zero gradient) for |x| < 1e-5
template<int N> MRCAL_AMD_HD Dual<N> dsinx_over_x(const Dual<N>& x, const Dual<N>& sinx)
{
    if(fabs(x.x) < 1e-5) return Dual<N>(1.0);
    return sinx/x;
}

// y = R(r) x  (or R(-r) x if inverted). r,x,y are 3-vectors of duals
template<int N>
MRCAL_AMD_HD void rotate_point_r_dual(Dual<N>* y, const Dual<N>* r, const Dual<N>* x, bool inverted)
{
    const double sgn = inverted ? -1.0 : 1.0;
    const Dual<N> th2 = r[0]*r[0] + r[1]*r[1] + r[2]*r[2];
    const Dual<N> cr[3] = { (r[1]*x[2] - r[2]*x[1])*sgn,
                            (r[2]*x[0] - r[0]*x[2])*sgn,
                            (r[0]*x[1] - r[1]*x[0])*sgn };
    const Dual<N> rx = r[0]*x[0] + r[1]*x[1] + r[2]*x[2];
    if(th2.x < 1e-10)
    {
        // lim th->0: x + r cross x + r (r.x)/2
        for(int i=0;i<3;i++) y[i] = x[i] + cr[i] + r[i]*rx/2.0;
        return;
    }
    const Dual<N> th = dsqrt(th2);
    Dual<N> s, c;
    dsincos(th, &s, &c);
    const Dual<N> a = s/th;
    const Dual<N> b = (Dual<N>(1.0) - c)/th2;
    for(int i=0;i<3;i++) y[i] = x[i]*c + cr[i]*a + r[i]*rx*b;
}

// r01 = rodrigues vector of R(r0) R(r1)
template<int N>
MRCAL_AMD_HD void compose_r_dual(Dual<N>* r01, const Dual<N>* r0, const Dual<N>* r1)
{
    const double eps = 1e-8;

    const Dual<N> n0 = r0[0]*r0[0] + r0[1]*r0[1] + r0[2]*r0[2];
    const Dual<N> n1 = r1[0]*r1[0] + r1[1]*r1[1] + r1[2]*r1[2];
    const double  A_val = sqrt(n0.x)/2.0;
    const double  B_val = sqrt(n1.x)/2.0;

    const Dual<N> inner = r0[0]*r1[0] + r0[1]*r1[1] + r0[2]*r1[2];
    const Dual<N> cr[3] = { r0[1]*r1[2] - r0[2]*r1[1],
                            r0[2]*r1[0] - r0[0]*r1[2],
                            r0[0]*r1[1] - r0[1]*r1[0] };

    if(A_val < eps/2.0)
    {
        if(B_val < eps)
        {
            // both tiny: first order
            for(int i=0;i<3;i++) r01[i] = r0[i] + r1[i];
            return;
        }
        // r0 tiny: perturbation of r1
        const Dual<N> B  = dsqrt(n1)/2.0;
        const Dual<N> Bt = B/dtan(B);
        const Dual<N> k  = Dual<N>(1.0) - inner*(Bt - 1.0)/(B*B*4.0);
        for(int i=0;i<3;i++) r01[i] = r1[i]*k + r0[i]*Bt + cr[i]/2.0;
        return;
    }
    if(B_val < eps)
    {
        // r1 tiny: perturbation of r0
        const Dual<N> A  = dsqrt(n0)/2.0;
        const Dual<N> At = A/dtan(A);
        const Dual<N> k  = Dual<N>(1.0) - inner*(At - 1.0)/(A*A*4.0);
        for(int i=0;i<3;i++) r01[i] = r0[i]*k + r1[i]*At + cr[i]/2.0;
        return;
    }

    const Dual<N> A = dsqrt(n0)/2.0;
    const Dual<N> B = dsqrt(n1)/2.0;
    Dual<N> sA, cA, sB, cB;
    dsincos(A, &sA, &cA);
    dsincos(B, &sB, &cB);
    const Dual<N> sA_A = dsinx_over_x(A, sA);
    const Dual<N> sB_B = dsinx_over_x(B, sB);

    // half-angle composition: cosC, and u = r01 sinC/C
    const Dual<N> cosC = cA*cB - sA_A*sB_B*inner/4.0;
    for(int i=0;i<3;i++)
        r01[i] = sA_A*cB*r0[i] + sB_B*cA*r1[i] + sA_A*sB_B*cr[i]/2.0;

    if(cosC.x - 1.0 > -eps*eps/2.0)
    {
        // C ~ 0: sinC/C ~ 1, u is the answer
    }
    else if(cosC.x + 1.0 < eps*eps/2.0)
    {
        // C ~ pi: a full turn. Wrap around: r' ~ -u
        for(int i=0;i<3;i++) r01[i] = -r01[i];
    }
    else
    {
        const Dual<N> C    = dacos(cosC);
        const Dual<N> sinC = dsqrt(Dual<N>(1.0) - cosC*cosC);
        // cosC<0: report the equivalent rotation with C in [-pi/2,pi/2]
        const Dual<N> k = (cosC.x < 0.0) ? (C - M_PI)/sinC : C/sinC;
        for(int i=0;i<3;i++) r01[i] = r01[i]*k;
    }
}

////////////////////////////////////////////////////////////////////////////////
// Duals that know WHICH variables they depend on (round 6): value + the partials with respect to the variables
// [LO, HI) of a common numbering; every other partial is structurally zero and is neither stored nor computed.
// The triangulated pairs' residual depends on twelve variables - the two cameras' poses -, but half of its chain
// involves one camera's six only (and the observation vector's rotation three): as Dual<12> every operation of that
// half spent half its multiply-adds on zeros the compiler may not drop (0 * x is not 0 for every x). The range of a
// result is the hull of its operands' ranges; a partial both operands have is the expression Dual<N> has for it, one
// that only one has is that expression less its zero term: the same value, to the bit.
////////////////////////////////////////////////////////////////////////////////
template<int LO, int HI>
struct DualR
{
    static constexpr int lo = LO, hi = HI, n = (HI > LO) ? HI - LO : 0;
    double x;
    double d[n > 0 ? n : 1];
    MRCAL_AMD_HD DualR() {}
    MRCAL_AMD_HD DualR(double v) : x(v) { for(int i=0;i<n;i++) d[i] = 0.0; }
    // from a dual of a narrower range
    template<int L2, int H2>
    MRCAL_AMD_HD DualR(const DualR<L2,H2>& o) : x(o.x)
    {
        static_assert(H2 <= L2 || (LO <= L2 && H2 <= HI), "a dual cannot be narrowed");
        for(int i=0;i<n;i++) d[i] = (LO + i >= L2 && LO + i < H2) ? o.d[(LO + i - L2) < 0 ? 0 : (LO + i - L2)] : 0.0;
    }
    // the partial with respect to variable ivar (0 outside the range)
    MRCAL_AMD_HD double partial(int ivar) const { return (ivar >= LO && ivar < HI) ? d[ivar - LO] : 0.0; }
    static MRCAL_AMD_HD DualR variable(double v, int ivar) { DualR r(v); if(ivar >= LO && ivar < HI) r.d[ivar - LO] = 1.0; return r; }
};
template<class A, class B> struct DualHull
{
    static constexpr bool ea = A::hi <= A::lo, eb = B::hi <= B::lo;
    static constexpr int lo = ea ? B::lo : (eb ? A::lo : (A::lo < B::lo ? A::lo : B::lo));
    static constexpr int hi = ea ? B::hi : (eb ? A::hi : (A::hi > B::hi ? A::hi : B::hi));
    typedef DualR<(ea && eb) ? 0 : lo, (ea && eb) ? 0 : hi> type;
};
#define MRCAL_AMD_DR2 template<int LA, int HA, int LB, int HB> MRCAL_AMD_HD typename DualHull<DualR<LA,HA>, DualR<LB,HB> >::type
#define MRCAL_AMD_IN(R, L, H, i) ((R::lo + (i)) >= (L) && (R::lo + (i)) < (H))
MRCAL_AMD_DR2 operator+(const DualR<LA,HA>& a, const DualR<LB,HB>& b)
{
    typedef typename DualHull<DualR<LA,HA>, DualR<LB,HB> >::type R; R r; r.x = a.x + b.x;
    for(int i=0;i<R::n;i++)
    {
        const bool ia = MRCAL_AMD_IN(R,LA,HA,i), ib = MRCAL_AMD_IN(R,LB,HB,i);
        const double da = ia ? a.d[ia ? R::lo + i - LA : 0] : 0.0, db = ib ? b.d[ib ? R::lo + i - LB : 0] : 0.0;
        r.d[i] = (ia && ib) ? da + db : (ia ? da : (ib ? db : 0.0));
    }
    return r;
}
MRCAL_AMD_DR2 operator-(const DualR<LA,HA>& a, const DualR<LB,HB>& b)
{
    typedef typename DualHull<DualR<LA,HA>, DualR<LB,HB> >::type R; R r; r.x = a.x - b.x;
    for(int i=0;i<R::n;i++)
    {
        const bool ia = MRCAL_AMD_IN(R,LA,HA,i), ib = MRCAL_AMD_IN(R,LB,HB,i);
        const double da = ia ? a.d[ia ? R::lo + i - LA : 0] : 0.0, db = ib ? b.d[ib ? R::lo + i - LB : 0] : 0.0;
        r.d[i] = (ia && ib) ? da - db : (ia ? da : (ib ? -db : 0.0));
    }
    return r;
}
MRCAL_AMD_DR2 operator*(const DualR<LA,HA>& a, const DualR<LB,HB>& b)
{
    typedef typename DualHull<DualR<LA,HA>, DualR<LB,HB> >::type R; R r; r.x = a.x*b.x;
    for(int i=0;i<R::n;i++)
    {
        const bool ia = MRCAL_AMD_IN(R,LA,HA,i), ib = MRCAL_AMD_IN(R,LB,HB,i);
        const double da = ia ? a.d[ia ? R::lo + i - LA : 0] : 0.0, db = ib ? b.d[ib ? R::lo + i - LB : 0] : 0.0;
        r.d[i] = (ia && ib) ? da*b.x + a.x*db : (ia ? da*b.x : (ib ? a.x*db : 0.0));
    }
    return r;
}
MRCAL_AMD_DR2 operator/(const DualR<LA,HA>& a, const DualR<LB,HB>& b)
{
    typedef typename DualHull<DualR<LA,HA>, DualR<LB,HB> >::type R; R r; r.x = a.x/b.x;
    const double inv2 = 1.0/(b.x*b.x);
    for(int i=0;i<R::n;i++)
    {
        const bool ia = MRCAL_AMD_IN(R,LA,HA,i), ib = MRCAL_AMD_IN(R,LB,HB,i);
        const double da = ia ? a.d[ia ? R::lo + i - LA : 0] : 0.0, db = ib ? b.d[ib ? R::lo + i - LB : 0] : 0.0;
        r.d[i] = (ia && ib) ? (da*b.x - a.x*db)*inv2 : (ia ? (da*b.x)*inv2 : (ib ? (-(a.x*db))*inv2 : 0.0));
    }
    return r;
}
#undef MRCAL_AMD_DR2
#undef MRCAL_AMD_IN
template<int L, int H> MRCAL_AMD_HD DualR<L,H> operator-(const DualR<L,H>& a)
{ DualR<L,H> r; r.x = -a.x; for(int i=0;i<DualR<L,H>::n;i++) r.d[i] = -a.d[i]; return r; }
template<int L, int H> MRCAL_AMD_HD DualR<L,H> operator*(const DualR<L,H>& a, double b)
{ DualR<L,H> r; r.x = a.x*b; for(int i=0;i<DualR<L,H>::n;i++) r.d[i] = a.d[i]*b; return r; }
template<int L, int H> MRCAL_AMD_HD DualR<L,H> operator*(double b, const DualR<L,H>& a) { return a*b; }
template<int L, int H> MRCAL_AMD_HD DualR<L,H> operator+(const DualR<L,H>& a, double b) { DualR<L,H> r = a; r.x += b; return r; }
template<int L, int H> MRCAL_AMD_HD DualR<L,H> operator-(const DualR<L,H>& a, double b) { DualR<L,H> r = a; r.x -= b; return r; }
template<int L, int H> MRCAL_AMD_HD DualR<L,H> operator/(const DualR<L,H>& a, double b) { return a*(1.0/b); }
template<int L, int H> MRCAL_AMD_HD DualR<L,H> dsqrt(const DualR<L,H>& a)
{
    DualR<L,H> r; r.x = sqrt(a.x);
    const double k = 1.0/(2.0*r.x);
    for(int i=0;i<DualR<L,H>::n;i++) r.d[i] = a.d[i]*k;
    return r;
}
template<int L, int H> MRCAL_AMD_HD void dsincos(const DualR<L,H>& a, DualR<L,H>* s, DualR<L,H>* c)
{
    double sv, cv;
    sincos(a.x, &sv, &cv);
    s->x = sv; c->x = cv;
    for(int i=0;i<DualR<L,H>::n;i++) { s->d[i] = cv*a.d[i]; c->d[i] = -sv*a.d[i]; }
}

// y = R(r) x  (or R(-r) x if inverted): rotate_point_r_dual() for duals of any ranges (y's: the hull of r's and x's)
template<class DY, class DRr, class DX>
MRCAL_AMD_HD void rotate_point_r_dualr(DY* y, const DRr* r, const DX* x, bool inverted)
{
    const double sgn = inverted ? -1.0 : 1.0;
    const auto th2 = r[0]*r[0] + r[1]*r[1] + r[2]*r[2];
    const DY cr[3] = { DY((r[1]*x[2] - r[2]*x[1])*sgn),
                       DY((r[2]*x[0] - r[0]*x[2])*sgn),
                       DY((r[0]*x[1] - r[1]*x[0])*sgn) };
    const DY rx = r[0]*x[0] + r[1]*x[1] + r[2]*x[2];
    if(th2.x < 1e-10)
    {
        for(int i=0;i<3;i++) y[i] = x[i] + cr[i] + r[i]*rx/2.0;
        return;
    }
    const auto th = dsqrt(th2);
    auto s = th, c = th;
    dsincos(th, &s, &c);
    const auto a = s/th;
    const auto b = (decltype(c)(1.0) - c)/th2;
    for(int i=0;i<3;i++) y[i] = x[i]*c + cr[i]*a + r[i]*rx*b;
}

// R (row-major 3x3) and dR[i][j]/dr[k] stored as dR[9*i + 3*j + k]
MRCAL_AMD_HD void R_from_r_with_grad(double* R, double* dR, const double* r)
{
    const double n2 = r[0]*r[0] + r[1]*r[1] + r[2]*r[2];
    if(n2 < DBL_EPSILON*DBL_EPSILON)
    {
        for(int i=0;i<9;i++)  R[i]  = 0.0;
        R[0] = R[4] = R[8] = 1.0;
        for(int i=0;i<27;i++) dR[i] = 0.0;
        // skew generators
        dR[9*1 + 3*2 + 0] = -1.0;  dR[9*2 + 3*1 + 0] =  1.0;
        dR[9*2 + 3*0 + 1] = -1.0;  dR[9*0 + 3*2 + 1] =  1.0;
        dR[9*0 + 3*1 + 2] = -1.0;  dR[9*1 + 3*0 + 2] =  1.0;
        return;
    }
    const double th  = sqrt(n2);
    double s, c;
    sincos(th, &s, &c);
    const double c1  = 1.0 - c;
    const double ith = 1.0/th;
    const double u[3] = { r[0]*ith, r[1]*ith, r[2]*ith };

    // R = c I + (1-c) u ut + s [u]x
    for(int i=0;i<3;i++)
        for(int j=0;j<3;j++)
            R[3*i+j] = c1*u[i]*u[j] + ((i==j) ? c : 0.0);
    R[3*0+1] -= s*u[2];  R[3*0+2] += s*u[1];
    R[3*1+0] += s*u[2];  R[3*1+2] -= s*u[0];
    R[3*2+0] -= s*u[1];  R[3*2+1] += s*u[0];

    // With u = r/th:  du_i/dr_k = (delta_ik - u_i u_k)/th,  dth/dr_k = u_k
    //   dR_ij/dr_k = -s u_k delta_ij
    //              + s u_k u_i u_j
    //              + c1 (du_i/dr_k u_j + u_i du_j/dr_k)
    //              + c u_k [u]x_ij
    //              + s d[u]x_ij/dr_k
    const double a2 = ith*c1;
    const double a4 = ith*s;
    for(int k=0;k<3;k++)
    {
        const double a0 = -s*u[k];
        const double a1 = (s - 2.0*a2)*u[k];
        const double a3 = (c - a4)*u[k];
        for(int i=0;i<3;i++)
            for(int j=0;j<3;j++)
            {
                double v = a1*u[i]*u[j];
                if(i==j) v += a0;
                if(i==k) v += a2*u[j];
                if(j==k) v += a2*u[i];
                dR[9*i + 3*j + k] = v;
            }
        // skew part: [u]x = [[0,-u2,u1],[u2,0,-u0],[-u1,u0,0]]
        dR[9*0 + 3*1 + k] -= a3*u[2];  dR[9*0 + 3*2 + k] += a3*u[1];
        dR[9*1 + 3*0 + k] += a3*u[2];  dR[9*1 + 3*2 + k] -= a3*u[0];
        dR[9*2 + 3*0 + k] -= a3*u[1];  dR[9*2 + 3*1 + k] += a3*u[0];
    }
    // s d[u]x/dr_k = a4 (generator_k)
    dR[9*1 + 3*2 + 0] -= a4;  dR[9*2 + 3*1 + 0] += a4;
    dR[9*2 + 3*0 + 1] -= a4;  dR[9*0 + 3*2 + 1] += a4;
    dR[9*0 + 3*1 + 2] -= a4;  dR[9*1 + 3*0 + 2] += a4;
}

} // namespace mrcal_amd
