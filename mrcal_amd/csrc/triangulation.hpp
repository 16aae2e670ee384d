// The triangulated-point residual: one scalar per PAIR of observations of one
// point, and its derivatives with respect to the two cameras' extrinsics.
//
// Reference behaviour being reproduced (math and branch points):
//   triangulation.cc:958-1123  _mrcal_triangulated_error()  (Lee-Civera "mid2"
//                              midpoint, angle to it, divergence penalty)
//   triangulation.cc:576-638   chirality()
//   triangulation.cc:767-805   angle_error__assume_small()  (th^2 < 1e-21 -> 0)
//   triangulation.cc:899-953   sigmoid()
//   mrcal.c:5180-5653          the loop: geometry of the pair, chain rule
// The reference chains hand-written gradients of the pose operations into the
// 6-variable autodiff of the error function. Here ONE forward-mode pass over
// the 12 extrinsics variables of the pair does the same: it is the same linear
// map - with duals that carry only the partials that can be nonzero (round 6). __host__ __device__: the host build is used by the outlier logic
// (solver.cpp) and by the CPU tests.
#pragma once
#include "device_math.hpp"

namespace mrcal_amd {

// (round 6) Everything below is written once for duals of ANY ranges (device_math.hpp DualR: a dual knows which of the
// pair's twelve variables it depends on): camera 0's part of the chain carries six partials (its ray's rotation three),
// camera 1's rotation adds three, its translation three, and only the error function itself carries all twelve. As
// Dual<12> throughout - until round 5 - triangulated_kernel was BASELINE configuration 5's longest launch: 25.8 us for 67 000
// pairs, its lanes issuing multiply-adds on structural zeros half the time

template<class A, class B> MRCAL_AMD_HD auto tri_cross_norm2(const A* a, const B* b) -> decltype(a[0]*b[0])
{
    const auto c0 = a[1]*b[2] - a[2]*b[1];
    const auto c1 = a[2]*b[0] - a[0]*b[2];
    const auto c2 = a[0]*b[1] - a[1]*b[0];
    return c0*c0 + c1*c1 + c2*c2;
}

// small angle between two vectors: sqrt(2 (1 - |cos|)); exactly 0 (and flat)
// below 1e-21
template<class A, class B> MRCAL_AMD_HD auto tri_angle_error_small(const A* v0, const B* v1) -> decltype(v0[0]*v1[0])
{
    typedef decltype(v0[0]*v1[0]) H;
    const auto i00 = v0[0]*v0[0] + v0[1]*v0[1] + v0[2]*v0[2];
    const auto i11 = v1[0]*v1[0] + v1[1]*v1[1] + v1[2]*v1[2];
    const auto i01 = v0[0]*v1[0] + v0[1]*v1[1] + v0[2]*v1[2];
    H costh = i01/dsqrt(i00*i11);
    if(costh.x < 0.0) costh = -costh;           // barely-divergent rays
    H th_sq = costh*(-2.0) + 2.0;
    if(th_sq.x < 1e-21) return H(0.0);
    return dsqrt(th_sq);
}

// 0 below 0, 1 above knee, two parabolas in between
template<class D> MRCAL_AMD_HD D tri_sigmoid(const D& x, double knee)
{
    if(x.x <= 0.0)  return D(0.0);
    if(knee <= x.x) return D(1.0);
    const double bq = 2./knee, c = 1./2.;
    const double a = (x.x < knee/2.0) ? 2./knee/knee : -2./knee/knee;
    const D dx = x - knee/2.;
    return dx*(dx*a + bq) + c;
}

// The points l0 v0 and t01 + l1 v1 should coincide. Would flipping the sign of
// l0, l1 or both bring them closer? worsening* = how much farther apart a flip
// puts them; all three positive: the signs are right
template<class W, class L, class V0, class V1, class T> MRCAL_AMD_HD
bool tri_chirality(W* w0, W* w1, W* w01,
                   const L& l0, const V0* v0, const L& l1, const V1* v1, const T* t01)
{
    *w0 = W(0.0); *w1 = W(0.0); *w01 = W(0.0);
    for(int i=0;i<3;i++)
    {
        const W xn  = ( l1*v1[i] + t01[i]) - l0*v0[i];
        const W x0  = ( l1*v1[i] + t01[i]) + l0*v0[i];
        const W x1  = (-(l1*v1[i]) + t01[i]) - l0*v0[i];
        const W x01 = (-(l1*v1[i]) + t01[i]) + l0*v0[i];
        *w0  = *w0  + (x0 *x0  - xn*xn);
        *w1  = *w1  + (x1 *x1  - xn*xn);
        *w01 = *w01 + (x01*x01 - xn*xn);
    }
    return w0->x > 0.0 && w1->x > 0.0 && w01->x > 0.0;
}

// v0: the observation vector of the camera we are in (no derivatives);
// v1: the other camera's observation vector, rotated into this camera;
// t01: the other camera's position in this camera's coordinates
template<class V1, class T> MRCAL_AMD_HD
auto tri_error(const double* v0_in, const V1* v1, const T* t01, bool* convergent) -> decltype(v1[0]*t01[0])
{
    typedef DualR<0,0> K;                           // a constant
    typedef decltype(v1[0]*t01[0]) H;               // what depends on everything
    const K v0[3] = { K(v0_in[0]), K(v0_in[1]), K(v0_in[2]) };
    const auto pr = K(1.0)/tri_cross_norm2(v0, v1);
    const H l0 = dsqrt(tri_cross_norm2(v1, t01)*pr);
    const H l1 = dsqrt(tri_cross_norm2(v0, t01)*pr);
    H m[3];
    for(int i=0;i<3;i++) m[i] = (v0[i]*l0 + t01[i] + v1[i]*l1)/2.0;

    // angle from this camera's ray to the midpoint, doubled: ray to ray
    H err = tri_angle_error_small(v0, m)*2.0;

    H w0, w1, w01;
    const bool ok = tri_chirality(&w0, &w1, &w01, l0, v0, l1, v1, t01);
    if(convergent) *convergent = ok;
    if(!ok)
    {
        // divergent rays: pull towards the vanishing point, smoothly
        const H evp = tri_angle_error_small(v0, v1);
        err = err + evp*(tri_sigmoid(-w0, 3.0) + tri_sigmoid(-w1, 3.0) + tri_sigmoid(-w01, 3.0));
    }
    return err;
}

// The pair (observation 0 in camera 0, observation 1 in camera 1): the
// residual as a function of the two cameras' rt_cam_ref. Independent variables
// of the duals: 0..5 = rt of camera 0, 6..11 = rt of camera 1; a camera at the
// reference (HAS == false) has none. v0, v1: the observation vectors in their own
// cameras' coordinates. GRAD == false: the value alone (every dual a constant)
// camera 0's ray and position in camera 1's coordinates (tri_error()'s v1 and t01), with the partials each can have:
// the ray depends on camera 0's rotation and camera 1's (variables 0..2, 6..8), the position on everything.
// G0, G1: differentiate with respect to camera 0's pose (variables 0..5) / camera 1's (6..11) - both, one, or neither
template<bool G0, bool G1> struct TriPairTypes
{
    typedef DualR<G0 ? 0 : (G1 ? 6 : 0), G1 ? 9  : (G0 ? 3 : 0)> V;
    typedef DualR<G0 ? 0 : (G1 ? 6 : 0), G1 ? 12 : (G0 ? 6 : 0)> T;
};
template<bool G0, bool G1, bool HAS0, bool HAS1> MRCAL_AMD_HD
void tri_pair_geometry(typename TriPairTypes<G0,G1>::V* v0_cam1_out, typename TriPairTypes<G0,G1>::T* t_10_out,
                       const double* v0, const double* rt0, const double* rt1)
{
    typedef DualR<0,0> K;
    typedef DualR<0, (G0 && HAS0) ? 3 : 0>  R0;  typedef DualR<(G0 && HAS0) ? 3 : 0, (G0 && HAS0) ? 6  : 0> T0;
    typedef DualR<(G1 && HAS1) ? 6 : 0, (G1 && HAS1) ? 9 : 0> R1;  typedef DualR<(G1 && HAS1) ? 9 : 0, (G1 && HAS1) ? 12 : 0> T1;
    typedef typename DualHull<R0, T0>::type RT0;    // what depends on camera 0's pose
    // camera 0's ray and position in the reference frame: v0_ref = R0^T v0, t_r0 = -R0^T t0
    R0  v0_ref[3];
    RT0 t_r0[3];
    if constexpr(HAS0)
    {
        R0 r0[3]; T0 t0[3]; K v0d[3]; RT0 tmp[3];
        for(int i=0;i<3;i++)
        {
            r0[i]  = R0::variable(rt0[i],   i);
            t0[i]  = T0::variable(rt0[3+i], 3+i);
            v0d[i] = K(v0[i]);
        }
        rotate_point_r_dualr(tmp, r0, t0, true);
        for(int i=0;i<3;i++) t_r0[i] = -tmp[i];
        rotate_point_r_dualr(v0_ref, r0, v0d, true);
    }
    else
        for(int i=0;i<3;i++) { v0_ref[i] = R0(v0[i]); t_r0[i] = RT0(0.0); }

    // both in camera 1: v0_cam1 = R1 v0_ref, t_10 = R1 t_r0 + t1
    typedef typename TriPairTypes<G0,G1>::V V;
    typedef typename TriPairTypes<G0,G1>::T T;
    if constexpr(HAS1)
    {
        R1 r1[3];
        for(int i=0;i<3;i++) r1[i] = R1::variable(rt1[i], 6+i);
        typename DualHull<R1, R0>::type rv[3];
        rotate_point_r_dualr(rv, r1, v0_ref, false);
        for(int i=0;i<3;i++) v0_cam1_out[i] = V(rv[i]);
        if constexpr(HAS0)
        {
            typename DualHull<R1, RT0>::type rot[3];
            rotate_point_r_dualr(rot, r1, t_r0, false);
            for(int i=0;i<3;i++) t_10_out[i] = T(rot[i] + T1::variable(rt1[3+i], 9+i));
        }
        else
            for(int i=0;i<3;i++) t_10_out[i] = T(T1::variable(rt1[3+i], 9+i));
    }
    else
        for(int i=0;i<3;i++) { v0_cam1_out[i] = V(v0_ref[i]); t_10_out[i] = T(t_r0[i]); }
}
// The pair (observation 0 in camera 0, observation 1 in camera 1): the
// residual as a function of the two cameras' rt_cam_ref. Independent variables
// of the duals: 0..5 = rt of camera 0, 6..11 = rt of camera 1; a camera at the
// reference (rt == NULL) has none. v0, v1: the observation vectors in their own
// cameras' coordinates. G0 / G1: with the partials with respect to camera 0's / camera 1's pose (de[0..5] / de[6..11];
// the others are not touched).
// The geometry - up to four rotations of a point, by which of the two cameras have poses - is the part that differs from
// pair to pair; the error function behind it is ONE piece of code for all of them (the lanes of a wave that hold pairs
// of different kinds part ways for the rotations only)
template<bool G0, bool G1> MRCAL_AMD_HD
double tri_pair_error_partials(double* de /* [12] */, const double* v0, const double* v1, const double* rt0, const double* rt1, bool* convergent)
{
    typename TriPairTypes<G0,G1>::V v0_cam1[3];
    typename TriPairTypes<G0,G1>::T t_10[3];
    if(rt0 != NULL)
    {
        if(rt1 != NULL) tri_pair_geometry<G0, G1, true,  true >(v0_cam1, t_10, v0, rt0, rt1);
        else            tri_pair_geometry<G0, G1, true,  false>(v0_cam1, t_10, v0, rt0, rt1);
    }
    else
    {
        if(rt1 != NULL) tri_pair_geometry<G0, G1, false, true >(v0_cam1, t_10, v0, rt0, rt1);
        else            tri_pair_geometry<G0, G1, false, false>(v0_cam1, t_10, v0, rt0, rt1);
    }
    const auto er = tri_error(v1, v0_cam1, t_10, convergent);
    if constexpr(G0) for(int i=0;i<6;i++)  de[i] = er.partial(i);
    if constexpr(G1) for(int i=6;i<12;i++) de[i] = er.partial(i);
    return er.x;
}
// ... as one call: N = 12 (value and the twelve partials) or 0 (the value)
template<int N> MRCAL_AMD_HD
Dual<N> tri_pair_error(const double* v0, const double* v1, const double* rt0, const double* rt1, bool* convergent)
{
    static_assert(N == 0 || N == 12, "the pair's residual: alone, or with all twelve partials");
    Dual<N> e;
    double de[12];
    e.x = tri_pair_error_partials<N == 12, N == 12>(de, v0, v1, rt0, rt1, convergent);
    for(int i=0;i<N;i++) e.d[i] = de[i];
    return e;
}

} // namespace mrcal_amd
