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
// map. __host__ __device__: the host build is used by the outlier logic
// (solver.cpp) and by the CPU tests.
#pragma once
#include "device_math.hpp"

namespace mrcal_amd {

template<int N> MRCAL_AMD_HD Dual<N> tri_cross_norm2(const Dual<N>* a, const Dual<N>* b)
{
    const Dual<N> c0 = a[1]*b[2] - a[2]*b[1];
    const Dual<N> c1 = a[2]*b[0] - a[0]*b[2];
    const Dual<N> c2 = a[0]*b[1] - a[1]*b[0];
    return c0*c0 + c1*c1 + c2*c2;
}

// small angle between two vectors: sqrt(2 (1 - |cos|)); exactly 0 (and flat)
// below 1e-21
template<int N> MRCAL_AMD_HD Dual<N> tri_angle_error_small(const Dual<N>* v0, const Dual<N>* v1)
{
    const Dual<N> i00 = v0[0]*v0[0] + v0[1]*v0[1] + v0[2]*v0[2];
    const Dual<N> i11 = v1[0]*v1[0] + v1[1]*v1[1] + v1[2]*v1[2];
    const Dual<N> i01 = v0[0]*v1[0] + v0[1]*v1[1] + v0[2]*v1[2];
    Dual<N> costh = i01/dsqrt(i00*i11);
    if(costh.x < 0.0) costh = -costh;           // barely-divergent rays
    Dual<N> th_sq = costh*(-2.0) + 2.0;
    if(th_sq.x < 1e-21) return Dual<N>(0.0);
    return dsqrt(th_sq);
}

// 0 below 0, 1 above knee, two parabolas in between
template<int N> MRCAL_AMD_HD Dual<N> tri_sigmoid(const Dual<N>& x, double knee)
{
    if(x.x <= 0.0)  return Dual<N>(0.0);
    if(knee <= x.x) return Dual<N>(1.0);
    const double bq = 2./knee, c = 1./2.;
    const double a = (x.x < knee/2.0) ? 2./knee/knee : -2./knee/knee;
    const Dual<N> dx = x - knee/2.;
    return dx*(dx*a + bq) + c;
}

// The points l0 v0 and t01 + l1 v1 should coincide. Would flipping the sign of
// l0, l1 or both bring them closer? worsening* = how much farther apart a flip
// puts them; all three positive: the signs are right
template<int N> MRCAL_AMD_HD
bool tri_chirality(Dual<N>* w0, Dual<N>* w1, Dual<N>* w01,
                   const Dual<N>& l0, const Dual<N>* v0, const Dual<N>& l1, const Dual<N>* v1, const Dual<N>* t01)
{
    *w0 = Dual<N>(0.0); *w1 = Dual<N>(0.0); *w01 = Dual<N>(0.0);
    for(int i=0;i<3;i++)
    {
        const Dual<N> xn  = ( l1*v1[i] + t01[i]) - l0*v0[i];
        const Dual<N> x0  = ( l1*v1[i] + t01[i]) + l0*v0[i];
        const Dual<N> x1  = (-(l1*v1[i]) + t01[i]) - l0*v0[i];
        const Dual<N> x01 = (-(l1*v1[i]) + t01[i]) + l0*v0[i];
        *w0  = *w0  + (x0 *x0  - xn*xn);
        *w1  = *w1  + (x1 *x1  - xn*xn);
        *w01 = *w01 + (x01*x01 - xn*xn);
    }
    return w0->x > 0.0 && w1->x > 0.0 && w01->x > 0.0;
}

// v0: the observation vector of the camera we are in (no derivatives);
// v1: the other camera's observation vector, rotated into this camera;
// t01: the other camera's position in this camera's coordinates
template<int N> MRCAL_AMD_HD
Dual<N> tri_error(const double* v0_in, const Dual<N>* v1, const Dual<N>* t01, bool* convergent)
{
    const Dual<N> v0[3] = { Dual<N>(v0_in[0]), Dual<N>(v0_in[1]), Dual<N>(v0_in[2]) };
    const Dual<N> pr = Dual<N>(1.0)/tri_cross_norm2<N>(v0, v1);
    const Dual<N> l0 = dsqrt(tri_cross_norm2<N>(v1, t01)*pr);
    const Dual<N> l1 = dsqrt(tri_cross_norm2<N>(v0, t01)*pr);
    Dual<N> m[3];
    for(int i=0;i<3;i++) m[i] = (v0[i]*l0 + t01[i] + v1[i]*l1)/2.0;

    // angle from this camera's ray to the midpoint, doubled: ray to ray
    Dual<N> err = tri_angle_error_small<N>(v0, m)*2.0;

    Dual<N> w0, w1, w01;
    const bool ok = tri_chirality<N>(&w0, &w1, &w01, l0, v0, l1, v1, t01);
    if(convergent) *convergent = ok;
    if(!ok)
    {
        // divergent rays: pull towards the vanishing point, smoothly
        const Dual<N> evp = tri_angle_error_small<N>(v0, v1);
        err = err + evp*(tri_sigmoid<N>(-w0, 3.0) + tri_sigmoid<N>(-w1, 3.0) + tri_sigmoid<N>(-w01, 3.0));
    }
    return err;
}

// The pair (observation 0 in camera 0, observation 1 in camera 1): the
// residual as a function of the two cameras' rt_cam_ref. Independent variables
// of the duals: 0..5 = rt of camera 0, 6..11 = rt of camera 1; a camera at the
// reference (rt == NULL) has none. v0, v1: the observation vectors in their own
// cameras' coordinates
template<int N> MRCAL_AMD_HD
Dual<N> tri_pair_error(const double* v0, const double* v1, const double* rt0, const double* rt1, bool* convergent)
{
    // camera 0's ray and position in the reference frame: v0_ref = R0^T v0, t_r0 = -R0^T t0
    Dual<N> v0_ref[3], t_r0[3];
    if(rt0 != NULL)
    {
        Dual<N> r0[3], t0[3], v0d[3], tmp[3];
        for(int i=0;i<3;i++)
        {
            r0[i]  = Dual<N>::variable(rt0[i],   i);
            t0[i]  = Dual<N>::variable(rt0[3+i], 3+i);
            v0d[i] = Dual<N>(v0[i]);
        }
        rotate_point_r_dual<N>(tmp, r0, t0, true);
        for(int i=0;i<3;i++) t_r0[i] = -tmp[i];
        rotate_point_r_dual<N>(v0_ref, r0, v0d, true);
    }
    else
        for(int i=0;i<3;i++) { v0_ref[i] = Dual<N>(v0[i]); t_r0[i] = Dual<N>(0.0); }

    // both in camera 1: v0_cam1 = R1 v0_ref, t_10 = R1 t_r0 + t1
    Dual<N> v0_cam1[3], t_10[3];
    if(rt1 != NULL)
    {
        Dual<N> r1[3];
        for(int i=0;i<3;i++) r1[i] = Dual<N>::variable(rt1[i], 6+i);
        rotate_point_r_dual<N>(v0_cam1, r1, v0_ref, false);
        if(rt0 != NULL)
        {
            rotate_point_r_dual<N>(t_10, r1, t_r0, false);
            for(int i=0;i<3;i++) t_10[i] = t_10[i] + Dual<N>::variable(rt1[3+i], 9+i);
        }
        else
            for(int i=0;i<3;i++) t_10[i] = Dual<N>::variable(rt1[3+i], 9+i);
    }
    else
        for(int i=0;i<3;i++) { v0_cam1[i] = v0_ref[i]; t_10[i] = t_r0[i]; }

    return tri_error<N>(v1, v0_cam1, t_10, convergent);
}

} // namespace mrcal_amd
