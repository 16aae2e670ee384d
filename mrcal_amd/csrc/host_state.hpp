// Host-side packing/unpacking of the state vector and small shared helpers.
// Reference semantics: mrcal.c:3308-3735 (pack divides by the scale, unpack
// multiplies; only the blocks selected for optimization appear in the state).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include "layout.hpp"

namespace mrcal_amd {

// last error of this thread, also echoed to stderr the way the reference's
// MSG() does (_util.h)
inline std::string& last_error_string()
{
    static thread_local std::string s;
    return s;
}
inline void set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_string() = buf;
    fprintf(stderr, "mrcal_amd: %s\n", buf);
}

// unpacked arrays -> packed state
inline void pack_state_from_arrays(double* b, const Layout& L,
                                   const double* intrinsics,
                                   const mrcal_pose_t* rt_cam_ref,
                                   const mrcal_pose_t* rt_ref_frame,
                                   const mrcal_point3_t* points,
                                   const mrcal_calobject_warp_t* calobject_warp)
{
    const Dims& d = L.dims;
    int i = 0;
    for(int icam=0; icam<d.Ncameras_intrinsics; icam++)
    {
        const double* in = &intrinsics[icam*L.Nintrinsics];
        if(L.Ncore_state)
        {
            b[i++] = in[0] / SCALE_INTRINSICS_FOCAL_LENGTH;
            b[i++] = in[1] / SCALE_INTRINSICS_FOCAL_LENGTH;
            b[i++] = in[2] / SCALE_INTRINSICS_CENTER_PIXEL;
            b[i++] = in[3] / SCALE_INTRINSICS_CENTER_PIXEL;
        }
        for(int k=0; k<L.Ndist_state; k++)
            b[i++] = in[L.Ncore + k] / SCALE_DISTORTION;
    }
    if(L.Nstate_extrinsics)
        for(int icam=0; icam<d.Ncameras_extrinsics; icam++)
        {
            for(int k=0;k<3;k++) b[i++] = rt_cam_ref[icam].r.xyz[k] / SCALE_ROTATION_CAMERA;
            for(int k=0;k<3;k++) b[i++] = rt_cam_ref[icam].t.xyz[k] / SCALE_TRANSLATION_CAMERA;
        }
    if(L.sel.do_optimize_frames)
    {
        for(int iframe=0; iframe<d.Nframes; iframe++)
        {
            for(int k=0;k<3;k++) b[i++] = rt_ref_frame[iframe].r.xyz[k] / SCALE_ROTATION_FRAME;
            for(int k=0;k<3;k++) b[i++] = rt_ref_frame[iframe].t.xyz[k] / SCALE_TRANSLATION_FRAME;
        }
        for(int ip=0; ip<d.Npoints - d.Npoints_fixed; ip++)
            for(int k=0;k<3;k++) b[i++] = points[ip].xyz[k] / SCALE_POSITION_POINT;
    }
    if(L.has_warp)
    {
        b[i++] = calobject_warp->x2 / SCALE_CALOBJECT_WARP;
        b[i++] = calobject_warp->y2 / SCALE_CALOBJECT_WARP;
    }
}

// packed state -> unpacked arrays; blocks that are not in the state are left
// untouched
inline void unpack_state_to_arrays(const double* b, const Layout& L,
                                   double* intrinsics,
                                   mrcal_pose_t* rt_cam_ref,
                                   mrcal_pose_t* rt_ref_frame,
                                   mrcal_point3_t* points,
                                   mrcal_calobject_warp_t* calobject_warp)
{
    const Dims& d = L.dims;
    int i = 0;
    for(int icam=0; icam<d.Ncameras_intrinsics; icam++)
    {
        double* in = &intrinsics[icam*L.Nintrinsics];
        if(L.Ncore_state)
        {
            in[0] = b[i++] * SCALE_INTRINSICS_FOCAL_LENGTH;
            in[1] = b[i++] * SCALE_INTRINSICS_FOCAL_LENGTH;
            in[2] = b[i++] * SCALE_INTRINSICS_CENTER_PIXEL;
            in[3] = b[i++] * SCALE_INTRINSICS_CENTER_PIXEL;
        }
        for(int k=0; k<L.Ndist_state; k++)
            in[L.Ncore + k] = b[i++] * SCALE_DISTORTION;
    }
    if(L.Nstate_extrinsics)
        for(int icam=0; icam<d.Ncameras_extrinsics; icam++)
        {
            for(int k=0;k<3;k++) rt_cam_ref[icam].r.xyz[k] = b[i++] * SCALE_ROTATION_CAMERA;
            for(int k=0;k<3;k++) rt_cam_ref[icam].t.xyz[k] = b[i++] * SCALE_TRANSLATION_CAMERA;
        }
    if(L.sel.do_optimize_frames)
    {
        for(int iframe=0; iframe<d.Nframes; iframe++)
        {
            for(int k=0;k<3;k++) rt_ref_frame[iframe].r.xyz[k] = b[i++] * SCALE_ROTATION_FRAME;
            for(int k=0;k<3;k++) rt_ref_frame[iframe].t.xyz[k] = b[i++] * SCALE_TRANSLATION_FRAME;
        }
        for(int ip=0; ip<d.Npoints - d.Npoints_fixed; ip++)
            for(int k=0;k<3;k++) points[ip].xyz[k] = b[i++] * SCALE_POSITION_POINT;
    }
    if(L.has_warp)
    {
        calobject_warp->x2 = b[i++] * SCALE_CALOBJECT_WARP;
        calobject_warp->y2 = b[i++] * SCALE_CALOBJECT_WARP;
    }
}

} // namespace mrcal_amd
