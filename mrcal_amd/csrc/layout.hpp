// State-vector / measurement-vector / CSR layout of a calibration problem.
//
// Host-side integer bookkeeping; everything here must agree BIT-EXACTLY with
// the reference (mrcal.c:337-882 measurement layout and Nnz, mrcal.c:3737-3880
// state layout). The layout is computed ONCE per problem into a flat Layout
// record that both the host code and the kernels consume, instead of being
// re-derived per observation as the reference does.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include "../../include/mrcal_amd.h"

namespace mrcal_amd {

// packed = unpacked / scale. Reference: scales.h:40-48
constexpr double SCALE_INTRINSICS_FOCAL_LENGTH = 500.0;
constexpr double SCALE_INTRINSICS_CENTER_PIXEL = 20.0;
constexpr double SCALE_ROTATION_CAMERA         = 0.1 * M_PI/180.0;
constexpr double SCALE_TRANSLATION_CAMERA      = 1.0;
constexpr double SCALE_ROTATION_FRAME          = 15.0 * M_PI/180.0;
constexpr double SCALE_TRANSLATION_FRAME       = 1.0;
constexpr double SCALE_POSITION_POINT          = SCALE_TRANSLATION_FRAME;
constexpr double SCALE_CALOBJECT_WARP          = 0.01;
constexpr double SCALE_DISTORTION              = 1.0;

inline bool lensmodel_is_opencv(mrcal_lensmodel_type_t t)
{
    return t >= MRCAL_LENSMODEL_OPENCV4 && t <= MRCAL_LENSMODEL_OPENCV12;
}

// Parameter count of a lens model; -1 if the type is unknown
inline int lensmodel_num_params(const mrcal_lensmodel_t& m)
{
    switch(m.type)
    {
    case MRCAL_LENSMODEL_PINHOLE:
    case MRCAL_LENSMODEL_STEREOGRAPHIC:
    case MRCAL_LENSMODEL_LONLAT:
    case MRCAL_LENSMODEL_LATLON:   return 4;
    case MRCAL_LENSMODEL_OPENCV4:  return 8;
    case MRCAL_LENSMODEL_OPENCV5:  return 9;
    case MRCAL_LENSMODEL_OPENCV8:  return 12;
    case MRCAL_LENSMODEL_OPENCV12: return 16;
    case MRCAL_LENSMODEL_CAHVOR:   return 9;
    case MRCAL_LENSMODEL_CAHVORE:  return 12;
    case MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC:
        return 4 +
            2 * (int)m.LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx *
                (int)m.LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny;
    default: return -1;
    }
}

// Every model the reference supports today has an fx,fy,cx,cy core
// (mrcal.c:255-288), but the layout logic is written against this predicate
inline bool lensmodel_has_core(const mrcal_lensmodel_t& m)
{
    return lensmodel_num_params(m) >= 4;
}

struct Dims
{
    int Ncameras_intrinsics, Ncameras_extrinsics, Nframes;
    int Npoints, Npoints_fixed;
    int Nobservations_board, Nobservations_point;
    int object_width_n, object_height_n;
};

// All the derived counts. "state" sizes are in state variables, "meas" sizes
// in measurement rows
struct Layout
{
    mrcal_lensmodel_t          lensmodel;
    mrcal_problem_selections_t sel;
    Dims                       dims;

    int Nintrinsics;        // all lens parameters of one camera
    int Ncore;              // 4 if the model has a core
    int Ncore_state;        // 4 if the core is being optimized
    int Ndist;              // Nintrinsics - Ncore
    int Ndist_state;        // Ndist if distortions are being optimized
    int Nintr_state;        // Ncore_state + Ndist_state, per camera
    bool has_warp;          // calobject_warp is in the state

    // first state index of each block; -1 if the block is not in the state
    int i_state_intrinsics, i_state_extrinsics, i_state_frames,
        i_state_points, i_state_warp;
    int Nstate_intrinsics, Nstate_extrinsics, Nstate_frames,
        Nstate_points, Nstate_warp;
    int Nstate;

    // intrinsics columns in one board/point row
    int Nintr_per_row;
    // splined models: side of the (order+1)x(order+1) patch; 0 otherwise
    int spline_runlen;

    int Nmeas_boards, Nmeas_points, Nmeas_triangulated, Nmeas_regularization;
    int i_meas_boards, i_meas_points, i_meas_triangulated, i_meas_regularization;
    int Nmeas;

    int Nreg_percamera;
    bool has_unity_cam01;
};

inline int num_measurements_triangulated_initial(const mrcal_observation_point_triangulated_t* obs,
                                                 int Nobs, int Npoints_limit)
{
    if(obs == NULL || Nobs <= 0) return 0;
    // A point seen by n cameras contributes n(n-1)/2 rows
    int Nmeas = 0, ipoint = 0, i = 0;
    while(i < Nobs && (Npoints_limit < 0 || ipoint < Npoints_limit))
    {
        int n = 1;
        while(i < Nobs-1 && !obs[i].last_in_set) { i++; n++; }
        Nmeas += n*(n-1)/2;
        i++;
        ipoint++;
    }
    return Nmeas;
}

// applies the adjustments the reference applies at the top of mrcal_optimize()
// and mrcal_optimizer_callback() (mrcal.c:6060-6065, 6249-6278)
inline mrcal_problem_selections_t
effective_selections(mrcal_problem_selections_t sel, const mrcal_lensmodel_t& lensmodel,
                     int Nobservations_board)
{
    if(Nobservations_board <= 0)         sel.do_optimize_calobject_warp  = false;
    if(!lensmodel_has_core(lensmodel))   sel.do_optimize_intrinsics_core = false;
    return sel;
}

inline Layout make_layout(const Dims& d,
                          mrcal_problem_selections_t sel,
                          const mrcal_lensmodel_t& lensmodel,
                          const mrcal_observation_point_triangulated_t* obs_triangulated,
                          int Nobs_triangulated)
{
    Layout L;
    memset(&L, 0, sizeof(L));
    L.lensmodel = lensmodel;
    L.sel       = sel;
    L.dims      = d;

    L.Nintrinsics = lensmodel_num_params(lensmodel);
    L.Ncore       = lensmodel_has_core(lensmodel) ? 4 : 0;
    L.Ndist       = L.Nintrinsics - L.Ncore;
    L.Ncore_state = (L.Ncore && sel.do_optimize_intrinsics_core) ? 4 : 0;
    L.Ndist_state = sel.do_optimize_intrinsics_distortions ? L.Ndist : 0;
    L.Nintr_state = L.Ncore_state + L.Ndist_state;
    L.has_warp    = sel.do_optimize_calobject_warp && d.Nobservations_board > 0;

    L.Nstate_intrinsics = d.Ncameras_intrinsics * L.Nintr_state;
    L.Nstate_extrinsics = sel.do_optimize_extrinsics ? 6*d.Ncameras_extrinsics : 0;
    L.Nstate_frames     = sel.do_optimize_frames     ? 6*d.Nframes             : 0;
    // points ride on the do_optimize_frames flag
    L.Nstate_points     = sel.do_optimize_frames     ? 3*(d.Npoints - d.Npoints_fixed) : 0;
    L.Nstate_warp       = L.has_warp ? 2 : 0;

    int i = 0;
    L.i_state_intrinsics = (L.Nstate_intrinsics > 0) ? i : -1;   i += L.Nstate_intrinsics;
    L.i_state_extrinsics = (L.Nstate_extrinsics > 0) ? i : -1;   i += L.Nstate_extrinsics;
    L.i_state_frames     = (L.Nstate_frames     > 0) ? i : -1;   i += L.Nstate_frames;
    L.i_state_points     = (L.Nstate_points     > 0) ? i : -1;   i += L.Nstate_points;
    L.i_state_warp       = (L.Nstate_warp       > 0) ? i : -1;   i += L.Nstate_warp;
    L.Nstate = i;

    // Each row sees one of (fx,fy) and one of (cx,cy): 2 core columns, not 4.
    // Splined models touch an (order+1)^2 patch of one of the two surfaces
    if(lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        L.spline_runlen = lensmodel.LENSMODEL_SPLINED_STEREOGRAPHIC__config.order + 1;
        L.Nintr_per_row =
            (sel.do_optimize_intrinsics_core        ? 4                               : 0) +
            (sel.do_optimize_intrinsics_distortions ? L.spline_runlen*L.spline_runlen : 0);
    }
    else
        L.Nintr_per_row = L.Nintr_state;
    if(sel.do_optimize_intrinsics_core && L.Ncore)
        L.Nintr_per_row -= 2;

    L.Nmeas_boards = (d.Nobservations_board > 0) ?
        d.Nobservations_board * d.object_width_n * d.object_height_n * 2 : 0;
    L.Nmeas_points = d.Nobservations_point * 2;
    L.Nmeas_triangulated = num_measurements_triangulated_initial(obs_triangulated, Nobs_triangulated, -1);

    L.Nreg_percamera = 0;
    if(sel.do_apply_regularization)
    {
        L.Nreg_percamera = L.Ndist_state;
        if(sel.do_optimize_intrinsics_core) L.Nreg_percamera += 2;
    }
    L.has_unity_cam01 =
        sel.do_apply_regularization_unity_cam01 &&
        sel.do_optimize_extrinsics &&
        d.Ncameras_extrinsics > 0;
    L.Nmeas_regularization =
        d.Ncameras_intrinsics * L.Nreg_percamera + (L.has_unity_cam01 ? 1 : 0);

    L.i_meas_boards         = 0;
    L.i_meas_points         = L.Nmeas_boards;
    L.i_meas_triangulated   = L.i_meas_points       + L.Nmeas_points;
    L.i_meas_regularization = L.i_meas_triangulated + L.Nmeas_triangulated;
    L.Nmeas                 = L.i_meas_regularization + L.Nmeas_regularization;
    return L;
}

// columns in one row of a board observation
inline int nnz_per_board_row(const Layout& L, int icam_extrinsics)
{
    return
        L.Nintr_per_row +
        ((L.sel.do_optimize_extrinsics && icam_extrinsics >= 0) ? 6 : 0) +
        (L.sel.do_optimize_frames ? 6 : 0) +
        (L.has_warp ? 2 : 0);
}
inline int nnz_per_point_row(const Layout& L, int icam_extrinsics, int i_point)
{
    return
        L.Nintr_per_row +
        ((L.sel.do_optimize_extrinsics && icam_extrinsics >= 0) ? 6 : 0) +
        ((L.sel.do_optimize_frames && i_point < L.dims.Npoints - L.dims.Npoints_fixed) ? 3 : 0);
}

inline int64_t num_j_nonzero(const Layout& L,
                             const mrcal_observation_board_t* obs_board,
                             const mrcal_observation_point_t* obs_point,
                             const mrcal_observation_point_triangulated_t* obs_tri,
                             int Nobs_tri)
{
    const Dims& d = L.dims;
    int64_t N = 0;
    const int64_t rows_per_board = (int64_t)2 * d.object_width_n * d.object_height_n;
    for(int i=0; i<d.Nobservations_board; i++)
        N += rows_per_board * nnz_per_board_row(L, obs_board[i].icam.extrinsics);
    for(int i=0; i<d.Nobservations_point; i++)
        N += 2 * nnz_per_point_row(L, obs_point[i].icam.extrinsics, obs_point[i].i_point);

    if(obs_tri != NULL && Nobs_tri > 0)
    {
        // every pair (i0<i1) inside a point's observation set: the columns of
        // both cameras
        for(int i0=0; i0<Nobs_tri; i0++)
        {
            if(obs_tri[i0].last_in_set) continue;
            const int n0 = L.Nintr_per_row +
                ((L.sel.do_optimize_extrinsics && obs_tri[i0].icam.extrinsics >= 0) ? 6 : 0);
            for(int i1=i0+1; i1<Nobs_tri; i1++)
            {
                const int n1 = L.Nintr_per_row +
                    ((L.sel.do_optimize_extrinsics && obs_tri[i1].icam.extrinsics >= 0) ? 6 : 0);
                N += n0 + n1;
                if(obs_tri[i1].last_in_set) break;
            }
        }
    }

    // regularization: one column per row, except that each splined-model knot
    // row mixes the knot's two values
    if(L.lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        if(L.sel.do_apply_regularization)
            N += (int64_t)d.Ncameras_intrinsics *
                 (2*L.Ndist_state + (L.sel.do_optimize_intrinsics_core ? 2 : 0));
    }
    else
        N += (int64_t)d.Ncameras_intrinsics * L.Nreg_percamera;
    if(L.has_unity_cam01) N += 3;
    return N;
}

// scale of state variable i (unpacked = packed * scale)
inline double state_scale(const Layout& L, int i)
{
    if(i < L.Nstate_intrinsics)
    {
        int k = i % L.Nintr_state;
        if(k < L.Ncore_state) return (k < 2) ? SCALE_INTRINSICS_FOCAL_LENGTH : SCALE_INTRINSICS_CENTER_PIXEL;
        return SCALE_DISTORTION;
    }
    i -= L.Nstate_intrinsics;
    if(i < L.Nstate_extrinsics) return (i%6 < 3) ? SCALE_ROTATION_CAMERA : SCALE_TRANSLATION_CAMERA;
    i -= L.Nstate_extrinsics;
    if(i < L.Nstate_frames)     return (i%6 < 3) ? SCALE_ROTATION_FRAME  : SCALE_TRANSLATION_FRAME;
    i -= L.Nstate_frames;
    if(i < L.Nstate_points)     return SCALE_POSITION_POINT;
    return SCALE_CALOBJECT_WARP;
}

} // namespace mrcal_amd
