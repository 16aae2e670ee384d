// Drop-in tier, integer/layout part: lens model names, state and measurement
// indexing, Nnz, pack/unpack of state vectors. Pure host code; runs without a
// GPU. Each function cites the reference function it replaces in
// include/mrcal_amd.h.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "layout.hpp"
#include "host_state.hpp"

using namespace mrcal_amd;

static Layout layout_from_args(int Ncameras_intrinsics, int Ncameras_extrinsics,
                               int Nframes,
                               int Npoints, int Npoints_fixed,
                               int Nobservations_board, int Nobservations_point,
                               int width_n, int height_n,
                               const mrcal_observation_point_triangulated_t* obs_tri, int Nobs_tri,
                               mrcal_problem_selections_t sel,
                               const mrcal_lensmodel_t* lensmodel)
{
    Dims d;
    d.Ncameras_intrinsics = Ncameras_intrinsics;
    d.Ncameras_extrinsics = Ncameras_extrinsics;
    d.Nframes             = Nframes;
    d.Npoints             = Npoints;
    d.Npoints_fixed       = Npoints_fixed;
    d.Nobservations_board = Nobservations_board;
    d.Nobservations_point = Nobservations_point;
    d.object_width_n      = width_n;
    d.object_height_n     = height_n;
    return make_layout(d, sel, *lensmodel, obs_tri, Nobs_tri);
}

extern "C" {

////////////////////////////////////////////////////////////////////////////////
// lens model names
////////////////////////////////////////////////////////////////////////////////
static const struct { const char* name; mrcal_lensmodel_type_t type; } simple_models[] =
{
    { "LENSMODEL_PINHOLE",       MRCAL_LENSMODEL_PINHOLE       },
    { "LENSMODEL_STEREOGRAPHIC", MRCAL_LENSMODEL_STEREOGRAPHIC },
    { "LENSMODEL_LONLAT",        MRCAL_LENSMODEL_LONLAT        },
    { "LENSMODEL_LATLON",        MRCAL_LENSMODEL_LATLON        },
    { "LENSMODEL_OPENCV4",       MRCAL_LENSMODEL_OPENCV4       },
    { "LENSMODEL_OPENCV5",       MRCAL_LENSMODEL_OPENCV5       },
    { "LENSMODEL_OPENCV8",       MRCAL_LENSMODEL_OPENCV8       },
    { "LENSMODEL_OPENCV12",      MRCAL_LENSMODEL_OPENCV12      },
    { "LENSMODEL_CAHVOR",        MRCAL_LENSMODEL_CAHVOR        },
};

bool mrcal_lensmodel_from_name(mrcal_lensmodel_t* lensmodel, const char* name)
{
    memset(lensmodel, 0, sizeof(*lensmodel));
    lensmodel->type = MRCAL_LENSMODEL_INVALID_TYPE;
    if(name == NULL) return false;

    for(size_t i=0; i<sizeof(simple_models)/sizeof(simple_models[0]); i++)
        if(0 == strcmp(name, simple_models[i].name))
        {
            lensmodel->type = simple_models[i].type;
            return true;
        }

    // configured models: NAME_key=value_key=value...
    static const char cahvore[] = "LENSMODEL_CAHVORE";
    static const char splined[] = "LENSMODEL_SPLINED_STEREOGRAPHIC";
    const char* prefix = NULL;
    if     (0 == strncmp(name, cahvore, sizeof(cahvore)-1)) prefix = cahvore;
    else if(0 == strncmp(name, splined, sizeof(splined)-1)) prefix = splined;
    if(prefix == NULL) return false;

    const char* cfg = name + strlen(prefix);
    if(*cfg == '\0') { lensmodel->type = MRCAL_LENSMODEL_INVALID_MISSINGCONFIG; return false; }
    if(*cfg != '_')  return false;

    int pos = 0;
    if(prefix == cahvore)
    {
        double linearity;
        if(1 == sscanf(cfg, "_linearity=%lf%n", &linearity, &pos) && cfg[pos] == '\0')
        {
            lensmodel->type = MRCAL_LENSMODEL_CAHVORE;
            lensmodel->LENSMODEL_CAHVORE__config.linearity = linearity;
            return true;
        }
    }
    else
    {
        unsigned short order, Nx, Ny, fov;
        if(4 == sscanf(cfg, "_order=%hu_Nx=%hu_Ny=%hu_fov_x_deg=%hu%n", &order, &Nx, &Ny, &fov, &pos) &&
           cfg[pos] == '\0')
        {
            lensmodel->type = MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC;
            lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.order     = order;
            lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx        = Nx;
            lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny        = Ny;
            lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.fov_x_deg = fov;
            return true;
        }
    }
    lensmodel->type = MRCAL_LENSMODEL_INVALID_BADCONFIG;
    return false;
}

bool mrcal_lensmodel_name(char* out, int size, const mrcal_lensmodel_t* lensmodel)
{
    for(size_t i=0; i<sizeof(simple_models)/sizeof(simple_models[0]); i++)
        if(lensmodel->type == simple_models[i].type)
            return size > snprintf(out, size, "%s", simple_models[i].name);
    if(lensmodel->type == MRCAL_LENSMODEL_CAHVORE)
        return size > snprintf(out, size, "LENSMODEL_CAHVORE_linearity=%.2f",
                               lensmodel->LENSMODEL_CAHVORE__config.linearity);
    if(lensmodel->type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
        return size > snprintf(out, size,
                               "LENSMODEL_SPLINED_STEREOGRAPHIC_order=%hu_Nx=%hu_Ny=%hu_fov_x_deg=%hu",
                               lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.order,
                               lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx,
                               lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny,
                               lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.fov_x_deg);
    return false;
}

int mrcal_lensmodel_num_params(const mrcal_lensmodel_t* lensmodel)
{
    return lensmodel_num_params(*lensmodel);
}

////////////////////////////////////////////////////////////////////////////////
// state layout
////////////////////////////////////////////////////////////////////////////////
#define STATE_LAYOUT()                                                  \
    const Layout L = layout_from_args(Ncameras_intrinsics, Ncameras_extrinsics, Nframes, \
                                      Npoints, Npoints_fixed, Nobservations_board, 0, \
                                      0,0, NULL,0, problem_selections, lensmodel)

int mrcal_num_intrinsics_optimization_params(mrcal_problem_selections_t problem_selections,
                                             const mrcal_lensmodel_t* lensmodel)
{
    const Layout L = layout_from_args(1,0,0,0,0,0,0,0,0,NULL,0, problem_selections, lensmodel);
    return L.Nintr_state;
}

int mrcal_num_states(int Ncameras_intrinsics, int Ncameras_extrinsics,
                     int Nframes,
                     int Npoints, int Npoints_fixed, int Nobservations_board,
                     mrcal_problem_selections_t problem_selections,
                     const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    return L.Nstate;
}

int mrcal_state_index_intrinsics(int icam_intrinsics,
                                 int Ncameras_intrinsics, int Ncameras_extrinsics,
                                 int Nframes,
                                 int Npoints, int Npoints_fixed, int Nobservations_board,
                                 mrcal_problem_selections_t problem_selections,
                                 const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    if(Ncameras_intrinsics <= 0 || L.Nintr_state <= 0) return -1;
    if(icam_intrinsics < 0 || icam_intrinsics >= Ncameras_intrinsics) return -1;
    return icam_intrinsics * L.Nintr_state;
}
int mrcal_num_states_intrinsics(int Ncameras_intrinsics,
                                mrcal_problem_selections_t problem_selections,
                                const mrcal_lensmodel_t* lensmodel)
{
    return Ncameras_intrinsics * mrcal_num_intrinsics_optimization_params(problem_selections, lensmodel);
}

int mrcal_state_index_extrinsics(int icam_extrinsics,
                                 int Ncameras_intrinsics, int Ncameras_extrinsics,
                                 int Nframes,
                                 int Npoints, int Npoints_fixed, int Nobservations_board,
                                 mrcal_problem_selections_t problem_selections,
                                 const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    if(Ncameras_extrinsics <= 0 || !problem_selections.do_optimize_extrinsics) return -1;
    if(icam_extrinsics < 0 || icam_extrinsics >= Ncameras_extrinsics) return -1;
    return L.Nstate_intrinsics + 6*icam_extrinsics;
}
int mrcal_num_states_extrinsics(int Ncameras_extrinsics,
                                mrcal_problem_selections_t problem_selections)
{
    return problem_selections.do_optimize_extrinsics ? 6*Ncameras_extrinsics : 0;
}

int mrcal_state_index_frames(int iframe,
                             int Ncameras_intrinsics, int Ncameras_extrinsics,
                             int Nframes,
                             int Npoints, int Npoints_fixed, int Nobservations_board,
                             mrcal_problem_selections_t problem_selections,
                             const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    if(Nframes <= 0 || !problem_selections.do_optimize_frames) return -1;
    if(iframe < 0 || iframe >= Nframes) return -1;
    return L.Nstate_intrinsics + L.Nstate_extrinsics + 6*iframe;
}
int mrcal_num_states_frames(int Nframes,
                            mrcal_problem_selections_t problem_selections)
{
    return problem_selections.do_optimize_frames ? 6*Nframes : 0;
}

int mrcal_state_index_points(int i_point,
                             int Ncameras_intrinsics, int Ncameras_extrinsics,
                             int Nframes,
                             int Npoints, int Npoints_fixed, int Nobservations_board,
                             mrcal_problem_selections_t problem_selections,
                             const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    const int Nvariable = Npoints - Npoints_fixed;
    if(Nvariable <= 0 || !problem_selections.do_optimize_frames) return -1;
    if(i_point < 0 || i_point >= Nvariable) return -1;
    return L.Nstate_intrinsics + L.Nstate_extrinsics + L.Nstate_frames + 3*i_point;
}
int mrcal_num_states_points(int Npoints, int Npoints_fixed,
                            mrcal_problem_selections_t problem_selections)
{
    return problem_selections.do_optimize_frames ? 3*(Npoints - Npoints_fixed) : 0;
}

int mrcal_state_index_calobject_warp(int Ncameras_intrinsics, int Ncameras_extrinsics,
                                     int Nframes,
                                     int Npoints, int Npoints_fixed, int Nobservations_board,
                                     mrcal_problem_selections_t problem_selections,
                                     const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    if(!L.has_warp) return -1;
    return L.Nstate_intrinsics + L.Nstate_extrinsics + L.Nstate_frames + L.Nstate_points;
}
int mrcal_num_states_calobject_warp(mrcal_problem_selections_t problem_selections,
                                    int Nobservations_board)
{
    return (problem_selections.do_optimize_calobject_warp && Nobservations_board > 0) ? 2 : 0;
}

void mrcal_pack_solver_state_vector(double* b,
                                    int Ncameras_intrinsics, int Ncameras_extrinsics,
                                    int Nframes,
                                    int Npoints, int Npoints_fixed, int Nobservations_board,
                                    mrcal_problem_selections_t problem_selections,
                                    const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    for(int i=0; i<L.Nstate; i++) b[i] /= state_scale(L, i);
}
void mrcal_unpack_solver_state_vector(double* b,
                                      int Ncameras_intrinsics, int Ncameras_extrinsics,
                                      int Nframes,
                                      int Npoints, int Npoints_fixed, int Nobservations_board,
                                      mrcal_problem_selections_t problem_selections,
                                      const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    for(int i=0; i<L.Nstate; i++) b[i] *= state_scale(L, i);
}

////////////////////////////////////////////////////////////////////////////////
// measurement layout
////////////////////////////////////////////////////////////////////////////////
int mrcal_num_measurements_boards(int Nobservations_board,
                                  int calibration_object_width_n, int calibration_object_height_n)
{
    if(Nobservations_board <= 0) return 0;
    return Nobservations_board * calibration_object_width_n*calibration_object_height_n * 2;
}
int mrcal_measurement_index_boards(int i_observation_board,
                                   int Nobservations_board, int Nobservations_point,
                                   int calibration_object_width_n, int calibration_object_height_n)
{
    if(Nobservations_board <= 0) return -1;
    return mrcal_num_measurements_boards(i_observation_board,
                                         calibration_object_width_n, calibration_object_height_n);
}
int mrcal_num_measurements_points(int Nobservations_point)
{
    return Nobservations_point * 2;
}
int mrcal_measurement_index_points(int i_observation_point,
                                   int Nobservations_board, int Nobservations_point,
                                   int calibration_object_width_n, int calibration_object_height_n)
{
    if(Nobservations_point <= 0) return -1;
    return
        mrcal_num_measurements_boards(Nobservations_board,
                                      calibration_object_width_n, calibration_object_height_n) +
        2*i_observation_point;
}
int mrcal_num_measurements_points_triangulated_initial_Npoints(const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                                               int Nobservations_point_triangulated,
                                                               int Npoints)
{
    return num_measurements_triangulated_initial(observations_point_triangulated,
                                                 Nobservations_point_triangulated, Npoints);
}
int mrcal_num_measurements_points_triangulated(const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                               int Nobservations_point_triangulated)
{
    return num_measurements_triangulated_initial(observations_point_triangulated,
                                                 Nobservations_point_triangulated, -1);
}
int mrcal_measurement_index_points_triangulated(int i_point_triangulated,
                                                int Nobservations_board, int Nobservations_point,
                                                const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                                int Nobservations_point_triangulated,
                                                int calibration_object_width_n, int calibration_object_height_n)
{
    if(observations_point_triangulated == NULL || Nobservations_point_triangulated <= 0)
        return -1;
    return
        mrcal_num_measurements_boards(Nobservations_board,
                                      calibration_object_width_n, calibration_object_height_n) +
        mrcal_num_measurements_points(Nobservations_point) +
        num_measurements_triangulated_initial(observations_point_triangulated,
                                              Nobservations_point_triangulated,
                                              i_point_triangulated);
}

bool mrcal_decode_observation_indices_points_triangulated(int* iobservation0, int* iobservation1,
                                                          int* iobservation_point0,
                                                          int* Nobservations_this_point,
                                                          int* Nmeasurements_this_point,
                                                          int* ipoint,
                                                          const int imeasurement,
                                                          const mrcal_observation_point_triangulated_t* obs,
                                                          int Nobs)
{
    if(obs == NULL || Nobs <= 0) return false;
    // walk the points; point p with n observations owns n(n-1)/2 consecutive
    // rows, pairs (o0<o1) ordered by o0 then o1
    int row0 = 0, first = 0;
    *ipoint = 0;
    while(first < Nobs)
    {
        int n = 1;
        while(first+n-1 < Nobs-1 && !obs[first+n-1].last_in_set) n++;
        const int nrows = n*(n-1)/2;
        if(imeasurement < row0 + nrows)
        {
            int m = imeasurement - row0;
            int o0 = 0;
            while(m >= n-1-o0) { m -= n-1-o0; o0++; }
            *iobservation0            = first + o0;
            *iobservation1            = first + o0 + 1 + m;
            *iobservation_point0      = first;
            *Nobservations_this_point = n;
            *Nmeasurements_this_point = nrows;
            return true;
        }
        row0  += nrows;
        first += n;
        (*ipoint)++;
    }
    return false;
}

int mrcal_num_measurements_regularization(int Ncameras_intrinsics, int Ncameras_extrinsics,
                                          int Nframes,
                                          int Npoints, int Npoints_fixed, int Nobservations_board,
                                          mrcal_problem_selections_t problem_selections,
                                          const mrcal_lensmodel_t* lensmodel)
{
    STATE_LAYOUT();
    return L.Nmeas_regularization;
}
int mrcal_measurement_index_regularization(const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                           int Nobservations_point_triangulated,
                                           int calibration_object_width_n, int calibration_object_height_n,
                                           int Ncameras_intrinsics, int Ncameras_extrinsics,
                                           int Nframes,
                                           int Npoints, int Npoints_fixed, int Nobservations_board, int Nobservations_point,
                                           mrcal_problem_selections_t problem_selections,
                                           const mrcal_lensmodel_t* lensmodel)
{
    const Layout L = layout_from_args(Ncameras_intrinsics, Ncameras_extrinsics, Nframes,
                                      Npoints, Npoints_fixed, Nobservations_board, Nobservations_point,
                                      calibration_object_width_n, calibration_object_height_n,
                                      observations_point_triangulated, Nobservations_point_triangulated,
                                      problem_selections, lensmodel);
    if(L.Nmeas_regularization <= 0) return -1;
    return L.i_meas_regularization;
}
int mrcal_num_measurements(int Nobservations_board, int Nobservations_point,
                           const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                           int Nobservations_point_triangulated,
                           int calibration_object_width_n, int calibration_object_height_n,
                           int Ncameras_intrinsics, int Ncameras_extrinsics,
                           int Nframes,
                           int Npoints, int Npoints_fixed,
                           mrcal_problem_selections_t problem_selections,
                           const mrcal_lensmodel_t* lensmodel)
{
    const Layout L = layout_from_args(Ncameras_intrinsics, Ncameras_extrinsics, Nframes,
                                      Npoints, Npoints_fixed, Nobservations_board, Nobservations_point,
                                      calibration_object_width_n, calibration_object_height_n,
                                      observations_point_triangulated, Nobservations_point_triangulated,
                                      problem_selections, lensmodel);
    return L.Nmeas;
}

int _mrcal_num_j_nonzero(int Nobservations_board,
                         int Nobservations_point,
                         const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                         int Nobservations_point_triangulated,
                         int calibration_object_width_n,
                         int calibration_object_height_n,
                         int Ncameras_intrinsics, int Ncameras_extrinsics,
                         int Nframes,
                         int Npoints, int Npoints_fixed,
                         const mrcal_observation_board_t* observations_board,
                         const mrcal_observation_point_t* observations_point,
                         mrcal_problem_selections_t problem_selections,
                         const mrcal_lensmodel_t* lensmodel)
{
    const Layout L = layout_from_args(Ncameras_intrinsics, Ncameras_extrinsics, Nframes,
                                      Npoints, Npoints_fixed, Nobservations_board, Nobservations_point,
                                      calibration_object_width_n, calibration_object_height_n,
                                      observations_point_triangulated, Nobservations_point_triangulated,
                                      problem_selections, lensmodel);
    return (int)num_j_nonzero(L, observations_board, observations_point,
                              observations_point_triangulated, Nobservations_point_triangulated);
}

////////////////////////////////////////////////////////////////////////////////
// vanilla-calibration camera bookkeeping
////////////////////////////////////////////////////////////////////////////////
bool mrcal_corresponding_icam_extrinsics(int* icam_extrinsics,
                                         int icam_intrinsics,
                                         int Ncameras_intrinsics,
                                         int Ncameras_extrinsics,
                                         int Nobservations_board,
                                         const mrcal_observation_board_t* observations_board,
                                         int Nobservations_point,
                                         const mrcal_observation_point_t* observations_point)
{
    if(!(Ncameras_intrinsics == Ncameras_extrinsics ||
         Ncameras_intrinsics == Ncameras_extrinsics+1))
    {
        set_error("Cannot compute icam_extrinsics. I don't have a vanilla calibration problem (stationary cameras, cam0 is reference)");
        return false;
    }
    const int UNSEEN = -100;
    std::vector<int> i_to_e(Ncameras_intrinsics,   UNSEEN);
    std::vector<int> e_to_i(Ncameras_extrinsics+1, UNSEEN); // slot 0 is "the reference"

    auto check = [&](const mrcal_camera_index_t& icam, int i, const char* what) -> bool
    {
        const int ii = icam.intrinsics;
        const int ie = icam.extrinsics < 0 ? -1 : icam.extrinsics;
        if(e_to_i[ie+1] == UNSEEN) e_to_i[ie+1] = ii;
        if(i_to_e[ii]   == UNSEEN) i_to_e[ii]   = ie;
        if(e_to_i[ie+1] != ii || i_to_e[ii] != ie)
        {
            set_error("Cannot compute icam_extrinsics. I don't have a vanilla calibration problem: %s observation %d has icam_intrinsics,icam_extrinsics %d,%d, inconsistent with an earlier observation",
                      what, i, ii, ie);
            return false;
        }
        return true;
    };
    for(int i=0; i<Nobservations_board; i++)
        if(!check(observations_board[i].icam, i, "board")) return false;
    for(int i=0; i<Nobservations_point; i++)
        if(!check(observations_point[i].icam, i, "point")) return false;

    *icam_extrinsics = i_to_e[icam_intrinsics];
    return true;
}

} // extern "C"
