// placeholder until the dog-leg solver lands
#include "host_state.hpp"
using namespace mrcal_amd;
extern "C"
mrcal_stats_t
mrcal_optimize( double* b_packed, int buffer_size_b_packed,
                double* x,        int buffer_size_x,
                double*                 intrinsics,
                mrcal_pose_t*           rt_cam_ref,
                mrcal_pose_t*           rt_ref_frame,
                mrcal_point3_t*         points,
                mrcal_calobject_warp_t* calobject_warp,
                int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                int Npoints, int Npoints_fixed,
                const mrcal_observation_board_t* observations_board,
                const mrcal_observation_point_t* observations_point,
                int Nobservations_board,
                int Nobservations_point,
                const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                int Nobservations_point_triangulated,
                mrcal_point3_t* observations_board_pool,
                mrcal_point3_t* observations_point_pool,
                const mrcal_lensmodel_t* lensmodel,
                const int* imagersizes,
                mrcal_problem_selections_t       problem_selections,
                const mrcal_problem_constants_t* problem_constants,
                double calibration_object_spacing,
                int calibration_object_width_n,
                int calibration_object_height_n,
                bool verbose,
                bool check_gradient)
{
    set_error("mrcal_optimize(): the GPU solver is not implemented yet");
    mrcal_stats_t s = { -1.0, 0, 0 };
    return s;
}
