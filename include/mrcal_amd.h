/* mrcal_amd: MI355X-native implementation of mrcal's optimize() /
 * optimizer_callback() hot path. C ABI of libmrcal_amd.so.
 *
 * Two tiers of entry points:
 *
 * 1. DROP-IN TIER. Same names, argument order, argument meaning and error
 *    behaviour as the functions the reference's Python wrapper (and any ctypes
 *    harness) binds for this path. A process that dlopen()s libmrcal_amd.so
 *    instead of libmrcal.so and calls these gets the GPU implementation. All
 *    pointers are HOST pointers, the caller allocates every buffer, sizes are
 *    in bytes where the reference says bytes. Each declaration cites the
 *    reference interface it replaces (file:line in dkogan/mrcal).
 *
 * 2. RESIDENT TIER (mrcal_amd_*). The same computation with the problem held
 *    in HBM across calls: create a problem once, then evaluate / step / solve
 *    without any host<->device traffic beyond scalars. This is what the
 *    benchmark and the multi-GPU driver use. No reference counterpart: the
 *    reference has no device.
 *
 * No torch types, no C++ types: plain pointers and sizes only.
 */
#pragma once
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Types. Binary-compatible with the reference's (types.h, basic-geometry.h) */
/* ------------------------------------------------------------------------ */

/* reference: types.h:105-127 (mrcal_lensmodel_type_t). Same numeric values */
typedef enum
{
    MRCAL_LENSMODEL_INVALID               = -2,
    MRCAL_LENSMODEL_INVALID_BADCONFIG     = -1,
    MRCAL_LENSMODEL_INVALID_MISSINGCONFIG = -3,
    MRCAL_LENSMODEL_INVALID_TYPE          = -4,
    MRCAL_LENSMODEL_PINHOLE               = 0,
    MRCAL_LENSMODEL_STEREOGRAPHIC         = 1,
    MRCAL_LENSMODEL_LONLAT                = 2,
    MRCAL_LENSMODEL_LATLON                = 3,
    MRCAL_LENSMODEL_OPENCV4               = 4,
    MRCAL_LENSMODEL_OPENCV5               = 5,
    MRCAL_LENSMODEL_OPENCV8               = 6,
    MRCAL_LENSMODEL_OPENCV12              = 7,
    MRCAL_LENSMODEL_CAHVOR                = 8,
    MRCAL_LENSMODEL_CAHVORE               = 9,
    MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC = 10
} mrcal_lensmodel_type_t;

/* reference: types.h:131-145 (mrcal_lensmodel_t): 16 bytes, the
   configuration union sits at offset 8 */
typedef struct
{
    mrcal_lensmodel_type_t type;
    union
    {
        struct { double   linearity;               } LENSMODEL_CAHVORE__config;
        struct { uint16_t order, Nx, Ny, fov_x_deg; } LENSMODEL_SPLINED_STEREOGRAPHIC__config;
    };
} mrcal_lensmodel_t;

/* reference: basic-geometry.h:60-91 */
typedef union { struct { double x,y;   }; double xy[2];  } mrcal_point2_t;
typedef union { struct { double x,y,z; }; double xyz[3]; } mrcal_point3_t;
typedef struct { mrcal_point3_t r, t; }                    mrcal_pose_t;

/* reference: types.h:139-148 */
typedef union { struct { double x2, y2; }; double values[2]; } mrcal_calobject_warp_t;

/* reference: types.h:195-263 */
typedef struct { int intrinsics; int extrinsics; /* -1: at the reference */ } mrcal_camera_index_t;
typedef struct { mrcal_camera_index_t icam; int iframe;  } mrcal_observation_board_t;
typedef struct { mrcal_camera_index_t icam; int i_point; } mrcal_observation_point_t;
typedef struct
{
    mrcal_camera_index_t icam;
    bool                 last_in_set : 1;
    bool                 outlier     : 1;
    mrcal_point3_t       px; /* UNPROJECTED observation vector */
} mrcal_observation_point_triangulated_t;

/* reference: types.h:283-307. One byte, passed BY VALUE. Bit order matters */
typedef struct
{
    bool do_optimize_intrinsics_core         : 1;
    bool do_optimize_intrinsics_distortions  : 1;
    bool do_optimize_extrinsics              : 1;
    bool do_optimize_frames                  : 1;
    bool do_optimize_calobject_warp          : 1;
    bool do_apply_regularization             : 1;
    bool do_apply_outlier_rejection          : 1;
    bool do_apply_regularization_unity_cam01 : 1;
} mrcal_problem_selections_t;

/* reference: types.h:313-315 (empty in the reference, kept as a placeholder) */
typedef struct { char _unused; } mrcal_problem_constants_t;

/* reference: types.h:321-344 */
typedef struct
{
    double rms_reproj_error__pixels; /* <0: failure */
    int    Noutliers_board;
    int    Noutliers_triangulated_point;
} mrcal_stats_t;

/* The slice of SuiteSparse's cholmod_sparse that the reference's callback
   touches (mrcal.c:4461-4463: p, i, x only). Field order as in CHOLMOD so
   that a caller holding a real cholmod_sparse can pass it as is. Jt is
   (Nstate x Nmeasurements) compressed-column, i.e. J in CSR.
   A translation unit that also includes SuiteSparse's <cholmod.h> (or
   libdogleg's <dogleg.h>, which does) defines MRCAL_AMD_HAVE_CHOLMOD_SPARSE
   first and gets the real type */
#ifndef MRCAL_AMD_HAVE_CHOLMOD_SPARSE
struct cholmod_sparse_struct
{
    size_t nrow, ncol, nzmax;
    void *p, *i, *nz, *x, *z;
    int stype, itype, xtype, dtype, sorted, packed;
};
#endif

/* ------------------------------------------------------------------------ */
/* DROP-IN TIER                                                              */
/* ------------------------------------------------------------------------ */

/* reference: mrcal.c:165-253 / mrcal.h (lens model names) */
bool        mrcal_lensmodel_from_name(mrcal_lensmodel_t* lensmodel, const char* name);
bool        mrcal_lensmodel_name     (char* out, int size, const mrcal_lensmodel_t* lensmodel);
/* reference: mrcal.c:303-335 */
int         mrcal_lensmodel_num_params(const mrcal_lensmodel_t* lensmodel);

/* reference: mrcal.h:374-375, mrcal.c:361-371 */
int mrcal_num_intrinsics_optimization_params(mrcal_problem_selections_t problem_selections,
                                             const mrcal_lensmodel_t* lensmodel);

/* reference: mrcal.h:388-421, mrcal.c:3450-3735. In place on (Nstate,) */
void mrcal_pack_solver_state_vector  (double* b,
                                      int Ncameras_intrinsics, int Ncameras_extrinsics,
                                      int Nframes,
                                      int Npoints, int Npoints_fixed, int Nobservations_board,
                                      mrcal_problem_selections_t problem_selections,
                                      const mrcal_lensmodel_t* lensmodel);
void mrcal_unpack_solver_state_vector(double* b,
                                      int Ncameras_intrinsics, int Ncameras_extrinsics,
                                      int Nframes,
                                      int Npoints, int Npoints_fixed, int Nobservations_board,
                                      mrcal_problem_selections_t problem_selections,
                                      const mrcal_lensmodel_t* lensmodel);

/* reference: mrcal.h:437-446, mrcal.c:3929-3969 */
bool mrcal_corresponding_icam_extrinsics(int* icam_extrinsics,
                                         int icam_intrinsics,
                                         int Ncameras_intrinsics,
                                         int Ncameras_extrinsics,
                                         int Nobservations_board,
                                         const mrcal_observation_board_t* observations_board,
                                         int Nobservations_point,
                                         const mrcal_observation_point_t* observations_point);

/* reference: mrcal.h:453-521, mrcal.c:6179-6624. The whole solve: dog-leg
   iterations + outlier rejection. In/out arrays are updated in place; new
   outliers are marked by negating observations_board_pool[].z */
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
                bool check_gradient);

/* Pixel -> observation vector (not normalized). Reference: mrcal.h:177-206,
   mrcal.c:3082-3286. HOST computation, like in the reference: it is the setup
   step that turns triangulated-point pixel observations into the vectors
   mrcal_observation_point_triangulated_t carries (mrcal-pywrap.c:1388-1395) */
bool mrcal_unproject(mrcal_point3_t* out, const mrcal_point2_t* q, int N,
                     const mrcal_lensmodel_t* lensmodel, const double* intrinsics);
/* The batch form ON THE GPU, with the gradients that mrcal.unproject(get_gradients=True)
   reports (mrcal/projections.py:112-395, which derives them from project()'s at the
   solution): dv_dq (N,3,2), dv_dintrinsics (N,3,Nintrinsics), either may be NULL
   (dv_dintrinsics needs dv_dq). Without gradients v is what mrcal_unproject() gives;
   with them it is, like the reference's, the stereographic representative of the
   same direction for the models inverted iteratively. normalize: unit vectors, and
   the gradients of the unit vectors */
bool mrcal_amd_unproject(mrcal_point3_t* v, double* dv_dq, double* dv_dintrinsics,
                         const mrcal_point2_t* q, int N,
                         const mrcal_lensmodel_t* lensmodel, const double* intrinsics, bool normalize);


/* Stand-alone projection of N camera-frame points (reference: mrcal.h:165-174,
   mrcal.c:2867-3069). Host pointers. dq_dp (N,2,3) and dq_dintrinsics
   (N,2,Nintrinsics) may be NULL. Runs the same device functions as the solver's
   kernels, one lane per point */
bool mrcal_project(mrcal_point2_t* q, mrcal_point3_t* dq_dp, double* dq_dintrinsics,
                   const mrcal_point3_t* p, int N,
                   const mrcal_lensmodel_t* lensmodel, const double* intrinsics);

/* reference: mrcal.h:539-609, mrcal.c:5972-6177. One evaluation of the cost
   function: b_packed, x and (if Jt != NULL) the CSR Jacobian: Jt->p
   (int32[Nmeas+1]), Jt->i (int32[Nnz]), Jt->x (double[Nnz]) */
bool mrcal_optimizer_callback(double* b_packed, int buffer_size_b_packed,
                              double* x,        int buffer_size_x,
                              struct cholmod_sparse_struct* Jt,
                              const double*                 intrinsics,
                              const mrcal_pose_t*           rt_cam_ref,
                              const mrcal_pose_t*           rt_ref_frame,
                              const mrcal_point3_t*         points,
                              const mrcal_calobject_warp_t* calobject_warp,
                              int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                              int Npoints, int Npoints_fixed,
                              const mrcal_observation_board_t* observations_board,
                              const mrcal_observation_point_t* observations_point,
                              int Nobservations_board,
                              int Nobservations_point,
                              const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                              int Nobservations_point_triangulated,
                              const mrcal_point3_t* observations_board_pool,
                              const mrcal_point3_t* observations_point_pool,
                              const mrcal_lensmodel_t* lensmodel,
                              const int* imagersizes,
                              mrcal_problem_selections_t       problem_selections,
                              const mrcal_problem_constants_t* problem_constants,
                              double calibration_object_spacing,
                              int calibration_object_width_n,
                              int calibration_object_height_n,
                              bool verbose);

/* reference: mrcal.h:613-669, uncertainty.c:798-1541 (Python: mrcal.drt_cross_reprojection__dbpacked(),
   mrcal-pywrap.c:2012-2110). K_packed = drt_ref_refperturbed/db_packed (icam_intrinsics < 0: "rrp") or
   drt_cam_camperturbed/db_packed for that camera ("ccp"), from the packed Jacobian Jt as
   mrcal_optimizer_callback() fills it and the packed state it was evaluated at. Each Kpacked* is (6, N of that
   block), rows stored densely (stride1 == sizeof(double); strides in bytes, <= 0: contiguous); NULL where the
   block is not in the state. The sums over the rows of J run on the GPU (csrc/uncertainty.hip) */
bool _mrcal_drt_cross_reprojection__dbpacked(double* Kpackede,  int Kpackede_stride0,  int Kpackede_stride1,
                                             double* Kpackedf,  int Kpackedf_stride0,  int Kpackedf_stride1,
                                             double* Kpackedp,  int Kpackedp_stride0,  int Kpackedp_stride1,
                                             double* Kpackedcw, int Kpackedcw_stride0, int Kpackedcw_stride1,
                                             const int icam_intrinsics,
                                             const double* b_packed, int buffer_size_b_packed,
                                             struct cholmod_sparse_struct* Jt,
                                             int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                                             int Npoints, int Npoints_fixed,
                                             int Nobservations_board, int Nobservations_point,
                                             const mrcal_lensmodel_t* lensmodel,
                                             mrcal_problem_selections_t problem_selections,
                                             int calibration_object_width_n, int calibration_object_height_n);

/* reference: mrcal.h:713-853 (layout of the measurement and state vectors),
   mrcal.c:337-735, 3737-3880 */
int mrcal_measurement_index_boards(int i_observation_board,
                                   int Nobservations_board, int Nobservations_point,
                                   int calibration_object_width_n, int calibration_object_height_n);
int mrcal_num_measurements_boards(int Nobservations_board,
                                  int calibration_object_width_n, int calibration_object_height_n);
int mrcal_measurement_index_points(int i_observation_point,
                                   int Nobservations_board, int Nobservations_point,
                                   int calibration_object_width_n, int calibration_object_height_n);
int mrcal_num_measurements_points(int Nobservations_point);
int mrcal_measurement_index_points_triangulated(int i_point_triangulated,
                                                int Nobservations_board, int Nobservations_point,
                                                const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                                int Nobservations_point_triangulated,
                                                int calibration_object_width_n, int calibration_object_height_n);
int mrcal_num_measurements_points_triangulated(const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                               int Nobservations_point_triangulated);
int mrcal_num_measurements_points_triangulated_initial_Npoints(const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                                               int Nobservations_point_triangulated,
                                                               int Npoints);
bool mrcal_decode_observation_indices_points_triangulated(int* iobservation0, int* iobservation1,
                                                          int* iobservation_point0,
                                                          int* Nobservations_this_point,
                                                          int* Nmeasurements_this_point,
                                                          int* ipoint,
                                                          const int imeasurement,
                                                          const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                                          int Nobservations_point_triangulated);
int mrcal_measurement_index_regularization(const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                                           int Nobservations_point_triangulated,
                                           int calibration_object_width_n, int calibration_object_height_n,
                                           int Ncameras_intrinsics, int Ncameras_extrinsics,
                                           int Nframes,
                                           int Npoints, int Npoints_fixed, int Nobservations_board, int Nobservations_point,
                                           mrcal_problem_selections_t problem_selections,
                                           const mrcal_lensmodel_t* lensmodel);
int mrcal_num_measurements_regularization(int Ncameras_intrinsics, int Ncameras_extrinsics,
                                          int Nframes,
                                          int Npoints, int Npoints_fixed, int Nobservations_board,
                                          mrcal_problem_selections_t problem_selections,
                                          const mrcal_lensmodel_t* lensmodel);
int mrcal_num_measurements(int Nobservations_board, int Nobservations_point,
                           const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                           int Nobservations_point_triangulated,
                           int calibration_object_width_n, int calibration_object_height_n,
                           int Ncameras_intrinsics, int Ncameras_extrinsics,
                           int Nframes,
                           int Npoints, int Npoints_fixed,
                           mrcal_problem_selections_t problem_selections,
                           const mrcal_lensmodel_t* lensmodel);
int mrcal_num_states(int Ncameras_intrinsics, int Ncameras_extrinsics,
                     int Nframes,
                     int Npoints, int Npoints_fixed, int Nobservations_board,
                     mrcal_problem_selections_t problem_selections,
                     const mrcal_lensmodel_t* lensmodel);
int mrcal_state_index_intrinsics(int icam_intrinsics,
                                 int Ncameras_intrinsics, int Ncameras_extrinsics,
                                 int Nframes,
                                 int Npoints, int Npoints_fixed, int Nobservations_board,
                                 mrcal_problem_selections_t problem_selections,
                                 const mrcal_lensmodel_t* lensmodel);
int mrcal_num_states_intrinsics(int Ncameras_intrinsics,
                                mrcal_problem_selections_t problem_selections,
                                const mrcal_lensmodel_t* lensmodel);
int mrcal_state_index_extrinsics(int icam_extrinsics,
                                 int Ncameras_intrinsics, int Ncameras_extrinsics,
                                 int Nframes,
                                 int Npoints, int Npoints_fixed, int Nobservations_board,
                                 mrcal_problem_selections_t problem_selections,
                                 const mrcal_lensmodel_t* lensmodel);
int mrcal_num_states_extrinsics(int Ncameras_extrinsics,
                                mrcal_problem_selections_t problem_selections);
int mrcal_state_index_frames(int iframe,
                             int Ncameras_intrinsics, int Ncameras_extrinsics,
                             int Nframes,
                             int Npoints, int Npoints_fixed, int Nobservations_board,
                             mrcal_problem_selections_t problem_selections,
                             const mrcal_lensmodel_t* lensmodel);
int mrcal_num_states_frames(int Nframes,
                            mrcal_problem_selections_t problem_selections);
int mrcal_state_index_points(int i_point,
                             int Ncameras_intrinsics, int Ncameras_extrinsics,
                             int Nframes,
                             int Npoints, int Npoints_fixed, int Nobservations_board,
                             mrcal_problem_selections_t problem_selections,
                             const mrcal_lensmodel_t* lensmodel);
int mrcal_num_states_points(int Npoints, int Npoints_fixed,
                            mrcal_problem_selections_t problem_selections);
int mrcal_state_index_calobject_warp(int Ncameras_intrinsics, int Ncameras_extrinsics,
                                     int Nframes,
                                     int Npoints, int Npoints_fixed, int Nobservations_board,
                                     mrcal_problem_selections_t problem_selections,
                                     const mrcal_lensmodel_t* lensmodel);
int mrcal_num_states_calobject_warp(mrcal_problem_selections_t problem_selections,
                                    int Nobservations_board);

/* reference: internal.h:99-114, mrcal.c:743-882 */
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
                         const mrcal_lensmodel_t* lensmodel);

/* .cameramodel files at the C level: the on-disk form of one camera (lens
   model, intrinsics, imager size, pose), a python dict literal. Host code.
   reference: types.h:347-376 (the struct: the intrinsics follow the header, as
   many as mrcal_lensmodel_num_params() says), mrcal.h:858-890 (the functions),
   cameramodel-parser.re:356-790, mrcal.c:6626-6680. Only lensmodel, intrinsics,
   imagersize and extrinsics/rt_cam_ref are read; other keys are skipped. The
   writer prints %.17g (the reference prints %f). */
typedef struct
{
    double            rt_cam_ref[6];
    unsigned int      imagersize[2];
    mrcal_lensmodel_t lensmodel;
    double            intrinsics[0];
} mrcal_cameramodel_VOID_t;
#define mrcal_cameramodel_t mrcal_cameramodel_VOID_t
/* these allocate; release with mrcal_free_cameramodel(). NULL on error. len > 0:
   the string need not be 0-terminated; len <= 0: it is */
mrcal_cameramodel_VOID_t* mrcal_read_cameramodel_string(const char* string, const int len);
mrcal_cameramodel_VOID_t* mrcal_read_cameramodel_file  (const char* filename);
void                      mrcal_free_cameramodel(mrcal_cameramodel_VOID_t** cameramodel);
/* these read into a caller's buffer with room for *Nintrinsics_max intrinsics.
   false on failure; if the buffer was too small, *Nintrinsics_max says what is
   needed, otherwise it comes back <= 0 */
bool mrcal_read_cameramodel_string_into(mrcal_cameramodel_VOID_t* model, int* Nintrinsics_max,
                                        const char* string, const int len);
bool mrcal_read_cameramodel_file_into  (mrcal_cameramodel_VOID_t* model, int* Nintrinsics_max,
                                        const char* filename);
bool mrcal_write_cameramodel_file(const char* filename, const mrcal_cameramodel_VOID_t* cameramodel);

/* ------------------------------------------------------------------------ */
/* RESIDENT TIER                                                             */
/* ------------------------------------------------------------------------ */

/* Returns NULL-terminated static string describing the last error on this
   thread ("" if none) */
const char* mrcal_amd_last_error(void);

/* Number of visible HIP devices; <=0 if there is no usable GPU */
int mrcal_amd_device_count(void);

typedef struct mrcal_amd_problem mrcal_amd_problem_t;

/* Uploads a whole optimization problem (same arguments as
   mrcal_optimizer_callback(), host pointers) to the current HIP device and
   builds the iteration-invariant structure (state/measurement layout, CSR
   rowptr/colidx) there. Returns NULL on error.

   shard_begin_frame/shard_end_frame select a contiguous range of FRAMES whose
   board observations this problem instance owns (multi-GPU sharding by
   frame, one process per GPU); pass 0,-1 for "everything". All ranks still
   see the full state vector; only the measurements are partitioned.
   is_shard_leader says whether this instance owns the rows that belong to no
   frame (discrete points, triangulated points, regularization). */
mrcal_amd_problem_t*
mrcal_amd_problem_create(const double*                 intrinsics,
                         const mrcal_pose_t*           rt_cam_ref,
                         const mrcal_pose_t*           rt_ref_frame,
                         const mrcal_point3_t*         points,
                         const mrcal_calobject_warp_t* calobject_warp,
                         int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                         int Npoints, int Npoints_fixed,
                         const mrcal_observation_board_t* observations_board,
                         const mrcal_observation_point_t* observations_point,
                         int Nobservations_board,
                         int Nobservations_point,
                         const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                         int Nobservations_point_triangulated,
                         const mrcal_point3_t* observations_board_pool,
                         const mrcal_point3_t* observations_point_pool,
                         const mrcal_lensmodel_t* lensmodel,
                         const int* imagersizes,
                         mrcal_problem_selections_t problem_selections,
                         double calibration_object_spacing,
                         int calibration_object_width_n,
                         int calibration_object_height_n,
                         int shard_begin_frame, int shard_end_frame,
                         bool is_shard_leader);
/* The same with the discrete points and the triangulated points sharded as well: the shard owns
   the points [shard_begin_point, shard_end_point) - their 3x3 blocks, and their observations wherever those sit
   in the caller's array - and the triangulated point SETS [shard_begin_tripoint, shard_end_tripoint) (a set = the
   consecutive observations of one point up to last_in_set; its pairs are its own rows). An end < 0: all of that
   kind with the leader, as mrcal_amd_problem_create() does. The camera block of the state and the regularization
   rows stay with the leader */
mrcal_amd_problem_t*
mrcal_amd_problem_create_sharded(const double*                 intrinsics,
                         const mrcal_pose_t*           rt_cam_ref,
                         const mrcal_pose_t*           rt_ref_frame,
                         const mrcal_point3_t*         points,
                         const mrcal_calobject_warp_t* calobject_warp,
                         int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                         int Npoints, int Npoints_fixed,
                         const mrcal_observation_board_t* observations_board,
                         const mrcal_observation_point_t* observations_point,
                         int Nobservations_board,
                         int Nobservations_point,
                         const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                         int Nobservations_point_triangulated,
                         const mrcal_point3_t* observations_board_pool,
                         const mrcal_point3_t* observations_point_pool,
                         const mrcal_lensmodel_t* lensmodel,
                         const int* imagersizes,
                         mrcal_problem_selections_t problem_selections,
                         double calibration_object_spacing,
                         int calibration_object_width_n,
                         int calibration_object_height_n,
                         int shard_begin_frame, int shard_end_frame,
                         int shard_begin_point, int shard_end_point,
                         int shard_begin_tripoint, int shard_end_tripoint,
                         bool is_shard_leader);
void mrcal_amd_problem_destroy(mrcal_amd_problem_t* problem);

/* sizes of the LOCAL (this shard's) problem */
int     mrcal_amd_problem_Nstate       (const mrcal_amd_problem_t* problem);
int     mrcal_amd_problem_Nmeasurements(const mrcal_amd_problem_t* problem);
int64_t mrcal_amd_problem_Nnz          (const mrcal_amd_problem_t* problem);
/* Algorithmic HBM bytes of one launch of the board Jacobian kernel on this
   problem: per observation of P corners with k nonzeros per row,
   24 P (read qx,qy,weight) + 16 P (write x) + 16 P k (write J values); where the
   triangulated pairs of a structure-from-motion problem ride in that launch (round 6: a
   problem with boards and pairs whose intrinsics are locked), + per pair 48 (two
   observation vectors) + its record + 8 (x) + 8 per partial written */
int64_t mrcal_amd_problem_jacobian_algorithmic_bytes(const mrcal_amd_problem_t* problem);
/* waits for everything queued on the problem's stream */
bool    mrcal_amd_problem_synchronize  (mrcal_amd_problem_t* problem);

/* Device pointers into the problem's resident buffers (valid until destroy):
   the packed state the next evaluation uses, and the outputs of the last
   evaluation. J is CSR: rowptr int32[Nmeas+1], colidx int32[Nnz], values
   double[Nnz], exactly the arrays mrcal_optimizer_callback() fills */
double*  mrcal_amd_problem_dev_b_packed(mrcal_amd_problem_t* problem);
double*  mrcal_amd_problem_dev_x       (mrcal_amd_problem_t* problem);
int32_t* mrcal_amd_problem_dev_J_rowptr(mrcal_amd_problem_t* problem);
int32_t* mrcal_amd_problem_dev_J_colidx(mrcal_amd_problem_t* problem);
double*  mrcal_amd_problem_dev_J_values(mrcal_amd_problem_t* problem);

/* host <-> device copies of the packed state */
bool mrcal_amd_problem_set_b_packed(mrcal_amd_problem_t* problem, const double* b_packed_host);
bool mrcal_amd_problem_get_b_packed(mrcal_amd_problem_t* problem, double* b_packed_host);
bool mrcal_amd_problem_get_x       (mrcal_amd_problem_t* problem, double* x_host);
/* any of the three may be NULL */
bool mrcal_amd_problem_get_J       (mrcal_amd_problem_t* problem,
                                    int32_t* rowptr_host, int32_t* colidx_host, double* values_host);

/* One evaluation of x (and J if with_jacobian) at the resident packed state.
   Asynchronous on the problem's stream unless sync. This is the Jacobian
   build the roofline is quoted on */
bool mrcal_amd_problem_evaluate(mrcal_amd_problem_t* problem, bool with_jacobian, bool sync);

/* The HIP stream (hipStream_t, as void*) all of this problem's kernels are
   launched on, so that callers can bracket them with their own events */
void* mrcal_amd_problem_stream(mrcal_amd_problem_t* problem);

/* Timing of the most recent evaluation's dominant kernel (the board
   Jacobian build), measured with HIP events on the problem's stream.
   Returns milliseconds, <0 if unavailable */
double mrcal_amd_problem_last_jacobian_kernel_ms(mrcal_amd_problem_t* problem);

/* Records a HIP event pair around every `stride`-th Jacobian-kernel launch from
   now on (up to `capacity` pairs), on the problem's stream; _end() stops, waits
   and reports the number of launches timed and their total/min/max duration
   (ms). This is how the benchmark measures the dominant kernel over its timed
   region. An event pair costs the stream ~11 us (5.6 on each side of the
   kernel), which is why the solver's steps carry none unless asked, and why the
   benchmark asks for a stride. _begin(p, n) == _begin_strided(p, n, 1).
   (mrcal_amd_problem_last_jacobian_kernel_ms() refers to host-driven
   mrcal_amd_problem_evaluate() calls.) */
bool mrcal_amd_problem_jacobian_timing_begin(mrcal_amd_problem_t* problem, int capacity);
bool mrcal_amd_problem_jacobian_timing_begin_strided(mrcal_amd_problem_t* problem, int capacity, int stride);
bool mrcal_amd_problem_jacobian_timing_end(mrcal_amd_problem_t* problem, int* Nlaunches,
                                           double* total_ms, double* min_ms, double* max_ms);
/* Round 5, the splined models with one camera: the camera block's coupled control points in a nested-dissection order
   where the boards leave a strip of the grid worth having - two sides whose panels the big Cholesky
   factors side by side, the strip last. out[9]: the rounds (panels a side) the factorization's launches are provided
   for (0: the dissection is not in use), the largest separator they serve, and the current operating point's plan:
   used or not, the columns of side A, of side B (padded to whole panels of 64), of the separator; what the best strip there would give (side A, side B,
   separator, unpadded). For tests and
   tools; MRCAL_AMD_NO_ND=1 in the environment when the problem is created turns the dissection off */
bool mrcal_amd_problem_dissection(mrcal_amd_problem_t* problem, int* out);

/* ---- multi-GPU: one process per GPU, frames sharded over the ranks ----------
   No counterpart in the reference (it is single-threaded). Every rank creates
   its shard with mrcal_amd_problem_create(shard_begin_frame, shard_end_frame,
   is_shard_leader) from the SAME inputs, attaches a communicator, and calls
   mrcal_amd_problem_solve() / _run_steps() like a single-GPU caller: the
   device-controlled dog-leg step then runs with TWO all-reduces per trial
   step, issued from C++ on the problem's stream (RCCL over xGMI):
       comm 0  [ S (Nc*Nc) | r (Nc) | g_S (Nc) | |x|^2 | status ]   after the local Schur reduction
       comm 1  [ g^T JtJ g | |g_E|^2 | |gn_E|^2 | gn_E . g_E ]       after the local back-substitution
   Rank-local: the frame poses of the shard's frames (state, gradient, steps),
   its rows of x and J. Replicated, and advanced identically by every rank from
   the same sums: the camera block of the state (intrinsics, extrinsics, warp)
   and the trust-region control block. Nothing is read back between trial steps.

   The communicator: rank 0 makes a 128-byte id (mrcal_amd_comm_unique_id),
   any side channel carries it to the other ranks, every rank calls
   mrcal_amd_comm_create(id, rank, world). RCCL is opened at run time */
typedef struct mrcal_amd_comm mrcal_amd_comm_t;
bool              mrcal_amd_comm_unique_id(void* id128);
mrcal_amd_comm_t* mrcal_amd_comm_create(const void* id128, int rank, int world);
/* The same interface over a POSIX shared-memory segment, for the ranks of one
   host: synchronous, staged through host memory, the sum taken in rank order.
   For exercising the sharded solve where RCCL cannot run (two ranks on one
   device); not a transport to measure. name: "/x", unused, the same on every
   rank. A rank that does not arrive within MRCAL_AMD_HOST_COMM_TIMEOUT
   seconds (default 120) makes the collective fail */
mrcal_amd_comm_t* mrcal_amd_comm_create_host(const char* name, int rank, int world);
void              mrcal_amd_comm_destroy(mrcal_amd_comm_t* comm);
int               mrcal_amd_comm_rank (const mrcal_amd_comm_t* comm);
int               mrcal_amd_comm_world(const mrcal_amd_comm_t* comm);
long              mrcal_amd_comm_Ncollectives(const mrcal_amd_comm_t* comm);
long long         mrcal_amd_comm_Ndoubles(const mrcal_amd_comm_t* comm);        /* doubles summed, over all collectives */
int               mrcal_amd_comm_world_observed(const mrcal_amd_comm_t* comm);  /* ncclCommCount(): what the transport itself says */
/* in-place sum over the ranks of n doubles in device memory, queued on the HIP stream */
bool              mrcal_amd_comm_allreduce_sum(mrcal_amd_comm_t* comm, double* buf_dev, int64_t n, void* stream);

/* From now on solve()/run_steps() of this shard run the sharded step over comm
   (not owned; must outlive the problem or be detached with NULL) */
bool  mrcal_amd_problem_attach_comm(mrcal_amd_problem_t* problem, mrcal_amd_comm_t* comm);
/* after a solve: every rank gets the whole state (the frame poses of the
   other ranks' frames) into its resident b_packed. One all-reduce of Nstate doubles */
bool  mrcal_amd_problem_gather_state(mrcal_amd_problem_t* problem);

/* The same step for a driver that brings its OWN collectives (the protocol
   tests: mrcal_amd/parallel.py drives two shards on one device over gloo). Per
   trial step the driver queues, on the problem's stream,
       enqueue(0,0) | all-reduce comm_buffer(0) | enqueue(0,1) | all-reduce comm_buffer(1)
   and never reads anything back in between. initial=1: the evaluation of the
   starting point, after sharded_reset(). snapshot(slot)/wait(slot): a pinned copy
   of the control block, queued after a trial step and waited for a few steps
   later, tells the host when the device has declared the solve finished; wait()
   writes out[4] = { done, error, Nsteps_accepted, Ntrials }. finish() drains the
   stream and makes the final point current; out_i[5] = { Nsteps_accepted,
   Nevaluations, Nfactorizations, Ntrials, error }, out_d[3] = { trust region,
   |x|^2, lambda } */
bool  mrcal_amd_problem_sharded_reset      (mrcal_amd_problem_t* problem, int check_termination, int max_iterations,
                                            double trustregion0);
bool  mrcal_amd_problem_sharded_enqueue    (mrcal_amd_problem_t* problem, int initial, int segment);
void* mrcal_amd_problem_sharded_comm_buffer(mrcal_amd_problem_t* problem, int which, int64_t* Nelements);
bool  mrcal_amd_problem_sharded_snapshot   (mrcal_amd_problem_t* problem, int slot);
bool  mrcal_amd_problem_sharded_wait       (mrcal_amd_problem_t* problem, int slot, int* out);
bool  mrcal_amd_problem_sharded_finish     (mrcal_amd_problem_t* problem, int* out_i, double* out_d);
/* What this shard holds. Writes min(Ninfo, MRCAL_AMD_SHARD_INFO_N) ints and returns how many it knows of (the caller
   says how big its buffer is; the list only ever grows at the end):
     info[0]  Nstate            the full state vector (every shard holds all of it)
     info[1]  S_split           first state index that is NOT one of the leading dense-block variables: in a sharded
                                (frames/points-eliminated) problem the state is [0,S_split) intrinsics + extrinsics,
                                [S_split, S_split+NE) the eliminated variables - frames (6 each), then the variable
                                points (3 each) -, then the board warp (mrcal_amd_problem_partition() has the general form)
     info[2]  NE                number of eliminated variables (all shards' together)
     info[3]  Nc                size of the dense camera block (intrinsics + extrinsics + warp)
     info[4], info[5]  frame_lo, frame_hi     the frames [lo,hi) whose board observations and 6x6 blocks this shard owns
     info[6]  is_leader         this shard owns the camera block of the state, the warp and the regularization rows
     info[7]  Ncorners_local    board corners in this shard's observations
     info[8]  Nframe_blocks     6x6 blocks of the whole problem
     info[9]  Npoint_blocks     3x3 blocks of the whole problem
     info[10], info[11]  point_lo, point_hi   the eliminated BLOCKS [lo,hi), lo >= Nframe_blocks, of the variable points
                                this shard owns (its discrete-point observations and its triangulated pairs) */
#define MRCAL_AMD_SHARD_INFO_N 12
int   mrcal_amd_problem_shard_info(mrcal_amd_problem_t* problem, int* info, int Ninfo);
/* Which pose blocks the problems created from now on eliminate: 0 the library chooses (default: the extrinsics when
   there are at least 4 rt_cam_ref and more extrinsics than frame + point variables, e.g. a moving camera or a big
   stationary rig seen in a handful of frames; the frames and points otherwise), 1 the frames and points, 2 the
   extrinsics where the problem allows it (board rows only, no unity_cam01 row, not splined, not sharded). The results
   of a solve do not depend on it beyond rounding; the block form of mrcal_amd_problem_get_normal_equations() and the
   cost of a step do. Returns the previous setting. (Processes that cannot call it: MRCAL_AMD_ELIMINATE=frames|extrinsics) */
int   mrcal_amd_set_elimination(int policy);
/* Round 6: the solve WITHOUT the Jacobian stream. mrcal_optimize() returns no Jacobian (mrcal.h:453-521) and nothing
   in the device-side dog leg reads the CSR values of J - the per-observation Grams carry the normal equations -, so a
   solve may leave the 16 P k bytes per observation unwritten: the board kernel forms its rows, the
   residuals and the Gram exactly as ever (the same instructions in the same order: x, b_packed, the outlier marks and
   every statistic come out THE SAME BITS) and does not stream the rows to HBM. A product mode; the benchmark's metric -
   a step = one evaluation of x AND J + one solve of the normal equations - is measured with the stream on.
     mrcal_amd_problem_set_jacobian_stream(problem, 0)   _solve() / _run_steps() of this resident problem leave the stream
         out (default 1: on). J is then made on demand: _get_J(), _dev_J_values() and _evaluate(with_jacobian) evaluate
         it at the resident state first. Honoured where the board kernel is the rows' only reader
         (_jacobian_stream_is_optional(): boards under a parametric lens model; the splined models' assembly reads
         the rows back, discrete points and triangulated pairs keep theirs: a few MB)
     mrcal_amd_set_optimize_jacobian_stream(1)   the drop-in mrcal_optimize(), whose problem does not outlive the call,
         streams J all the same (default 0: it does not)
   Both return the previous setting */
/* Hooks for the TESTS (not configuration): force, for the problems prepared after the call, paths a solve takes by
   itself only when a later point outgrows what its first point needed, so that the suite can hold them to the bits of
   the ordinary path. "lchol_likely_panels" = k: k launches of the big camera block's Cholesky are provided one by one,
   the rest goes through lchol_tail_kernel; "nd_rounds" = k: launches for k rounds of the nested dissection whatever
   the plan needs; "lchol_sweep" = 1: the big Cholesky's solve by the backward sweep (the stable fallback the solver
   switches to by itself when a factor's diagonal spans more than 1e10); "lchol_fallback_log10" = k: that threshold as
   10^k. 0: not forced. Returns the previous value,
   -1 for an unknown name. (Round 6: these were environment variables; the library reads six of those now -
   MRCAL_AMD_GRAPH, _ELIMINATE, _RCCL, _LIB, _NO_ND, _NO_SPL_COMPACT - and MRCAL_AMD_DEBUG_SOLVER, which only prints) */
int   mrcal_amd_set_test_hook(const char* name, int value);
/* Round 6: the big camera block's solve ends with d = -Y^T z, Y = L^-1 formed explicitly beside the panels: fast, and
   not backward stable - its error grows like n eps max/min of L's diagonal. The factorizations of a dog-leg pass leave
   that ratio behind (_lchol_diag_ratio(): min / max, 1 where no such factorization ran); below 1e-10 (the test hook
   "lchol_fallback_log10" = k: 10^k) the solver warns and switches THIS problem to the backward sweep in groups of panels
   for good (_uses_sweep(): no explicit inverse, no compaction of the camera block; about 100 us a step slower), from the
   next dog-leg pass on */
double mrcal_amd_problem_lchol_diag_ratio(mrcal_amd_problem_t* problem);
int    mrcal_amd_problem_uses_sweep(mrcal_amd_problem_t* problem);
int   mrcal_amd_problem_set_jacobian_stream(mrcal_amd_problem_t* problem, int stream);
int   mrcal_amd_problem_jacobian_stream_is_optional(mrcal_amd_problem_t* problem);
int   mrcal_amd_set_optimize_jacobian_stream(int stream);
/* Which part of the state the solver keeps as the dense block S and which it
   eliminates block by block (E). The state vector is the reference's either way;
     S index s -> state s (s < info[0]) or s + info[1];   E index e -> state info[2] + e
   info[3]: 0 the frames and points are eliminated (stationary cameras), 1 the
   extrinsics are (a moving camera: many rt_cam_ref, few frames; chosen at
   creation, test_calibration_helpers.py:422-493 builds such problems). The
   blocks of mrcal_amd_problem_get_normal_equations() are in these indices */
void  mrcal_amd_problem_partition(mrcal_amd_problem_t* problem, int info[4]);
/* The outlier bits of this problem's (this shard's) triangulated observations after a solve: flags[i] (0/1, at most
   Nflags written) belongs to observations_point_triangulated[*first + i] of the array the problem was created from.
   Returns the number of observations the problem holds. mrcal_optimize() writes them into the caller's array like
   the reference does (mrcal.c:4225, 4375); a sharded solve leaves the gathering to its driver */
int   mrcal_amd_problem_get_triangulated_outliers(mrcal_amd_problem_t* problem, int* flags, int Nflags, int* first);
/* outlier statistics / marking on the local board observations (mrcal.c:3978-4402);
   counts (device int[4]) and sums (device double[1]) are accumulated into */
bool  mrcal_amd_problem_phase_outlier_stats(mrcal_amd_problem_t* problem, int iop, double thresh_sq,
                                            int* counts_dev, double* sums_dev);
bool  mrcal_amd_problem_phase_mark_outliers(mrcal_amd_problem_t* problem, int iop, double thresh_sq, int* counts_dev);
/* which operating point get_b_packed()/get_x()/get_J() read */
void  mrcal_amd_problem_set_current(mrcal_amd_problem_t* problem, int iop);
int   mrcal_amd_problem_current(mrcal_amd_problem_t* problem);

/* The whole solve on a resident problem: dog-leg iterations + outlier
   rejection (if the problem selections ask for it), exactly what
   mrcal_optimize() does between packing and unpacking the state. Returns the
   rms error sqrt(|x|^2/Nmeasurements), <0 on failure. max_iterations<=0: the
   reference's 300. The solution stays resident: read it with
   mrcal_amd_problem_get_b_packed()/get_x() */
double mrcal_amd_problem_solve(mrcal_amd_problem_t* problem, int max_iterations,
                               int* Noutliers_board);

/* Exactly Nsteps dog-leg steps (accepted or rejected) from the current
   operating point, without termination tests or outlier rejection: the unit
   of work the benchmark times. trustregion_inout (may be NULL) carries the
   trust-region radius across calls; <=0 on input means "the default".
   Returns the number of steps taken, <0 on error */
int mrcal_amd_problem_run_steps(mrcal_amd_problem_t* problem, int Nsteps, double* trustregion_inout);

/* Counters of the most recent solve / run_steps. Any pointer may be NULL */
void mrcal_amd_problem_solver_stats(mrcal_amd_problem_t* problem,
                                    int* Niterations, int* Nevaluations, int* Nfactorizations,
                                    int* Noutlier_passes, double* norm2_x, double* lambda, double* seconds);

/* K (6, Nstate) row-major of _mrcal_drt_cross_reprojection__dbpacked() from the Jacobian the problem holds
   (evaluated here at its current state; nothing crosses PCIe but K and one small record per observation).
   Zero outside the extrinsics / frames / points / calobject_warp blocks, like the reference's wrapper leaves it */
bool mrcal_amd_problem_drt_cross_reprojection(mrcal_amd_problem_t* problem, int icam_intrinsics, double* K);

/* Test/diagnostic access. Evaluates at the resident state and copies out the
   normal equations N = JtJ in the solver's block form (see
   csrc/solver_kernels.hip): A (Nc*Nc), Bt (NE*Nc), D (NEb*6*6), g = Jt x
   (Nstate), |x|^2; dims = {Nc, NE, NEb, Nfb, S_split, Nwarp}: S_split as in
   mrcal_amd_problem_partition() - with the frames eliminated it is the number of
   intrinsics + extrinsics variables ("Nie" before round 3); with the extrinsics
   eliminated (moving cameras) it is the number of intrinsics variables, and the
   frames follow in S behind the shift. Any pointer may be NULL */
bool mrcal_amd_problem_get_normal_equations(mrcal_amd_problem_t* problem,
                                            double* A, double* Bt, double* D, double* g,
                                            double* norm2_x, int* dims);
/* d = -(JtJ + lambda I)^-1 Jt x at the resident state, d (Nstate) to the host */
bool mrcal_amd_problem_gauss_newton_step(mrcal_amd_problem_t* problem, double* step);
/* the board observation pool of this shard, with any newly marked outliers */
bool mrcal_amd_problem_get_board_pool(mrcal_amd_problem_t* problem, mrcal_point3_t* pool_local);

/* ---- factorization of JtJ: the CHOLMOD_factorization equivalent -----------
   Reference: the mrcal.CHOLMOD_factorization Python type, mrcal-pywrap.c:111-214
   (constructor from Jt), :425-569 (solve_xt_JtJ_bt), :580-592 (rcond).
   The Jacobian comes as a host CSR matrix (Nmeas x Nstate, int32 offsets, packed
   units), exactly what optimizer_callback() returns. The state partition tells
   the structured solver which variables form the dense shared block (the first
   Nstate_shared_leading: intrinsics and extrinsics; plus the last Nwarp) and
   which are the mutually independent 6x6 frame / 3x3 point blocks in between.
   Pass (Nstate,0,0,0) for a matrix without that structure (dense; small only).
   Returns NULL if JtJ is not positive definite (the reference returns None) */
typedef struct mrcal_amd_factorization mrcal_amd_factorization_t;
mrcal_amd_factorization_t*
mrcal_amd_factorization_create(int Nmeas, int Nstate,
                               const int32_t* rowptr, const int32_t* colidx, const double* values,
                               int Nstate_shared_leading, int Nframe_blocks, int Npoint_blocks, int Nwarp);
/* The same object from a resident problem, at its current state: x, J and the block normal equations by the problem's own
   kernels (no atomics: the same bits every time), copied device to device. What optimizer_callback() returns */
mrcal_amd_factorization_t* mrcal_amd_factorization_create_from_problem(mrcal_amd_problem_t* problem);
/* Why the last _create() / _create_from_problem() of the calling thread returned what it did: 0 a factorization; 1 NULL
   because JtJ is not positive definite (the reference's None, mrcal-pywrap.c:1981-1988); 2 NULL for any other reason
   (no device memory, a shard, a matrix without the declared structure): mrcal_amd_last_error() says which */
int mrcal_amd_factorization_last_status(void);
int    mrcal_amd_factorization_Nmeasurements(const mrcal_amd_factorization_t* f);
void   mrcal_amd_factorization_destroy(mrcal_amd_factorization_t* f);
int    mrcal_amd_factorization_Nstate (const mrcal_amd_factorization_t* f);
/* xt[i,:] = (JtJ)^-1 bt[i,:]; host, C-contiguous (Nrhs,Nstate) */
bool   mrcal_amd_factorization_solve  (mrcal_amd_factorization_t* f, const double* bt, int Nrhs, double* xt);
/* The other systems of cholmod_solve2() (mrcal-pywrap.c:467-493), sys = CHOLMOD's codes:
   0 A, 1 LDLt, 2 LD, 3 DLt, 4 L, 5 Lt, 6 D, 7 P, 8 Pt. The factorization kept here is
       L L^T = P (JtJ) P^T      (an LL^T factorization: D = I, so LD == L, DLt == Lt)
   with P the ordering [ frame blocks | point blocks | intrinsics, extrinsics | warp ]: the
   eliminated blocks first. As with CHOLMOD, the vectors of the L/D systems are in the
   order of P; "P" takes a vector there (x = P b), "Pt" back. mrcal's projection
   uncertainty does  A1 = solve(b,'P'); A2 = solve(A1,'L'); A3 = solve(A2,'D');
   Var = A2 A3^T  (mrcal/model_analysis.py:837-843): the same three calls work here */
bool   mrcal_amd_factorization_solve_sys(mrcal_amd_factorization_t* f, int sys, const double* bt, int Nrhs, double* xt);
/* The consumers of J that go with it (mrcal-genpywrap.py:477-731), on the J the
   factorization was made from, which is resident on the device:
   y (Nstate) = Jt x (Nmeas) ; out (Nx*Nx) = A Jt J At over the leading rows of J, A (Nx*Nstate), Nx <= 8 */
bool   mrcal_amd_factorization_Jt_x     (mrcal_amd_factorization_t* f, const double* x, double* y);
bool   mrcal_amd_factorization_A_Jt_J_At(mrcal_amd_factorization_t* f, const double* A, int Nx, int Nleading_rows_J, double* out);
/* the same for a CSR matrix in host memory (the reference's signatures: mrcal._mrcal_npsp._Jt_x, _A_Jt_J_At) */
bool   mrcal_amd_csr_Jt_x     (int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji, const double* Jx,
                               const double* x, double* y);
bool   mrcal_amd_csr_A_Jt_J_At(int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji, const double* Jx,
                               const double* A, int Nx, int Nleading_rows_J, double* out);
/* (min diag L / max diag L)^2, like cholmod_rcond() */
double mrcal_amd_factorization_rcond  (mrcal_amd_factorization_t* f);

#ifdef __cplusplus
}
#endif
