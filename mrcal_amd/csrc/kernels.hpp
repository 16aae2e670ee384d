// Host-visible launch interface of kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "problem.hpp"

namespace mrcal_amd {


// Device pointers of one operating point of the solver: a state, the cost
// function there and its normal equations. Two of these live in device memory
// (mrcal_amd_problem::d_ops); kernels reach them through an OpRef
struct OpDev
{
    double* b;            // packed state             [Nstate]
    double* x;            // residuals                [Nmeas]
    double* Jv;           // CSR values               [Nnz]
    // normal equations (solver_kernels.hpp); NULL until the solver is prepared
    double* A;            // [Nc][Nc]
    double* Bt;           // [NE][Nc]
    double* D;            // [NEb][6][6]
    double* g;            // [Nstate]   Jt x
    double* scalars;      // [NSCALARS]
    double* step_cauchy;  // [Nstate]
    double* step_gn;      // [Nstate]
    // splined models: the box of control points under each board observation at this point, [Nobs_board][4] =
    // (min ix, max ix, min iy, max iy). Reset by the prologue launch, filled by board_splined_kernel<true>
    // (integer atomicMin/Max: any order, the same result), read by assemble_splined_kernel. NULL: another model
    int*    spl_box;
    // splined models (round 5): the camera block with the control points no board covers at this point put last -
    // [Nc] position -> camera-block variable | [Nc] variable -> position | [1] how many come first (the coupled ones).
    // Made by spl_compact_kernel from spl_box after an evaluation; read by the reduction of a trial step, which leaves
    // the Cholesky the coupled part alone (cholesky_large.hip, LcholCompact). NULL: not tracked
    int*    cperm;
    // ... and, if not NULL, the nested-dissection plan of the coupled part (solver_kernels.hpp, NDH_*)
    int*    ndp;
};

// Which operating point a kernel works on. The index is either known to the
// host (sel == NULL: ops already points at the right one) or lives in device
// memory (the dog-leg control block decides it on the device, so that a whole
// solver step can be queued without the host knowing which of the two points
// is current). skip, if given, points at a device flag that turns the kernel
// into a no-op (solver finished / step aborted)
struct OpRef
{
    const OpDev* ops;
    const int*   sel;
    const int*   skip;
};
__device__ __forceinline__ bool opref_skip(const OpRef& r) { return r.skip != NULL && *r.skip != 0; }
__device__ __forceinline__ const OpDev& opref_get(const OpRef& r) { return r.ops[r.sel ? *r.sel : 0]; }

struct ChooseArgs;
// threads of a workgroup of the prologue launch and of the stand-alone choice of the trial point: the SAME number,
// since every workgroup of either sums the dog-leg scalars over the state with its threads, in an order that
// their number fixes - and every rank of a sharded solve, with or without boards in its shard, must get the same bits
// (64, one wave. 256 - a quarter of the trips of the scalars' reduction - was measured: the launch 21.0 us instead
//  of 17.5 at the metric's size; the pose records' long dependent chains want their waves spread over all CUs)
#ifndef PRO_T
#define PRO_T 64
#endif      // dogleg_choose.hpp
// (round 6) the pose record of an observation is made by PRO_LPO lanes: lane l < 6 carries the forward-mode tangent
// of variable l of the pose chain (kernels.hip joint_pose_record_lanes). Pose workgroups of the prologue launch
// - where there are few enough observations for that to shorten the launch: with 8 lanes each the metric's 8000
// observations are 1000 pose workgroups instead of 125, every one of which derives the dog-leg step's scalars for itself
// first (dogleg_choose_scalars), and the launch measured 2 us LONGER (20.2 against 18.2 us); at 1600 observations 2 us
// shorter (12.6 against 14.7). Either way the same bits (the lanes run Dual<6>'s instructions component by component)
#define PRO_LPO 8
#define PRO_LPO_MAX_OBS 4096
__host__ __device__ static inline int prologue_lanes(int Nobs_board) { return Nobs_board <= PRO_LPO_MAX_OBS ? PRO_LPO : 1; }
__host__ __device__ static inline int prologue_obs_blocks(int Nobs_board)
{
    const int per_wg = PRO_T/prologue_lanes(Nobs_board);
    return (Nobs_board + per_wg - 1)/per_wg;
}
// what one evaluation reads and writes
struct EvalBuffers
{
    OpRef    R;      // b in; x, Jv out
    double*  joint;  // prologue scratch         [Nobs_board][JOINT_STRIDE] + unpacked intrinsics
    int32_t* Jp;     // CSR rowptr               [Nmeas+1]
    int32_t* Ji;     // CSR colidx               [Nnz]
    double*  gram;   // per-observation Gram     [Nobs_board][gram_stride(Ndist)]; NULL: don't form it
    // if zero_total > 0 the prologue kernel also clears the point's normal
    // equations: A[zero_n[0]], Bt[zero_n[1]], D[zero_n[2]], g[zero_n[3]], scalars[zero_n[4]]
    long long zero_n[5];
    long long zero_total;
    // HOST pointer, read when the prologue is launched (never on the device). Given: the solver's trial step lets
    // the prologue launch choose the trial point it evaluates (board_prologue_kernel<true>); R then only names the
    // operating point for the kernels AFTER the prologue
    const ChooseArgs* choose;
    // (round 6) false: the board kernel forms its rows, the residuals and the Gram but does NOT stream the CSR values
    // of J to HBM - for solves in which nothing reads them (mrcal_optimize() returns no Jacobian, mrcal.h:453-521; the
    // device-side dog leg works from the Grams). Honoured only where gram != NULL and the rows have no other reader
    // (board_kernel; never the splined models, whose assembly reads the rows back, nor points / pairs). The same bits
    // in x and the Gram either way. The metric's step is defined WITH the stream
    bool store_jacobian;
};
// can the evaluation's prologue launch carry the choice of the trial point (EvalBuffers::choose)?
bool prologue_takes_choose(const DeviceProblem& P);
// the triangulated pairs ride in the board kernel's launch (kernels.hip board_tri_kernel) when the Jacobian and the Grams are asked for
bool board_launch_takes_triangulated(const DeviceProblem& P);

bool lens_supported(int lens_type);

// x (and J values if with_jacobian) at B.b. ev_j0/ev_j1, if given, bracket the
// board Jacobian kernel on the stream
// parts: which of the evaluation's kernels to queue (the solver splits an
// evaluation around the board kernel when that kernel is being timed)
// ZERO: clearing the point's normal equations (needed before REST assembles them; independent of the rest)
// ASSEMBLE: the block normal equations from what was evaluated (the fused solver step does that itself)
enum { EVAL_PART_PROLOGUE = 1, EVAL_PART_BOARD = 2, EVAL_PART_REST = 4, EVAL_PART_ZERO = 8, EVAL_PART_ASSEMBLE = 16,
       EVAL_PART_ALL = 31 };
hipError_t launch_evaluate(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                           int lds_bytes, hipStream_t stream,
                           hipEvent_t ev_j0, hipEvent_t ev_j1, int parts = EVAL_PART_ALL);

// rowptr/colidx; iteration-invariant
hipError_t launch_structure(const DeviceProblem& P, const EvalBuffers& B, hipStream_t stream);

// N points through a lens model, device pointers; dq_dp / dq_dintrinsics may be
// NULL. For the splined model dq_dintrinsics must have been zeroed
hipError_t launch_project_points(int lens_type, const LensConfig& cfg, int N, int Nintrinsics,
                                 const double* p, const double* intrinsics,
                                 double* q, double* dq_dp, double* dq_dintrinsics, hipStream_t stream);

// N pixels back to observation vectors, device pointers; dv_dq (N,3,2) / dv_di (N,3,Nintrinsics) may be NULL (dv_di
// needs dv_dq). scratch: q (2N), dq_dv (6N), dq_di (2 N Nintrinsics) for the models inverted iteratively
hipError_t launch_unproject_points(int lens_type, const LensConfig& cfg, int N, int Nintrinsics,
                                   const double* q, const double* intr,
                                   double* v, double* dv_dq, double* dv_di,
                                   double* scratch_q, double* scratch_dq_dv, double* scratch_dq_di,
                                   bool normalize, hipStream_t stream);

} // namespace mrcal_amd
