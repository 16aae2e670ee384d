// The problem as the kernels see it: a POD passed by value at launch, pointing
// at HBM-resident arrays that live for the life of a mrcal_amd_problem_t.
//
// HBM layout (all allocated once in mrcal_amd_problem_create()):
//
//   seeds (unpacked units; used for any block that is locked down)
//     seed_intrinsics   double[Nci][Nintrinsics]
//     seed_rt_cam_ref   double[Nce][6]
//     seed_rt_ref_frame double[Nf][6]
//     seed_points       double[Npoints][3]
//   observations
//     board_meta        BoardObsMeta[Nobs_board]     (indices + CSR offsets)
//     board_pool        double[Nobs_board][H][W][3]  (qx,qy,weight)
//     point_meta        PointObsMeta[Nobs_point]
//     point_pool        double[Nobs_point][3]
//   per-evaluation scratch
//     joint             double[Nobs_board][JOINT_STRIDE]   (see JointPose below)
//   outputs
//     b_packed          double[Nstate]
//     x                 double[Nmeas]
//     J_rowptr          int32[Nmeas+1]    } CSR of J, identical to what the
//     J_colidx          int32[Nnz]        } reference's callback writes into
//     J_values          double[Nnz]       } Jt->p, Jt->i, Jt->x
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "layout.hpp"
#include "lens_models.hpp"

namespace mrcal_amd {

// iteration-invariant description of one board observation
struct BoardObsMeta
{
    int32_t icam_intrinsics;
    int32_t icam_extrinsics;   // <0: camera sits at the reference
    int32_t iframe;
    int32_t nnz_per_row;       // k
    int32_t i_state_intrinsics;// first state index of this camera's intrinsics; <0 if none
    int32_t i_state_extrinsics;// <0 if none in this row
    int32_t i_state_frame;     // <0 if none
    int32_t i_meas0;           // first measurement row of this observation
    int64_t i_nnz0;            // first CSR entry of this observation
    int64_t _pad;
};
static_assert(sizeof(BoardObsMeta) == 48, "BoardObsMeta layout");

struct PointObsMeta
{
    int32_t icam_intrinsics;
    int32_t icam_extrinsics;
    int32_t i_point;
    int32_t nnz_per_row;
    int32_t i_state_intrinsics;
    int32_t i_state_extrinsics;
    int32_t i_state_point;     // <0 if this point is fixed / not optimized
    int32_t i_meas0;
    int64_t i_nnz0;
    int64_t _pad;
};
static_assert(sizeof(PointObsMeta) == 48, "PointObsMeta layout");

// The per-observation geometry the board kernel needs, produced by the
// prologue kernel (one lane per observation) and consumed wave-uniformly.
//
// With the joint transform  p = Rj pt + tj,  pt = (X,Y,Z) a board corner:
//   dp_i/drc_l = X Mc[0][i][l] + Y Mc[1][i][l] + Z Mc[2][i][l] + dtj_drc[i][l]
//   dp_i/drf_l = X Mf[0][i][l] + Y Mf[1][i][l] + Z Mf[2][i][l]
//   dp/dtc     = I,   dp/dtf = dtj_dtf
// where M?[j][i][l] = sum_k dRj[i][j]/drj[k] drj[k]/dr?[l]. Folding the chain
// rule through the joint rotation ONCE per observation turns the reference's
// per-point 3x(3x3)(3x3) products (mrcal.c:2304-2361) into 3 axpys per point
enum
{
    JOINT_R       = 0,    // 9
    JOINT_T       = 9,    // 3
    JOINT_MC      = 12,   // 27
    JOINT_DTJ_DRC = 39,   // 9
    JOINT_MF      = 48,   // 27
    JOINT_DTJ_DTF = 75,   // 9
    JOINT_STRIDE  = 84
};

// Per-board-observation Gram matrix G = Tt T of the observation's Jacobian
// tile T (columns in state order, plus the residual as a last column), as the
// MFMA accumulators leave it: up to 3 column blocks of 16 -> 6 upper-triangular
// 16x16 tiles, tile-major, each tile [v][lane] with G[16bi + lane/16 + 4v][16bj + lane%16]
enum { GRAM_NB_MAX = 3, GRAM_NT_MAX = 6, GRAM_STRIDE = GRAM_NT_MAX*256 };
__host__ __device__ inline int gram_tile_index(int bi, int bj) // bi <= bj
{
    // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    return bi*GRAM_NB_MAX - bi*(bi-1)/2 + (bj - bi);
}
// G[i][j], any i,j
__host__ __device__ inline double gram_get(const double* g, int i, int j)
{
    if(i > j) { int t = i; i = j; j = t; }
    int bi = i >> 4, bj = j >> 4, ii = i & 15, jj = j & 15;
    if(bi == bj && ii > jj) { int t = ii; ii = jj; jj = t; } // diagonal tiles are symmetric
    return g[gram_tile_index(bi,bj)*256 + (ii >> 2)*64 + ((ii & 3) << 4) + jj];
}

struct DeviceProblem
{
    // layout
    int lens_type;
    int Nintrinsics, Ncore, Ncore_state, Ndist, Ndist_state, Nintr_state;
    int i_state_intrinsics, i_state_extrinsics, i_state_frames, i_state_points, i_state_warp;
    int Nstate, Nmeas;
    int do_optimize_extrinsics, do_optimize_frames;
    int has_warp_state;  // warp is a state variable
    int has_warp_seed;   // a warp was given (it is applied whether optimized or not)
    int Ncameras_intrinsics, Ncameras_extrinsics, Nframes, Npoints, Npoints_fixed;
    int Nobs_board, Nobs_point;
    int W, H;
    int board_tile_stride;   // doubles per LDS tile row of the board kernel, sized for the widest row
    double spacing;
    double seed_warp[2];

    // model configuration that is not in the intrinsics vector
    LensConfig cfg;

    // performance-debugging knob (tools/probe_board.py): bit0 skip the J copy-out,
    // bit1 skip the MFMAs, bit2 skip the projection arithmetic. 0 in production
    int debug_ablate;

    // regularization
    int do_apply_regularization;
    int has_unity_cam01;
    int i_meas_regularization;
    int64_t i_nnz_regularization;
    double imager_width_cam0;

    const double* seed_intrinsics;
    const double* seed_rt_cam_ref;
    const double* seed_rt_ref_frame;
    const double* seed_points;

    const BoardObsMeta* board_meta;
    const double*       board_pool;
    const PointObsMeta* point_meta;
    const double*       point_pool;
    const int*          imagersizes;
};

} // namespace mrcal_amd
