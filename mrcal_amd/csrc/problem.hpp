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

// One row of the triangulated-point residuals: a pair (i0 < i1) of observations
// of one point (mrcal.c:5180-5653)
struct TriPairMeta
{
    int32_t i0, i1;                 // indices into the triangulated observations
    int32_t icam_extrinsics0;       // <0: that camera sits at the reference
    int32_t icam_extrinsics1;
    int32_t i_state_extrinsics0;    // first state index of camera 0's rt; <0 if none in this row
    int32_t i_state_extrinsics1;
    int32_t i_meas;
    int32_t _pad;
    int64_t i_nnz0;
};
static_assert(sizeof(TriPairMeta) == 40, "TriPairMeta layout");

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

// The Jacobian tile of a board observation, as the board kernel lays it out in
// LDS: one row per measurement, columns at FIXED positions that depend on the
// lens model only (NDIST = its number of distortion parameters), whether or
// not the corresponding block is being optimized:
//
//   [0..3]   fx fy cx cy      an x row holds (dq/dfx, 0, w, 0), a y row (0, dq/dfy, 0, w)
//   [4..)    distortions      NDIST
//   [EXT0..) r_cam t_cam      6
//   [FRAME0..) r_frame t_frame 6
//   [WARP0..) warp            2
//   [XCOL]   the residual x
//   padding to a multiple of 4, zero
//
// Blocks that are not in the state (or a camera that sits at the reference)
// hold zeros. The fixed layout costs a few zero columns in the Gram; it buys
// compile-time register indexing in the kernel.
__host__ __device__ inline int tile_ext0  (int ndist) { return 4 + ndist; }
__host__ __device__ inline int tile_frame0(int ndist) { return 4 + ndist + 6; }
__host__ __device__ inline int tile_warp0 (int ndist) { return 4 + ndist + 12; }
__host__ __device__ inline int tile_xcol  (int ndist) { return 4 + ndist + 14; }
__host__ __device__ inline int tile_ncols (int ndist) { return 4 + ndist + 15; }
__host__ __device__ inline int tile_nblk  (int ndist) { return (tile_ncols(ndist) + 3) >> 2; }
// LDS row stride in doubles: odd (conflict-free per-lane column writes), >= 4 nblk
__host__ __device__ inline int tile_stride(int ndist) { return (4*tile_nblk(ndist)) | 1; }

// Per-board-observation Gram matrix G = Tt T of the tile, formed with
// v_mfma_f64_4x4x4f64 (4 independent 4x4 blocks per instruction, k = 4 tile
// rows per step). G is cut into 4x4 blocks; the NBLK(NBLK+1)/2 blocks (bi<=bj)
// are numbered row-major over the upper triangle, block pair p goes to MFMA
// p/4, block slot p%4. Measured operand layout on gfx950
// (tools/mfma_f64_4x4x4_layout_probe.hip):
//   A lane = 16 k + 4 slot + i     B lane = 16 k + 4 slot + j     D lane = 16 i + 4 slot + j
// so accumulator m, lane l holds G[4 bi + l/16][4 bj + l%4] of the pair in slot (l%16)/4.
// Storage: gram[iobs][m][lane], gram_nmfma(ndist)*64 doubles per observation
__host__ __device__ inline int gram_npairs(int ndist) { const int n = tile_nblk(ndist); return n*(n+1)/2; }
__host__ __device__ inline int gram_nmfma (int ndist) { return (gram_npairs(ndist) + 3) >> 2; }
__host__ __device__ inline int gram_stride(int ndist) { return gram_nmfma(ndist)*64; }
__host__ __device__ inline int gram_pair_index(int nblk, int bi, int bj) // bi <= bj
{
    return bi*nblk - bi*(bi-1)/2 + (bj - bi);
}
__host__ __device__ inline void gram_pair_unrank(int nblk, int p, int* bi, int* bj)
{
    int i = 0;
    while(i < nblk-1 && p >= nblk - i) { p -= nblk - i; i++; }
    *bi = i; *bj = i + p;
}
// G[i][j], any i,j
__host__ __device__ inline double gram_get(const double* g, int nblk, int i, int j)
{
    if((i >> 2) > (j >> 2)) { int t = i; i = j; j = t; }
    const int p = gram_pair_index(nblk, i >> 2, j >> 2);
    return g[(p >> 2)*64 + 16*(i & 3) + 4*(p & 3) + (j & 3)];
}

struct DeviceProblem
{
    // layout
    int lens_type;
    int Nintrinsics, Ncore, Ncore_state, Ndist, Ndist_state, Nintr_state;
    int Ndist_row;       // distortion columns in one board/point row: Ndist_state, or (order+1)^2 for splined models
    int i_state_intrinsics, i_state_extrinsics, i_state_frames, i_state_points, i_state_warp;
    int Nstate, Nmeas;
    int do_optimize_extrinsics, do_optimize_frames;
    int has_warp_state;  // warp is a state variable
    int has_warp_seed;   // a warp was given (it is applied whether optimized or not)
    int Ncameras_intrinsics, Ncameras_extrinsics, Nframes, Npoints, Npoints_fixed;
    int Nobs_board, Nobs_point;
    int W, H;
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
    // triangulated points: pairs, the observation vectors (3 doubles each, in
    // their camera's coordinates) and the outlier marks (one int per observation)
    int                 Npairs_tri;
    const TriPairMeta*  tri_meta;
    const double*       tri_px;
    const int*          tri_outlier;
    const int*          imagersizes;

    // state unpacked by the prologue kernel, every evaluation:
    //   [Ncameras_intrinsics][Nintrinsics] intrinsics, then the 2 warp values
    const double*       unpacked;
};

} // namespace mrcal_amd
