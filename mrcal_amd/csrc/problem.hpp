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
// v_mfma_f64_4x4x4f64 (4 independent 4x4 products per instruction, one per
// "slot"; k = 4 tile rows per step). Measured operand layout on gfx950
// (tools/mfma_f64_4x4x4_layout_probe.hip):
//   A lane = 16 k + 4 slot + i     B lane = 16 k + 4 slot + j     D lane = 16 i + 4 slot + j
// With lane l reading tile[row 4 step + l/16][16 g + l%16], ONE LDS read is a
// valid operand (A and B alike) holding the 4x4 column blocks 4g..4g+3 of
// "group" g in its four slots. G is cut into 4x4 blocks; the block pairs are
// produced by multiplying a group operand with a slot-ROTATED group operand
// (DPP row_ror within each 16-lane row, no LDS traffic):
//   candidate (gA, gB, r): slot s holds block pair (4 gA + s, 4 gB + (s+r)%4)
//     same group: r = 0 the diagonal blocks, r = 1 (s,s+1) incl. the mirrored
//                 (3,0), r = 2 (0,2),(1,3) [slots 2,3 would repeat them]
//     groups 0x1: r = 0..3, all 16 pairs
// i.e. 2 LDS reads and <= 10 MFMAs per step for up to 32 tile columns, where
// gathering each pair's operands separately costs 2 reads per MFMA. The LDS
// pipe, shared by all the waves of a CU, is what this kernel runs out of first.
// Candidates with no valid slot (a short last group) are dropped; the
// remaining ones are the accumulators m = 0.., stored as gram[iobs][m][lane]:
// lane l of accumulator m holds G[4 bA + l/16][4 bB + l%4], (bA,bB) the pair
// in slot (l%16)/4.
__host__ __device__ constexpr int  gram_ngroups(int nblk) { return (nblk + 3) >> 2; }
__host__ __device__ constexpr void gram_cand_decode(int c, int* gA, int* gB, int* r)
{
    if(c < 3)      { *gA = 0; *gB = 0; *r = c; }
    else if(c < 6) { *gA = 1; *gB = 1; *r = c - 3; }
    else           { *gA = 0; *gB = 1; *r = c - 6; }
}
__host__ __device__ constexpr bool gram_slot_valid(int nblk, int c, int s)
{
    int gA = 0, gB = 0, r = 0;
    gram_cand_decode(c, &gA, &gB, &r);
    const int bA = 4*gA + s, bB = 4*gB + ((s + r) & 3);
    if(bA >= nblk || bB >= nblk)        return false;
    if(gA == gB && r == 2 && s >= 2)    return false;   // (2,0),(3,1): repeats of (0,2),(1,3)
    return true;
}
__host__ __device__ constexpr bool gram_cand_valid(int nblk, int c)
{
    return gram_slot_valid(nblk,c,0) || gram_slot_valid(nblk,c,1) || gram_slot_valid(nblk,c,2) || gram_slot_valid(nblk,c,3);
}
// number of accumulators, and the candidate behind accumulator m
__host__ __device__ constexpr int gram_nmfma_blk(int nblk)
{
    int n = 0;
    for(int c = 0; c < 10; c++) if(gram_cand_valid(nblk, c)) n++;
    return n;
}
__host__ __device__ constexpr int gram_cand(int nblk, int m)
{
    for(int c = 0; c < 10; c++)
        if(gram_cand_valid(nblk, c)) { if(m == 0) return c; m--; }
    return -1;
}
__host__ __device__ inline int gram_stride(int ndist) { return gram_nmfma_blk(tile_nblk(ndist))*64; }
// position pos = 64 m + lane of a stored Gram -> the entry G[i][j] it holds.
// diag: the entry is in a diagonal 4x4 block, where (i,j) and (j,i) are both
// stored; elsewhere only one of them is (i > j can happen: the mirrored (3,0)
// block). Returns false for the unused slots
__host__ __device__ inline bool gram_pos_to_entry(int nblk, int pos, int* i, int* j, bool* diag)
{
    const int lane = pos & 63, s = (lane >> 2) & 3;
    const int c = gram_cand(nblk, pos >> 6);
    if(c < 0 || !gram_slot_valid(nblk, c, s)) return false;
    int gA = 0, gB = 0, r = 0;
    gram_cand_decode(c, &gA, &gB, &r);
    const int bA = 4*gA + s, bB = 4*gB + ((s + r) & 3);
    *i = 4*bA + (lane >> 4);
    *j = 4*bB + (lane & 3);
    *diag = (bA == bB);
    return true;
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
    long long* debug_ts;     // per-observation phase timestamps (profiling builds, -DBOARD_TS)

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
