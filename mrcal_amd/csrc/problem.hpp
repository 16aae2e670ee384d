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
    JOINT_REC     = 84,   // what the board kernels read of a record
    JOINT_STRIDE  = 112   // doubles between records: 896 bytes, 7 cache lines
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
// G is cut into 4x4 blocks (NBLK = ncols/4 of them per side). An operand is ONE
// LDS read per lane, tile[row 4 step + l/16][4 blk(slot) + l%4], which serves
// as A or B alike and holds one column block per slot:
//   X     blocks (0,1,2,3): lane l reads column l%16, 16 contiguous doubles per row
//   Y_q   q = 0..ny-1, ny = NBLK-4 < 4: the remaining blocks in cyclic order,
//         slot s holds block 4 + (s+q) % ny
//   Y     (ny = 4) blocks (4,5,6,7)
// and the block pairs come from
//   X . rot_r(X), r = 0,1,2   (rot_r: slots rotated by DPP row_ror, no LDS traffic)
//                             r = 0 the diagonal blocks, r = 1 (s,s+1) incl. the
//                             mirrored (3,0), r = 2 (0,2),(1,3) [slots 2,3 repeat them]
//   ny < 4:  X . Y_q (all of X x Y in ny instructions), Y_0 . Y_0, Y_0 . Y_1
//   ny = 4:  Y . rot_r(Y) r = 0,1,2 and X . rot_r(Y) r = 0..3
// 8 MFMAs, 4 LDS reads and 2 rotations per step for the 27 columns of OPENCV8,
// where gathering every pair's operands separately costs 2 reads per MFMA and
// makes the LDS pipe (shared by all the waves of a CU) the limit, and forming
// everything from X, Y and their rotations costs 10 MFMAs.
// The accumulators are stored as they are, gram[iobs][m][lane]: lane l of
// accumulator m holds G[4 bA + l/16][4 bB + l%4], (bA,bB) the pair in slot (l%16)/4.
enum { GRAM_XX = 0, GRAM_YY, GRAM_XY };
struct GramDesc { int kind, q; };     // q: the rotation r, or the index of the Y_q operand
__host__ __device__ constexpr int  gram_nx(int nblk) { return nblk < 4 ? nblk : 4; }
__host__ __device__ constexpr int  gram_ny(int nblk) { return nblk - gram_nx(nblk); }
__host__ __device__ constexpr bool gram_rotated_y(int nblk) { return gram_ny(nblk) == 4; }
// LDS reads per step: X, then Y (ny = 4) or Y_0..Y_{ny-1}
__host__ __device__ constexpr int  gram_nreads(int nblk) { return 1 + (gram_rotated_y(nblk) ? 1 : gram_ny(nblk)); }
// the candidate accumulators, in order: XX r=0,1,2 | YY | XY
__host__ __device__ constexpr int  gram_ncand(int nblk)
{
    const int ny = gram_ny(nblk);
    if(ny == 0) return 3;
    if(ny == 4) return 3 + 3 + 4;
    return 3 + (ny >= 2 ? 2 : 1) + ny;
}
__host__ __device__ constexpr GramDesc gram_cand_desc(int nblk, int c)
{
    const int ny  = gram_ny(nblk);
    const int nyy = (ny == 4) ? 3 : (ny >= 2 ? 2 : (ny == 1 ? 1 : 0));
    if(c < 3)       return GramDesc{ GRAM_XX, c };
    if(c < 3 + nyy) return GramDesc{ GRAM_YY, c - 3 };
    return GramDesc{ GRAM_XY, c - 3 - nyy };
}
// slot s of candidate c: its block pair; false if the slot is unused or repeats another
__host__ __device__ constexpr bool gram_slot_pair(int nblk, int c, int s, int* bA, int* bB)
{
    const int nx = gram_nx(nblk), ny = gram_ny(nblk);
    const GramDesc d = gram_cand_desc(nblk, c);
    int a = 0, b = 0;
    bool ok = true;
    if(d.kind == GRAM_XX)
    {
        a = s; b = (s + d.q) & 3;
        ok = a < nx && b < nx && !(d.q == 2 && s >= 2);
    }
    else if(ny == 4)
    {
        b = 4 + ((s + d.q) & 3);
        if(d.kind == GRAM_YY) { a = 4 + s; ok = !(d.q == 2 && s >= 2); }
        else                  { a = s;     ok = a < nx; }
    }
    else if(d.kind == GRAM_XY)
    {
        a = s; b = 4 + (s + d.q) % ny;
        ok = a < nx;
    }
    else
    {
        a = 4 + s % ny; b = 4 + (s + d.q) % ny;
        ok = s < ny && (d.q == 0 || ny == 3 || s == 0);
    }
    *bA = a; *bB = b;
    return ok;
}
__host__ __device__ constexpr bool gram_cand_valid(int nblk, int c)
{
    int a = 0, b = 0;
    return gram_slot_pair(nblk,c,0,&a,&b) || gram_slot_pair(nblk,c,1,&a,&b) ||
           gram_slot_pair(nblk,c,2,&a,&b) || gram_slot_pair(nblk,c,3,&a,&b);
}
// number of accumulators, and the candidate behind accumulator m
__host__ __device__ constexpr int gram_nmfma_blk(int nblk)
{
    int n = 0;
    for(int c = 0; c < gram_ncand(nblk); c++) if(gram_cand_valid(nblk, c)) n++;
    return n;
}
__host__ __device__ constexpr int gram_cand(int nblk, int m)
{
    for(int c = 0; c < gram_ncand(nblk); c++)
        if(gram_cand_valid(nblk, c)) { if(m == 0) return c; m--; }
    return -1;
}
__host__ __device__ inline int gram_stride(int ndist) { return gram_nmfma_blk(tile_nblk(ndist))*64; }
// position pos = 64 m + lane of a stored Gram -> the entry G[i][j] it holds.
// diag: the entry is in a diagonal 4x4 block, where (i,j) and (j,i) are both
// stored; elsewhere only one of them is (i > j can happen: mirrored blocks).
// Returns false for the unused slots
__host__ __device__ inline bool gram_pos_to_entry(int nblk, int pos, int* i, int* j, bool* diag)
{
    const int lane = pos & 63, s = (lane >> 2) & 3;
    const int c = gram_cand(nblk, pos >> 6);
    int bA = 0, bB = 0;
    if(c < 0 || !gram_slot_pair(nblk, c, s, &bA, &bB)) return false;
    *i = 4*bA + (lane >> 4);
    *j = 4*bB + (lane & 3);
    *diag = (bA == bB);
    return true;
}
// tile column this lane reads for operand `iread` (0: X; then Y or Y_q)
__host__ __device__ inline int gram_read_col(int nblk, int iread, int lane)
{
    if(iread == 0) return lane & 15;
    const int ny = gram_ny(nblk), s = (lane >> 2) & 3;
    if(ny == 4) return 16 + (lane & 15);
    return 16 + 4*((s + (iread - 1)) % ny) + (lane & 3);
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
    int elim_extrinsics;       // the solver eliminates the extrinsics blocks, not the frames (NormalDims, solver_kernels.hpp)
    int has_warp_state;  // warp is a state variable
    int has_warp_seed;   // a warp was given (it is applied whether optimized or not)
    int Ncameras_intrinsics, Ncameras_extrinsics, Nframes, Npoints, Npoints_fixed;
    int Nobs_board, Nobs_point;
    int W, H;
    double spacing;
    double inv_Wm1, inv_Hm1;   // 1/(W-1), 1/(H-1): the board warp's normalized coordinates
    double seed_warp[2];

    // model configuration that is not in the intrinsics vector
    LensConfig cfg;

#ifdef MRCAL_AMD_DEV
    // MEASUREMENT BUILDS ONLY (bash csrc/build.sh -DMRCAL_AMD_DEV [-DBOARD_TS ...] -> libmrcal_amd_dev.so; the shipped
    // library has neither field nor any of the code behind them). Ablation knob of tools/probe_board.py: bit0 skip the J
    // copy-out, bit1 skip the MFMAs, bit2 skip the projection arithmetic, bit4 exit after the start-up loads
    int debug_ablate;
    long long* debug_ts;     // per-observation phase timestamps (-DBOARD_TS)
#endif

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
