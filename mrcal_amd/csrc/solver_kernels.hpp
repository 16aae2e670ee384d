// Host-visible interface of the solver's kernels (assembly.hip, assembly_splined.hip, schur.hip, cholesky_lds.hip,
// cholesky_large.hip, step.hip, factorization_solve.hip: until round 6 one translation unit, solver_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "problem.hpp"
#include "kernels.hpp"

namespace mrcal_amd {

// Partition of the state into the dense shared block S and the block-diagonal
// eliminated set E (see assembly.hip). Which blocks are eliminated is a
// property of the problem: the numerous, mutually independent
// ones. Stationary cameras and a moving board (or points): the frames and
// points are eliminated, S = intrinsics + extrinsics + warp. A moving camera
// and a stationary board (many rt_cam_ref, few frames): the extrinsics are
// eliminated, S = intrinsics + frames + points + warp (elim_extrinsics). The
// state vector keeps the reference's order either way:
//   [intrinsics | extrinsics | frames | points | warp]
// and S, E are index ranges of it:
//   S index s  ->  state s            for s <  S_split
//                  state s + S_shift  for s >= S_split
//   E index e  ->  state E_state0 + e
struct NormalDims
{
    int Nstate;
    int Nwarp;        // 0 or 2: the last S indices
    int i_state_warp;
    int Nc;           // size of S
    int NE;           // size of E
    int Nfb;          // 6x6 blocks of E (frames, or extrinsics), first in E
    int Npb;          // 3x3 blocks of E (points), after them
    int NEb;          // Nfb + Npb
    int S_split, S_shift, E_state0;
    int elim_extrinsics;
};
__host__ __device__ inline int S_to_state(const NormalDims& nd, int s) { return (s < nd.S_split) ? s : s + nd.S_shift; }
__host__ __device__ inline int E_to_state(const NormalDims& nd, int e) { return nd.E_state0 + e; }
// the stationary-camera partition over a state [shared Nshared | E (NE) | warp (Nwarp)]
__host__ __device__ inline void normal_dims_set_partition(NormalDims& nd, int Nshared_leading)
{
    nd.S_split = Nshared_leading; nd.S_shift = nd.i_state_warp - Nshared_leading; nd.E_state0 = Nshared_leading;
    nd.elim_extrinsics = 0;
}

// The E blocks a shard owns: a contiguous range of frame blocks plus (on the
// shard leader) all the point blocks
struct BlockRanges
{
    int frame_lo, frame_hi;   // frame blocks [lo,hi)
    int point_lo, point_hi;   // point blocks, as block indices (>= Nfb)
    __host__ __device__ int count() const { return (frame_hi - frame_lo) + (point_hi - point_lo); }
    __host__ __device__ int block(int i) const
    {
        const int nf = frame_hi - frame_lo;
        return (i < nf) ? frame_lo + i : point_lo + (i - nf);
    }
    // E-row range of part 0 (frames) / 1 (points)
    __host__ __device__ void e_range(const NormalDims& nd, int part, int* lo, int* hi) const
    {
        if(part == 0) { *lo = 6*frame_lo; *hi = 6*frame_hi; }
        else          { *lo = 6*nd.Nfb + 3*(point_lo - nd.Nfb); *hi = 6*nd.Nfb + 3*(point_hi - nd.Nfb); }
    }
};

// per-operating-point scalars (OpDev::scalars). All are zeroed when the
// point's normal equations are assembled; [SC_GN_LENSQ..SC_STEP_SS] again at
// the start of every trial step taken from the point
enum
{
    SC_NORM2_X = 0,        // |x|^2                                   (assembly)
    SC_G_GNG,              // g^T N g  } the quadratic-form kernel     (after the assembly)
    SC_G_GG,               // g . g    } writes (vNv, g.v, v.v)
    SC_G_GG2,              // g . g    } to 3 consecutive slots
    SC_GN_LENSQ,           // |step_gn|^2
    SC_GN_DOT_CAUCHY,      // step_gn . step_cauchy
    SC_STEP_SNS,           // step^T N step
    SC_STEP_GS,            // g . step
    SC_STEP_SS,            // |step|^2
    SC_TMP0, SC_TMP1, SC_TMP2, SC_TMP3,   // host-driven paths
    SC_BAD_STRUCTURE,      // > 0: a row handed to the generic assembly couples two eliminated blocks, or has a column out of range
    NSCALARS = 16
};

// The nested-dissection order of the splined models' camera block (round 5; cholesky_large.hip, lchol_nd_*).
// OpDev::ndp, made by spl_compact_body after every evaluation:  [NDH_WORDS] the plan | [Nc] camera-block variable -> its
// class << 28 | its index within the class (class 0: separator, 1: side A, 2: side B, 3: not coupled) | [Nc + 2 ND_PANEL]
// position -> variable over [A padded | B padded | separator] (-1: a pad)
#define ND_PANEL 64              // = LCH_NB: the sides are padded to whole panels of the factorization
#define LCH_ND_WMAX 1024         // the most columns a side may have
enum { NDH_ACTIVE = 0,           // this point's camera block goes by the dissection
       NDH_NA, NDH_NB,           // columns of the two sides, padded (0 if not active)
       NDH_NS,                   // the separator's (not active: all the coupled variables)
       NDH_IDEAL_A, NDH_IDEAL_B, NDH_IDEAL_NS,   // what the best strip would give (unpadded), whether it fits what the host provided or not: learn_likely_size() sizes that from it
       NDH_NSEFF,                // the size of the matrix the ordinary factorization factors: NDH_NS if active, else the coupled variables
       NDH_ARAW, NDH_BRAW,       // the sides' variables (NDH_NA, NDH_NB less the pads)
       NDH_WORDS };
__host__ __device__ inline size_t nd_plan_ints(int Nc) { return (size_t)NDH_WORDS + (size_t)Nc + (size_t)Nc + 2*ND_PANEL; }
hipError_t launch_nd_plans_off(const struct OpDev* ops, int Nc, hipStream_t stream);
// what the host provided launches for: [0] rounds (panels a side; 0: none - the plans are made and not used) | [1] the largest separator
struct NdLimits { int rounds, ns_max; };

// factorization scratch, one set
struct FactorBuffers
{
    double* Wt;       // [NE][Nc]   L_e^-1 Bt_e
    double* LD;       // [NEb][6][6] Cholesky factors of the D blocks
    double* y;        // [NE]       L_e^-1 g_e
    double* S;        // [Nc][Nc]   Schur complement -> its Cholesky factor (lower)
    double* r;        // [Nc]       reduced rhs -> d_s
    double* Spart;    // [schur_partial_doubles(nd)] per-slice partial products of the SYRK, summed by schur_reduce_kernel
    double* Linv;     // [cholesky_large_workspace_doubles(Nc)] inverses of the 64x64 diagonal blocks (large Nc only)
    int*    status;   // [1] nonzero: not positive definite
    unsigned* occ;    // [NEb][occ_words(nd)] bit per 16-column tile of the camera block: does the block's Wt hold a nonzero there?
                      // Written by eblock_factor_kernel, read by the sparse SYRK (the splined models). NULL: not tracked
    int     lchol_likely_panels; // with cperm_cur: the factorization's launches the host provides one by one (the rest: lchol_tail_kernel); 0: all
    int*    cperm_cur; // [2 Nc + 2] (the last word: lchol_tail_kernel's barrier) the permutation (OpDev::cperm) of the point whose camera block was reduced last: what the
                      // factorization and the solve behind that reduction go by. NULL: no compaction
    double* iso;      // with cperm_cur: [4 (Nc/2 + 1)] the 2 x 2 blocks of the isolated pairs (s00, s10, s11, -) | [Nc] their rhs
    // the dissection (NULL / 0: none): the two sides' matrices and workspaces, the plan of the point reduced last (a copy
    // of its OpDev::ndp), the limits as the host set them (and their device copy, which the plans are made against)
    double* ndMA; double* ndMB; double* ndLinvA; double* ndLinvB;
    double* ndPart;   // [ceil(Nc/16)][2 LCH_ND_WMAX] L_SX^T d_S in shares of 16 entries of d_S (LcholCompact::ndpart)
    int*    ndp_cur;
    int*    nd_lim_dev;
    NdLimits nd_lim;
    int     nd_likely_panels;   // the separator's panels at the solve's first point (as lchol_likely_panels)
    int     use_sweep; // the large Cholesky's solve by the backward sweep in groups of panels (backward stable; slower: no explicit
                      // L^-1, no compaction, the end-of-trial logic in launches of its own) instead of d = -Y^T z. Set by the
                      // automatic fallback (solver.cpp: a factor whose diagonal spans more than 1e10) or by a test hook
    unsigned long long* diag_minmax; // [2] the smallest / largest diagonal entry of the big camera block's Cholesky factor since the solve's
                      // start (bit patterns of positive doubles; set to +inf, 0 by ctl_reset()): what the fallback to the sweep goes by. NULL: no large Cholesky
    double* Wtile;    // with occ: a second copy of the tiles of Wt that hold something, tile column by tile column -
                      // [ceil(Nc/16)][NE][16] - for the sparse SYRK: a block's rows of a tile are 768 contiguous bytes
                      // (in Wt itself they are six pieces 9.6 KB apart, and a workgroup's few blocks that count are all over 46 MB)
};
inline int occ_words(const NormalDims& nd) { return (((nd.Nc + 15) >> 4) + 31) >> 5; }

// Hooks for the tests (mrcal_amd_set_test_hook(), include/mrcal_amd.h): force paths a solve takes by itself only when a
// later point of it outgrows what its first point needed. 0: not forced. Read where a solve starts
struct TestHooks
{
    int lchol_likely_panels;   // launches of the large Cholesky provided one by one; the rest go through lchol_tail_kernel
    int nd_rounds;             // rounds of the dissection provided for, whatever the plan needs
    int lchol_sweep;           // the large Cholesky's solve by the backward sweep (FactorBuffers::use_sweep) from the start
    int lchol_fallback_log10;  // the automatic fallback's threshold on min / max of the factor's diagonal as a power of ten (default -10)
};
TestHooks& test_hooks();

// size of FactorBuffers::Linv: the multi-launch Cholesky of camera blocks that do not fit the LDS
size_t cholesky_large_workspace_doubles(int n);
// size of FactorBuffers::Spart (schur.hip: SYRK slicing)
size_t schur_partial_doubles(const NormalDims& nd);

// iteration-invariant work lists for the assembly
// what one position of one observation's Gram contributes to
enum { PAIROP_NONE = 0,
       PAIROP_A,      // A[a][b] (+ A[b][a])          aux = a | b << 16 (S indices)
       PAIROP_G,      // g[aux]                       aux = state index
       PAIROP_NORM,   // |x|^2
       PAIROP_D,      // D_f[a][b] (+ D_f[b][a])      aux = a | b << 16 (0..5)
       PAIROP_BT,     // Bt_f[a][b]                   aux = frame variable a | S index b << 16
       PAIROP_GF,     // g_f[a]
       PAIROP_MIRROR = 0x100 };
struct PairOp { int op, aux; };
enum { FRAMEPOS_NONE = 0, FRAMEPOS_D, FRAMEPOS_D_MIRROR, FRAMEPOS_GF, FRAMEPOS_BT_INTRINSICS, FRAMEPOS_BT_EXTRINSICS, FRAMEPOS_BT_WARP };

#ifndef REDUCE_CHUNK
#define REDUCE_CHUNK 64
#endif
// (REDUCE_CHUNK)    // observations of one pair summed by one workgroup (reduce_pair_chunk), in batches of 16 loads
// splined models: the packed lower triangle of a pass's local Gram (128 local columns), and the knots it spans
#define SPL_TRI (128*129/2)
#define SPLG_E  8         // workgroups sharing a row of the camera block that every pass holds (assemble_splined_gather_kernel)
struct SplHdr { int ix0, iy0, wx, wy; };      // wx < 0: the observation went row by row, nothing is staged
// An observation whose box of control points does not fit the local tile (more than SPL_SUB_MAX^2 of them) is cut into
// up to SPL_MAXSUB SUB-BOXES of at most SPL_SUB_MAX x SPL_SUB_MAX control points that overlap by `order` - every corner's
// (order+1)^2 patch lies whole in the sub-box its first control point belongs to - and each sub-box is a pass of its
// own over the corners it owns: its own header, its own staged triangle. (Close-ups: a board over a third of a 30 x 20
// grid is 17 x 17 control points, 2 x 2 sub-boxes; one that fills the imager is the whole grid, 4 x 3 of them.
// They used to go row by row, with atomics)
#define SPL_MAXSUB  12
#define SPL_SUB_MAX 10
#ifndef QF_ROWS_PER_WAVE
#define QF_ROWS_PER_WAVE 2    // rows of [A ; Bt] a wave of the quadratic-form workgroups takes
#endif
// Rows that do not come from board Grams and add to destinations they SHARE with other rows: discrete points
// (the intrinsics and extrinsics of their camera, their point's block) and triangulated pairs (the two cameras'
// extrinsics). Summed in a fixed order - no floating-point atomics - like the board path:
//   camera-block part   rows with the same list of camera-block columns form a GROUP (a point row: one per
//                       (camera, x/y); a pair: one per ordered camera pair). The rows of a group, in row order,
//                       are cut into chunks of GEN_CHUNK; a workgroup per chunk forms the chunk's Gram
//                       sum_rows s s^T (upper triangle), sum s x and sum x^2 - every output summed over the
//                       rows in order by ONE thread - into part[chunk][]. assemble_finalize() then adds, per
//                       destination, its (group, position) sources chunk by chunk: the same machinery as for
//                       the pairs' chunk_part
//   eliminated blocks   a wave per point block walks the block's rows in order and owns its rows of Bt, its
//                       D block and its part of g outright
// Built once (problem_prepare_solver) from the CSR structure itself; not built (Nrows = 0: the rows go one lane
// each, with atomics) where that structure changes between evaluations (the splined models' patch columns) or a
// row holds more than GEN_KMAX camera-block columns
#define GEN_CHUNK 128
#define GEN_KMAX  40
struct GenPlan
{
    int     Nrows, Nchunks, Ngroups, stride;   // stride: doubles of a chunk's partial sums (<= 1023)
    int     row_first, row_end;                // the measurement rows it covers
    int     kmax;                              // most camera-block columns in a row
    int*    rows;          // [Nrows] grouped, in row order within a group
    int*    chunk_begin;   // [Nchunks+1] into rows
    int*    chunk_group;   // [Nchunks]
    int*    group_k;       // [Ngroups] camera-block columns per row
    int*    group_off;     // [Ngroups] into spos / scol
    int*    spos;          // position within the row of each camera-block column
    int*    scol;          // its S index
    double* part;          // [Nchunks][stride]: pairs (p <= q) row-major, then k sums s x, then sum x^2
    int     Ndest;         // the finalize lists, as AssemblyPlan::dest_*
    int*    dest_id; int* dest_begin; int* dest_src; int* group_chunk_begin;
    int     Neblocks;      // eliminated blocks that have such rows
    int*    eb_block;      // [Neblocks] block index
    int*    eb_begin;      // [Neblocks+1] into eb_rows
    int*    eb_rows;       // row ids, in row order
    int*    eb_group;      // the group of each of those rows
    int*    eb_epos;       // position within the row of the block's first column
};
// Rows of a splined problem that no fixed-order plan covers (round 5): discrete points under a splined model whose
// distortions are optimized (their patch columns move with every evaluation), and a board observation whose box of
// control points is past SPL_MAXSUB sub-boxes. They used to add to A, Bt, D, g with floating-point atomics, in whatever
// order they landed. Now: sums in which no addition rounds (the pre-rounded sums of launch_assemble_rows(): three levels
// of [A | Bt | D | g | |x|^2], cleared and combined only when such rows exist), added to the blocks at a fixed place in
// the launch order. lvl[0] == NULL: this problem cannot have such rows
struct ReproStep
{
    double*             lvl[3];      // each `one` doubles: [A: Nc^2 | Bt: NE Nc | D: NEb 36 | g: Nstate | |x|^2: 1]; zero at rest
    unsigned long long* cmax;        // [Nstate + 1] the bits of each column's largest |value| over those rows (the last: x); zero at rest
    int*                any;         // [1] some row of this evaluation went this way
    size_t              one;
};
bool splined_needs_repro_rows(const DeviceProblem& P);
struct AssemblyPlan
{
    GenPlan gen;
    ReproStep repro;
    int  spl_compact;      // splined models: the evaluation's assembly also makes OpDev::cperm (spl_compact_kernel)
    const int* nd_lim;     // ... and OpDev::ndp, against these limits (FactorBuffers::nd_lim_dev); NULL: no dissection
    int* frame_obs_begin;  // [blocks+1] the board observations of each 6x6 eliminated block (a frame: contiguous) ...
    int* frame_obs;        // ... or, if not NULL, entries [begin, end) of this list (a camera's, with elim_extrinsics)
    int* chunk_begin;      // [Nchunks+1]
    int* pair_obs;         // [Nobs_board] observation indices grouped by (intrinsics, extrinsics) pair
    int* chunk_pair;       // [Nchunks] the pair of each chunk
    int* obs_pair;         // [Nobs_board] the pair of each observation
    int  Nchunks, Npairs;
    // what each position of a stored Gram holds (gram_pos_to_entry(), evaluated once):
    // bit 31 valid, bit 17 "involves a frame column", bit 16 "in a diagonal block", bits 8..15 i, bits 0..7 j
    int*    pos_table;     // [gram_stride]
    PairOp* pair_table;    // [Npairs][gram_stride]
    // The frame part of the same, without the trip through pair_table: what a position adds to depends on the
    // pair only through WHERE the camera's intrinsics and extrinsics sit in the camera block.
    //   frame_pos[pos]   FRAMEPOS_* kind | a << 3 | k << 6  (a: frame variable; k: second frame variable, or the
    //                    offset within the camera's intrinsics / extrinsics, or the S index of a warp term)
    //   obs_cols[o][2]   S index of the first intrinsic of observation o's camera / of the first variable of its
    //                    pose in the camera block (the camera's extrinsics; the frame with elim_extrinsics); -1: none
    int*    frame_pos;     // [gram_stride]
    int*    obs_cols;      // [Nobs_board][2]
    // Fixed-order reduction of the camera-block part (no atomics: the sums do
    // not depend on scheduling). reduce_pair_chunk() leaves one partial sum per
    // (chunk, Gram position); assemble_finalize() then adds, for every
    // destination (an entry of A, of g, or |x|^2), its sources in a fixed order:
    //   dest_id[k]    destination: [0,Nc^2) entry of A, then Nc entries of g (S index), then |x|^2
    //   dest_begin[k] .. dest_begin[k+1]: its sources in dest_src, each pair << 10 | pos
    //   pair_chunk_begin[pair] .. [pair+1]: the chunks of a pair (contiguous)
    double* chunk_part;       // [Nchunks][gram_stride]
    int*    dest_id;          // [Ndest]
    int*    dest_begin;       // [Ndest+1]
    int*    dest_src;         // [dest_begin[Ndest]]
    int*    pair_chunk_begin; // [Npairs+1]
    int     Ndest;
    double* row_part;         // [row_part_n] per-workgroup partial |x|^2 of the rows that do not come from board Grams
    int     row_part_n;
    double* qf_part;          // [qf_part_n][4] per-workgroup partials of the quadratic form g^T N g (and of |g_E|^2)
    int     qf_part_n;        // = quadform workgroups: (Nc + NE) rows, 4 waves x QF_ROWS_PER_WAVE rows each
    double* dots_part;        // [NEb][2] per-block (|d_e|^2, d_e . g_e) of the back-substitution
    // splined models (assemble_splined_kernel): chunk_part holds the staged Grams, [2 Nobs_board][SPL_TRI]
    SplHdr* spl_hdr;          // [Nobs_board] the knot box of each observation's first (mostly: only) pass, with the number of
                              // its sub-boxes in wy's upper half
    SplHdr* spl_hdr_extra;    // [Nobs_board][SPL_MAXSUB - 1] the boxes of the other sub-boxes
    double* chunk_extra;      // [2 Nobs_board (SPL_MAXSUB - 1)][SPL_TRI] ... and their staged triangles (spl_slot())
    double* spl_part;         // [rows every pass holds][SPLG_E][Nc+1]
};

// state index -> S index (>=0) or -(1 + E index)
__host__ __device__ inline int state_to_SE(const NormalDims& nd, int col)
{
    if(col >= nd.E_state0 && col < nd.E_state0 + nd.NE) return -(1 + (col - nd.E_state0));
    return (col < nd.S_split) ? col : col - nd.S_shift;
}
// What a column of the board kernel's tile (problem.hpp) is, for one observation
// (COL_FRAME: a variable of the observation's ELIMINATED pose - its frame, or with elim_extrinsics its camera)
enum { COL_ABSENT = 0, COL_S, COL_FRAME, COL_X };
struct TileColInfo { int kind; int idx; };   // COL_S: state index; COL_FRAME: 0..5
__host__ __device__ inline
TileColInfo board_tile_col_info(const DeviceProblem& P, const BoardObsMeta& m, int col)
{
    TileColInfo r = { COL_ABSENT, 0 };
    const int nd = P.Ndist;
    if(col < 4)
    {
        if(P.Ncore_state) { r.kind = COL_S; r.idx = m.i_state_intrinsics + col; }
    }
    else if(col < 4 + nd)
    {
        if(P.Ndist_state) { r.kind = COL_S; r.idx = m.i_state_intrinsics + P.Ncore_state + (col - 4); }
    }
    else if(col < tile_frame0(nd))
    {
        if(P.do_optimize_extrinsics && m.icam_extrinsics >= 0)
        {
            if(P.elim_extrinsics) { r.kind = COL_FRAME; r.idx = col - tile_ext0(nd); }
            else                  { r.kind = COL_S;     r.idx = m.i_state_extrinsics + (col - tile_ext0(nd)); }
        }
    }
    else if(col < tile_warp0(nd))
    {
        if(P.do_optimize_frames)
        {
            if(P.elim_extrinsics) { r.kind = COL_S;     r.idx = m.i_state_frame + (col - tile_frame0(nd)); }
            else                  { r.kind = COL_FRAME; r.idx = col - tile_frame0(nd); }
        }
    }
    else if(col < tile_xcol(nd))
    {
        if(P.has_warp_state) { r.kind = COL_S; r.idx = P.i_state_warp + (col - tile_warp0(nd)); }
    }
    else if(col == tile_xcol(nd))
        r.kind = COL_X;
    return r;
}

// The dog-leg control block: everything the trust-region logic needs, in
// device memory. libdogleg keeps this on the host between callbacks; here the
// decisions are taken by one-thread kernels between the vector kernels, so a
// whole step is queued without a host round trip.
struct SolverCtl
{
    // configuration (written by the host before a run)
    double trustregion_decrease_factor, trustregion_decrease_threshold;
    double trustregion_increase_factor, trustregion_increase_threshold;
    double update_threshold, trustregion_threshold;
    int    max_iterations;      // accepted steps
    int    check_termination;   // 0: run exactly the queued steps (benchmark)

    // state
    double trustregion;
    double lambda;              // diagonal regularization, raised when JtJ is not positive definite
    double norm2_x[2], cauchy_lensq[2], gn_lensq[2];
    double gn_dot_g[2];         // g . step_gn, and the lambda the step was computed with
    double gn_lambda[2];
    int    gn_valid[2], did_step_to_edge[2];
    int    ib, ia;              // operating point before / after the step being tried
    int    done;                // the solve has terminated: every later kernel is a no-op
    int    abort_step;          // this trial step is void (factorization failed, lambda was raised)
    int    need_gn;             // this step needs the Gauss-Newton direction, and it is not computed yet
    int    error;               // 1: lambda ran away; 3, 4: internal (solver.cpp solver_error_text())

    // the step being tried
    double step_len_sq, expected_improvement;
    double k_cauchy, k_gn;      // step = k_cauchy step_cauchy + k_gn step_gn

    // counters
    int    Nsteps_accepted, Ntrials, Nfactorizations, Nevaluations;

    // (the fused step, launch_step2_*)
    int    refactor;            // the current point must be (re-)eliminated before a step can be chosen:
                                // lambda was just raised, or its Gauss-Newton step was never computed
    int    gn_fresh;            // step_gn of the current point was just computed: its dot products are not known yet
    int    derive;              // the current point is new: g^T N g, |g|^2 and its Cauchy step are still to be derived
    int    _pad;
};

hipError_t launch_zero_normal(const NormalDims& nd, const OpRef& R, hipStream_t stream);
hipError_t launch_factor_local(const NormalDims& nd, const BlockRanges& br,
                               const OpRef& R, const FactorBuffers& F,
                               double lambda, const SolverCtl* ctl, bool is_leader, hipStream_t stream);
hipError_t launch_solve_backsub(const NormalDims& nd, const BlockRanges& br,
                                const OpRef& R, const FactorBuffers& F, const int* skip_also, bool keep_factor,
                                hipStream_t stream);
hipError_t launch_quadform(const NormalDims& nd, const OpRef& R, const double* v, double* out,
                           hipStream_t stream);
hipError_t launch_dot(int n, const double* a, const double* b, double* out, hipStream_t stream);
hipError_t launch_axpby(int n, double alpha, const double* a, double beta, const double* b, double* y,
                        hipStream_t stream);
// part: scratch of outlier_partial_doubles() doubles (per-workgroup partial sums, added up in a fixed order)
size_t     outlier_partial_doubles();
hipError_t launch_outlier_stats(int Npoints_board, double thresh_sq, const double* x, const double* pool,
                                int* counts, double* sums, double* part, hipStream_t stream);
hipError_t launch_mark_outliers(int Npoints_board, double thresh_sq, const double* x, double* pool,
                                int* counts, hipStream_t stream);

// solves against a kept factorization (F as left by launch_factor_local() +
// launch_solve_backsub(keep_factor)): (JtJ) x = b, device vectors in state order
hipError_t launch_fsolve(const NormalDims& nd, const FactorBuffers& F,
                         const double* b, double* x, hipStream_t stream);
// the systems of cholmod_solve2(), same codes (factorization_solve.hip explains the factor and its order)
enum { FSOLVE_A = 0, FSOLVE_LDLt, FSOLVE_LD, FSOLVE_DLt, FSOLVE_L, FSOLVE_Lt, FSOLVE_D, FSOLVE_P, FSOLVE_Pt };
hipError_t launch_fsolve_sys(const NormalDims& nd, const FactorBuffers& F, int sys,
                             const double* b, double* x, hipStream_t stream);
// nrhs of them side by side: b, x [nrhs][Nstate]; scratch y [nrhs][NE], r [nrhs][Nc], part [nrhs][part_per_rhs]
// (fsolve_batch_scratch_doubles(nd, nrhs) = NE + Nc + part_per_rhs for a batch of that size)
size_t     fsolve_batch_scratch_doubles(const NormalDims& nd, int nrhs);
hipError_t launch_fsolve_sys_batch(const NormalDims& nd, const FactorBuffers& F, int sys,
                                   const double* b, double* x, int nrhs,
                                   double* y, double* r, double* part, size_t part_per_rhs, hipStream_t stream);
// y += Jt x ; out (NX x NX) += A Jt J At over the leading rows (mrcal-genpywrap.py:477-731), CSR J on the device
// y = Jt x and A Jt J At of a device-resident CSR matrix, in a fixed summation order (no atomics). scratch:
// csr_Jt_x_scratch_doubles(Nrows, Ncols) doubles for the first, 64*((Nrows + 255)/256) for the second
size_t csr_Jt_x_scratch_doubles(int Nrows, int Ncols);
hipError_t launch_csr_Jt_x(int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji, const double* Jx, const double* x, double* y,
                           double* scratch, hipStream_t stream);
hipError_t launch_csr_A_Jt_J_At(int NX, int Nrows, int Nstate, const int32_t* Jp, const int32_t* Ji, const double* Jx,
                                const double* A, double* out, double* scratch, hipStream_t stream);
// out2[0] = min, out2[1] = max of the factor's diagonal; preset to (+big, 0)
hipError_t launch_fsolve_diag_minmax(const NormalDims& nd, const FactorBuffers& F, double* out2, hipStream_t stream);
// the normal equations of a bare CSR matrix into the blocks of R's operating point
// (scratch, assemble_rows_scratch_doubles(nd) doubles: the sums made so that no addition rounds - the same bits whatever the
//  order of the atomics; NULL: the plain sums)
size_t assemble_rows_scratch_doubles(const NormalDims& nd);
hipError_t launch_assemble_rows(const NormalDims& nd, const OpRef& R, int Nmeas,
                                const int32_t* Jp, const int32_t* Ji, hipStream_t stream, double* scratch, long long Nnz);

// ---- The device-controlled dog-leg trial step (solver.cpp enqueue_trial_step()). Per trial, in this order:
//   choose            the dog-leg step from the current point, b_trial; the first trial from a new point also
//                     derives its Cauchy step                                          (launch_step2_choose)
//   [prologue, board] x, J, Grams at the trial point                                   (launch_evaluate)
//   assemble+factor   block normal equations of the trial point from the Grams, in a fixed order; the
//                     frame blocks are eliminated on the spot (L, Wt, y): a Gauss-Newton solve of the
//                     point, should it be accepted, is already under way                (launch_step2_assemble)
//   syrk + finalize   Wt^T Wt partial tiles ; A, g_S, |x|^2 from the chunk partials     (launch_step2_reduce)
//   reduce            S, r and the tail of comm1
//                     -- sharded: all-reduce of comm1 --
//   finish + Cholesky accept/reject, trust region; Cholesky of S if the point was accepted  (launch_step2_factor)
//   backsub+quadform  the frame/point part of the Gauss-Newton step ; g^T N g of the new point
//                     -- sharded: pack + all-reduce of comm2 (4 doubles) --
// initial: the evaluation of the starting point (no choose, no accept)
struct Step2Args
{
    const DeviceProblem* P; const NormalDims* nd; const BlockRanges* br; const AssemblyPlan* plan;
    const OpDev* ops; SolverCtl* ctl; const FactorBuffers* F; const double* gram;
    const int32_t* Jp; const int32_t* Ji; double* step; bool is_leader;
    const double* comm2;     // sharded: [4] g^T N g, |g_E|^2, |gn_E|^2, gn_E . g_E summed over the ranks; NULL: single GPU
    SolverCtl* snap;         // host-visible (pinned) copy of the control block to leave behind at the end of the step; NULL: none
    // a side stream with its fork and join events (NULL: none): _assemble may leave work there that _reduce waits for
    hipStream_t side; hipEvent_t ev_fork, ev_join;
};
hipError_t launch_step2_choose(const Step2Args& a, hipStream_t stream);
// the same as arguments, for an evaluation whose prologue launch carries the choice (EvalBuffers::choose)
struct ChooseArgs;
ChooseArgs step2_choose_args(const Step2Args& a);
hipError_t launch_step2_assemble(const Step2Args& a, bool initial, hipStream_t stream);
// ... comm1 = [S | r | g_S | |x|^2 | status] (F.S, step2_comm1_doubles()) is this rank's summand after _reduce;
// _factor expects it summed over the ranks, and leaves this rank's summand of comm2 (if a.comm2 is given)
// initial (0 / 1; -1: not said): the reduction of a trial step / of the starting point - where the dissection's launches follow
// (FactorBuffers::nd_lim), the end-of-trial logic rides in this launch and launch_step2_factor(a, initial) leaves it out
hipError_t launch_step2_reduce(const Step2Args& a, hipStream_t stream, int initial = -1);
hipError_t launch_step2_factor(const Step2Args& a, bool initial, hipStream_t stream);
int64_t    step2_comm1_doubles(const NormalDims& nd);
hipError_t launch_mask_state(const NormalDims& nd, const BlockRanges& br, bool is_leader, double* b, hipStream_t stream);
const int* solver_ctl_skip_eval2(const SolverCtl* ctl);
// host-driven evaluation: deterministic block normal equations of the point R from the Grams (no elimination)
// (with side and the two events: what only A, g of the camera block and |x|^2 wait for goes to the side stream,
//  forked after the kernels Bt, D come from; *forked tells the caller to wait for ev_join before it reads those)
hipError_t launch_assemble(const DeviceProblem& P, const NormalDims& nd, const BlockRanges& br, const AssemblyPlan& plan,
                           const EvalBuffers& B, hipStream_t stream,
                           hipStream_t side = NULL, hipEvent_t ev_fork = NULL, hipEvent_t ev_join = NULL, bool* forked = NULL);

// the control block is followed in memory by its derived flags
size_t     solver_ctl_bytes();
void       solver_ctl_init_flags(void* ctl_image, int icur);
const int* solver_ctl_skip_factor(const SolverCtl* ctl);   // device pointers, given the device pointer of ctl
const int* solver_ctl_skip_eval  (const SolverCtl* ctl);

} // namespace mrcal_amd
