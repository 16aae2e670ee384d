// HIP kernels of the dog-leg step: assembly of the normal equations in
// arrowhead block form from the per-observation Gram matrices, the Schur
// complement onto the camera block, dense Cholesky, back-substitution, and the
// small vector kernels of the trust-region logic.
//
// This replaces what the reference delegates to libdogleg + CHOLMOD
// (mrcal.c:6435 dogleg_optimize2(); per step: Jt x, cholmod_factorize(Jt) =
// Cholesky of JtJ, cholmod_solve). Nothing here is a port: CHOLMOD is a
// general sparse direct solver; this is a structured solver for the one
// sparsity pattern calibration problems have.
//
// Structure. Split the state into
//   S ("shared"):     all intrinsics, all camera extrinsics, the board warp.
//                     Nc variables, dense coupling.
//   E ("eliminated"): frame poses (6 each) and discrete points (3 each).
//                     No measurement row touches two E blocks, so JtJ
//                     restricted to E is block diagonal.
//   N = JtJ = [ A  B ]     A: Nc x Nc dense          (stored full, row-major)
//             [ Bt D ]     Bt: NE x Nc dense         (row e = column e of B)
//                          D: block diagonal, 6x6 / 3x3 blocks
// and solve N d = -g by  S = A - B D^-1 Bt,  S d_s = -(g_s - B D^-1 g_e),
// d_e = -D^-1 (g_e + Bt d_s).
//
// Dense Bt costs Nc*NE*8 bytes (6.7 MB at 8 cameras x 1000 frames; 46 MB for a
// 1200-parameter splined camera x 800 frames): trivial against 288 GB of HBM,
// and it turns the Schur complement into one SYRK.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "problem.hpp"
#include "solver_kernels.hpp"
#include "dogleg_choose.hpp"
#include <type_traits>

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// index helpers
////////////////////////////////////////////////////////////////////////////////
// E index -> (block, offset in block, block size, E index of block start)
__device__ __forceinline__ void E_to_block(const NormalDims& nd, int e, int* blk, int* a, int* de, int* e0)
{
    if(e < 6*nd.Nfb) { *blk = e/6; *a = e - 6*(*blk); *de = 6; *e0 = 6*(*blk); }
    else
    {
        const int ee = e - 6*nd.Nfb;
        const int ib = ee/3;
        *blk = nd.Nfb + ib; *a = ee - 3*ib; *de = 3; *e0 = 6*nd.Nfb + 3*ib;
    }
}

////////////////////////////////////////////////////////////////////////////////
// assembly from the per-observation Grams
////////////////////////////////////////////////////////////////////////////////

// What to do with each position of an observation's Gram is known in advance:
// it depends on the position and on which (intrinsics, extrinsics) pair the
// observation belongs to, nothing else. problem_prepare_solver() evaluates it
// once into plan.pair_table[pair][pos] (PairOp, solver_kernels.hpp); the kernels
// below only read Grams and add.

// One workgroup per frame. Its observations are contiguous (the API requires
// frame-sorted observations, mrcal-pywrap.c:1063-1138). The Grams are read
// coalesced, position by position; the frame's rows of Bt, its D block and its
// part of g are accumulated in LDS and written out whole:
//   D_f  = sum G[frame,frame]      g_f = sum G[frame,x]     Bt[frame rows][S cols] = sum G[S,frame]
// NO ATOMICS: within one observation every Gram position adds to a different
// entry (gram_pos_to_entry: each unordered block pair is stored once), so the
// observations of the frame are applied one after the other, a barrier in
// between, each thread adding its positions with plain LDS read-modify-writes.
// The sums therefore do not depend on scheduling: the solve is bit-reproducible.
//
// mode 1: from the Grams (the point was just evaluated). mode 2: the blocks are
// read back from the point (re-elimination with a new lambda).
// do_factor: the frame is eliminated on the spot, while its blocks are in LDS:
//   L L^T = D_f + lambda I;  Wt_f = L^-1 Bt_f;  y_f = L^-1 g_f        (what eblock_factor_kernel does)
// lds_f: Btf[6][Nc] | Df[36] | gf[6] | L[36] | rinv[6]
__device__ __forceinline__
void assemble_frame_block(const DeviceProblem& P, const NormalDims& nd, const OpDev& O, const AssemblyPlan& plan,
                          const double* __restrict__ gram, int f, int o0, int o1 /* the frame's observations (mode 1) */,
                          int mode, bool do_factor, double lambda,
                          const FactorBuffers& F, double* __restrict__ lds_f)
{
    double* __restrict__ Btf  = lds_f;
    double* __restrict__ Df   = lds_f + 6*nd.Nc;
    double* __restrict__ gf   = Df + 36;
    double* __restrict__ Ls   = gf + 6;
    double* __restrict__ rinv = Ls + 36;
    const int t = threadIdx.x;
#ifdef ASM_TS
    long long ats[16]; int nats = 0;
#define ATS() do { if(t == 0 && nats < 16) ats[nats++] = clock64(); } while(0)
#else
#define ATS()
#endif
    ATS();
    const int e0 = 6*f;   // frame blocks come first in E
    double* __restrict__ Bt = O.Bt;
    double* __restrict__ D  = O.D;
    double* __restrict__ g  = O.g;

    if(mode == 1)
    {
        for(int i = t; i < 6*nd.Nc + 42; i += blockDim.x) lds_f[i] = 0.0;
        __syncthreads();
        const int npos = gram_stride(P.Ndist);
        for(int ob = o0; ob < o1; ob += 8)
        {
            // where the cameras of the 8 observations in flight sit in the camera block (uniform loads, in flight
            // together with the Gram loads below: nothing here waits for anything but the frame's range)
            int col_i[8], col_e[8];
            int obs[8];
#pragma unroll
            for(int u = 0; u < 8; u++)
            {
                const int i = (ob + u < o1) ? ob + u : o0;
                obs[u] = plan.frame_obs ? plan.frame_obs[i] : i;      // (a frame's observations are contiguous, a camera's a list)
            }
#pragma unroll
            for(int u = 0; u < 8; u++) { col_i[u] = plan.obs_cols[2*obs[u]]; col_e[u] = plan.obs_cols[2*obs[u]+1]; }
            // 8 Gram loads per position in flight
            for(int base = 0; base < npos; base += 2*blockDim.x)
            {
                const int pos0 = base + t;
                double vv[2][8];
                int    rec[2];
#pragma unroll
                for(int w = 0; w < 2; w++)
                {
                    const int pos = pos0 + w*blockDim.x;
                    const int pc  = (pos < npos) ? pos : t;
                    rec[w] = (pos < npos) ? plan.frame_pos[pc] : FRAMEPOS_NONE;
                    // (Masking the loads of the positions without a frame column - whole 64-byte
                    //  lines of a Gram are camera-block only - was measured: 28 us against 26.
                    //  The kernel is not bound by its traffic)
#pragma unroll
                    for(int u = 0; u < 8; u++) vv[w][u] = 0.0;
                    if((rec[w] & 7) != FRAMEPOS_NONE)
                    {
#pragma unroll
                        for(int u = 0; u < 8; u++) vv[w][u] = gram[(size_t)obs[u]*npos + pc];
                    }
                }
                // The observations in order, NO barrier between them: a destination belongs to one position -
                // position = (a camera-block tile column, a frame column), and whatever the camera the tile
                // column lands in a state of its own kind (an intrinsic of some camera, an extrinsic of some
                // camera, a warp term) - hence to one thread, which adds its observations one after the other
#pragma unroll
                for(int w = 0; w < 2; w++)
                {
                    const int kind = rec[w] & 7, a = (rec[w] >> 3) & 7, k = rec[w] >> 6;
                    if(kind == FRAMEPOS_NONE) continue;
                    if(kind == FRAMEPOS_BT_INTRINSICS || kind == FRAMEPOS_BT_EXTRINSICS)
                    {
                        double* __restrict__ row = Btf + a*nd.Nc + k;
#pragma unroll
                        for(int u = 0; u < 8; u++)
                        {
                            const int c = (kind == FRAMEPOS_BT_INTRINSICS) ? col_i[u] : col_e[u];
                            if(ob + u < o1 && c >= 0) row[c] += vv[w][u];
                        }
                    }
                    else
                    {
                        // the same entry for every observation: summed on top of what is there, in order
                        double* __restrict__ dst = (kind == FRAMEPOS_GF)      ? gf + a :
                                                   (kind == FRAMEPOS_BT_WARP) ? Btf + a*nd.Nc + k : Df + a*6 + k;
                        double acc = *dst;
#pragma unroll
                        for(int u = 0; u < 8; u++) if(ob + u < o1) acc += vv[w][u];
                        *dst = acc;
                        if(kind == FRAMEPOS_D_MIRROR) Df[k*6 + a] = acc;
                    }
                }
                ATS();
            }
        }
        __syncthreads();
        ATS();
        for(int i = t; i < 6*nd.Nc; i += blockDim.x) Bt[(size_t)e0*nd.Nc + i] = Btf[i];
        if(t < 36)      D[(size_t)f*36 + t]     = Df[t];
        else if(t < 42) g[nd.E_state0 + e0 + (t-36)] = gf[t-36];
    }
    else
    {
        for(int i = t; i < 6*nd.Nc; i += blockDim.x) Btf[i] = Bt[(size_t)e0*nd.Nc + i];
        if(t < 36)      Df[t]    = D[(size_t)f*36 + t];
        else if(t < 42) gf[t-36] = g[nd.E_state0 + e0 + (t-36)];
        __syncthreads();
    }
    if(!do_factor) return;

    // the 6x6 factorization in registers, one thread (eblock_factor_kernel explains why)
    if(t == 0)
    {
        double M[6][6];
#pragma unroll
        for(int i=0;i<6;i++)
#pragma unroll
            for(int j=0;j<6;j++) M[i][j] = Df[i*6+j] + ((i == j) ? lambda : 0.0);
        bool ok = true;
#pragma unroll
        for(int j=0;j<6;j++)
        {
            double d = M[j][j];
#pragma unroll
            for(int k=0;k<j;k++) d -= M[j][k]*M[j][k];
            if(!(d > 0.0)) { ok = false; d = 1.0; }
            d = sqrt(d);
            const double rd = 1.0/d;
            M[j][j] = d;
            rinv[j] = rd;
#pragma unroll
            for(int i=j+1;i<6;i++)
            {
                double v = M[i][j];
#pragma unroll
                for(int k=0;k<j;k++) v -= M[i][k]*M[j][k];
                M[i][j] = v*rd;
            }
        }
#pragma unroll
        for(int i=0;i<6;i++)
#pragma unroll
            for(int j=0;j<6;j++) Ls[i*6+j] = (j <= i) ? M[i][j] : 0.0;
        if(!ok) atomicExch(F.status, 1);
    }
    ATS();
    __syncthreads();
    ATS();
    if(t < 36) F.LD[(size_t)f*36 + t] = Ls[t];
    double Lr[6][6], ri[6];             // (the strict lower triangle is all the substitution reads)
#pragma unroll
    for(int i=0;i<6;i++)
    {
        ri[i] = rinv[i];
#pragma unroll
        for(int k=0;k<i;k++) Lr[i][k] = Ls[i*6+k];
    }
    // forward substitution, one column of [Bt_f | g_f] per thread and pass
    for(int c = t; c <= nd.Nc; c += blockDim.x)
    {
        double w[6];
#pragma unroll
        for(int i=0;i<6;i++)
        {
            double v = (c < nd.Nc) ? Btf[i*nd.Nc + c] : gf[i];
#pragma unroll
            for(int k=0;k<i;k++) v -= Lr[i][k]*w[k];
            w[i] = v*ri[i];
        }
        if(c < nd.Nc) { for(int i=0;i<6;i++) F.Wt[(size_t)(e0+i)*nd.Nc + c] = w[i]; }
        else          { for(int i=0;i<6;i++) F.y[e0+i] = w[i]; }
    }
    ATS();
#ifdef ASM_TS
    if(t == 0 && (f == 0 || f == 300 || f == 999))
        printf("asm ts f=%d n=%d (zero | loads + adds | sync | bt-write+factor | sync | fsub): %lld %lld %lld %lld %lld %lld\n", f, nats,
               ats[1]-ats[0], ats[2]-ats[1], ats[3]-ats[2], ats[4]-ats[3], ats[5]-ats[4], ats[6]-ats[5]);
#endif
}

// S-S part: observations that see the same (intrinsics, extrinsics) pair add to
// the same entries of A. One workgroup per chunk of one pair's observation list:
// each thread sums its Gram positions over the chunk, in order (coalesced reads,
// 16 in flight), and leaves the sum in chunk_part[chunk][pos]. assemble_finalize()
// adds the chunks up, again in a fixed order. No atomics
// a pair chunk is reduced by one workgroup per 256 Gram positions
__host__ __device__ __forceinline__ int assemble_chunk_slices(const DeviceProblem& P) { return (gram_stride(P.Ndist) + 255) >> 8; }
__device__ __forceinline__
void reduce_pair_chunk(const DeviceProblem& P, const AssemblyPlan& plan,
                       const double* __restrict__ gram, int ichunk, int islice /* which 256 positions */)
{
    const int c0 = plan.chunk_begin[ichunk], c1 = plan.chunk_begin[ichunk+1];
    const int npos = gram_stride(P.Ndist);
    const PairOp* __restrict__ ops = plan.pair_table + (size_t)plan.chunk_pair[ichunk]*npos;
    double* __restrict__ out = plan.chunk_part + (size_t)ichunk*npos;
    const int nobs = c1 - c0;
#ifdef ASM_TS
    const long long cts0 = clock64();
#endif
    // one position per thread (a workgroup per 256 positions of the chunk: two position per thread and half
    // as many workgroups needed 112 registers, and at four waves per SIMD the launch no longer fit the chip
    // at once), 16 observations of it in flight together
    const int  pos  = islice*blockDim.x + threadIdx.x;
    const int  kind = (pos < npos) ? (ops[pos].op & 0xff) : PAIROP_NONE;
    const bool live = (kind == PAIROP_A || kind == PAIROP_G || kind == PAIROP_NORM);
    double acc = 0.0;
    for(int u0 = 0; u0 < nobs; u0 += 16)
    {
        double vv[16];
        unsigned ob[16];                // (problem_prepare_solver() refuses Grams past 2^32 doubles)
#pragma unroll
        for(int u = 0; u < 16; u++)
        {
            const int k = (u0 + u < nobs) ? c0 + u0 + u : c0;
            ob[u] = (unsigned)plan.pair_obs[k]*(unsigned)npos;
        }
#pragma unroll
        for(int u = 0; u < 16; u++) vv[u] = 0.0;
        if(live)
        {
#pragma unroll
            for(int u = 0; u < 16; u++) vv[u] = gram[(size_t)ob[u] + pos];
        }
#pragma unroll
        for(int u = 0; u < 16; u++) acc += (u0 + u < nobs) ? vv[u] : 0.0;
    }
    if(pos < npos) out[pos] = acc;
#ifdef ASM_TS
    if(threadIdx.x == 0 && islice == 0 && (ichunk == 0 || ichunk == plan.Nchunks-1)) printf("asm ts chunk %d of %d (nobs %d) dur=%lld\n", ichunk, plan.Nchunks, nobs, clock64()-cts0);
#endif
}

// Every destination of the camera-block part - an entry of A, of g (S part) or
// |x|^2 - adds its sources (chunk partials of the pairs that touch it) in the
// order of the plan. One thread per destination that has sources. What the rows
// that do not come from Grams added earlier (atomically, into the zeroed
// buffers: regularization rows, which have disjoint destinations; discrete
// points) stays. |x|^2 also takes those rows' per-workgroup partials, in order
// 16 lanes (one DPP row) per destination: the lanes split the chunks of each source,
// then add up in a fixed order
#define FIN_LANES 16
__device__ __forceinline__
void assemble_finalize(int npos, const NormalDims& nd, const OpDev& O, const AssemblyPlan& plan,
                       int gid /* global thread index: destination gid/16, lane gid%16 */)
{
    const int k = gid / FIN_LANES, j = gid % FIN_LANES;
    const bool live = k < plan.Ndest;
    const int kc = live ? k : 0;
    const int d  = plan.dest_id[kc];
    const int s0 = plan.dest_begin[kc], s1 = live ? plan.dest_begin[kc+1] : s0;
    double acc = 0.0;
    for(int s = s0; s < s1; s++)
    {
        const int src = plan.dest_src[s];
        const int pair = src >> 10, pos = src & 1023;
        const int c0 = plan.pair_chunk_begin[pair], c1 = plan.pair_chunk_begin[pair+1];
        const double* __restrict__ cp = plan.chunk_part + pos;
        double a0 = 0.0, a1 = 0.0;
        int c = c0 + j;
        for(; c + FIN_LANES < c1; c += 2*FIN_LANES)
        {
            const double v0 = cp[(size_t)c*npos], v1 = cp[(size_t)(c + FIN_LANES)*npos];
            a0 += v0; a1 += v1;
        }
        if(c < c1) a0 += cp[(size_t)c*npos];
        acc += a0 + a1;
    }
    // (all 16 lanes of the row take part, whether the destination is live or not)
    for(int off = FIN_LANES/2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if(!live || j != 0) return;
    const int nA = nd.Nc*nd.Nc;
    if(d < nA)              O.A[d] += acc;
    else if(d < nA + nd.Nc) { const int sc = d - nA; O.g[S_to_state(nd, sc)] += acc; }
    else
    {
        for(int b = 0; b < plan.row_part_n; b++) acc += plan.row_part[b];
        O.scalars[SC_NORM2_X] += acc;
    }
}

// Rows that do not come from board observations (discrete points,
// regularization): one lane per CSR row, scattered with atomics. These are few
__device__ __forceinline__
void rows_generic_row(const NormalDims& nd, const OpDev& O, int r, int row1,
                      const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                      double* n2_local = NULL /* if given: x^2 goes there instead of into |x|^2 (atomically) */)
{
    const double* __restrict__ Jv = O.Jv;
    const double* __restrict__ x  = O.x;
    double* __restrict__ A  = O.A;
    double* __restrict__ Bt = O.Bt;
    double* __restrict__ D  = O.D;
    double* __restrict__ g  = O.g;
    double* __restrict__ norm2_x = &O.scalars[SC_NORM2_X];
    if(r >= row1) return;
    const int p0 = Jp[r], p1 = Jp[r+1];
    const double xr = x[r];
    if(n2_local) *n2_local = xr*xr; else atomicAdd(norm2_x, xr*xr);
    for(int p = p0; p < p1; p++)
    {
        const int    ci = Ji[p];
        const double vi = Jv[p];
        if((unsigned)ci >= (unsigned)nd.Nstate) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }
        atomicAdd(&g[ci], vi*xr);
        const int si = state_to_SE(nd, ci);
        for(int q = p0; q < p1; q++)
        {
            const int    cj = Ji[q];
            if((unsigned)cj >= (unsigned)nd.Nstate) continue;
            const double v  = vi*Jv[q];
            const int    sj = state_to_SE(nd, cj);
            if(si >= 0 && sj >= 0)
                atomicAdd(&A[(size_t)si*nd.Nc + sj], v);
            else if(si < 0 && sj >= 0)
                atomicAdd(&Bt[(size_t)(-si-1)*nd.Nc + sj], v);
            else if(si < 0 && sj < 0)
            {
                int bi, ai, di, e0i, bj, aj, dj, e0j;
                E_to_block(nd, -si-1, &bi, &ai, &di, &e0i);
                E_to_block(nd, -sj-1, &bj, &aj, &dj, &e0j);
                if(bi == bj) atomicAdd(&D[(size_t)bi*36 + ai*6 + aj], v);
                else         O.scalars[SC_BAD_STRUCTURE] = 1.0;      // no row may touch two E blocks
            }
        }
    }
}

// Board rows of the SPLINED models. There is no per-observation Gram for them
// from the Jacobian kernel (the columns of a row depend on where the corner
// lands in the knot grid), and one lane per row with global atomics for every
// pair of its ~26 entries is 108 M atomics at 160k rows: 13 ms. But:
//   - the rows of ONE observation only touch a small set of camera-block
//     variables: the core, the extrinsics, the warp and the knots under the
//     board, K = (order+1 + span)^2 of them per surface;
//   - an x row touches the x surface only, a y row the y surface only
//     (board_splined_kernel: column col0 + .. + xy), and all rows are equally long
// Three kernels, no atomics whose order matters, the same bits every time:
//   assemble_splined_kernel: one workgroup per frame and surface (x rows, y rows: a pass each). A pass writes its
//     rows DENSELY over the local columns
//         [ K knots | 4 core | 6 extrinsics | 2 warp | 6 frame | x ]
//     into an LDS tile and forms the lower triangle of that matrix's Gram on the FP64 matrix cores: one product
//     yields the A, Bt, D_f, g and |x|^2 contributions at once. What belongs to the frame (Bt, D_f, g_f) leaves
//     through LDS sums - one addition per entry and workgroup -; the camera-block rows and the x row are
//     STAGED, a packed lower triangle per pass, with the knot box in a header. An observation whose box does not fit
//     the tile (more than SPL_TW-19 = 109 knots: a close-up) is cut into overlapping SUB-BOXES, each a pass of its own
//     over the corners it owns (solver_kernels.hpp SPL_MAXSUB)
//   assemble_splined_gather_knots_kernel: a workgroup per control point's row of the camera block, a LANE per place
//     of the row that can hold anything (25 + the core); a pass that holds the row is one load for half a wave
//   assemble_splined_gather_kernel: one workgroup per row that every pass holds (core, extrinsics, warp; the x row:
//     g and |x|^2), split over SPLG_E workgroups each and summed in order by assemble_splined_combine_kernel. It walks
//     the passes in order and adds the row's staged entries into an LDS copy of the row. (Until the end of round 4
//     the control points' rows came this way too: MRCAL_AMD_SPL_ROW_GATHER)
// (The one-kernel version flushed each pass's tile sums with global atomics, 8000 of them
//  per observation; two thirds of a workgroup's time was that flush: 340 us at 30 x 20 knots,
//  800 frames.) An observation of more than SPL_MAXSUB sub-boxes goes the generic way,
// row by row, with atomics; its header says so. Only the lower triangle of A is written
#define SPL_TW      128         // local columns
#define SPL_NDENSE  12
#define SPL_NEXTRA  (SPL_NDENSE + 6 + 1)
// The Gram of a pass on the matrix cores (round 4; it was 36 multiply-adds per row and thread fed by 16 LDS reads:
// 57k of a workgroup's 142k cycles). NS = ceil(NC/16) tile columns in use; the lower triangle of the NS x NS grid
// of 16 x 16 tiles is dealt to the four waves BY TILE ROW, so that a wave's tiles share their operands: wave w has
// row Ia = NS-1-w with its Ia+1 tiles and, if it exists, row Ib = w-(8-NS) (NS = 8: 9 tiles each; 7: 7 each;
// 6: 6,5,5,5; 5: 5,4,3,3). v_mfma_f64_16x16x4: lane (r16 = lane % 16, kq = lane / 16) feeds Jd[4 s + kq][16 I + r16]
// and Jd[4 s + kq][16 J + r16], register v of the result is G[16 I + kq + 4 v][16 J + r16]. A step s of a wave is
// Ia+1 (+2) LDS reads for up to 9 matrix instructions; the reads of step s+1 are issued before the instructions of
// step s. Row stride LD = 16 NS, + 16 if that is a multiple of 32 doubles: the four kq groups of a read then start 32
// banks apart. The rows of a pass that fit the tile (64 KB: two workgroups a CU with room to spare) are taken in one go (100 corners x 80 columns do).
#define SPL_LDS_DOUBLES 8192
typedef double spl_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int spl_tri(int r) { return (r*(r + 1)) >> 1; }
// where pass (observation o, surface xy, sub-box isub) stages its triangle, and its header: the first sub-box of every
// observation where the only one always was (the usual case reads what it always read), the others behind all of those
// (in allocations of their own: with one allocation four times the size, the launches that run side by side - the
//  gather, the SYRK - were 40% slower at BASELINE configuration 2, which has no second sub-box anywhere)
__device__ __forceinline__ double* spl_slot(const AssemblyPlan& plan, int o, int xy, int isub)
{
    return (isub == 0) ? plan.chunk_part  + ((size_t)2*o + xy)*SPL_TRI
                       : plan.chunk_extra + (((size_t)2*o + xy)*(SPL_MAXSUB - 1) + (isub - 1))*SPL_TRI;
}
__device__ __forceinline__ SplHdr* spl_hdr_at(const AssemblyPlan& plan, int o, int isub)
{
    return (isub == 0) ? plan.spl_hdr + o : plan.spl_hdr_extra + (size_t)o*(SPL_MAXSUB - 1) + (isub - 1);
}
// a workgroup barrier that orders the LDS traffic only: the global stores in flight (the staged triangle) are not waited for
__device__ __forceinline__ void spl_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// (+)= the Gram of nsteps x 4 rows (rows past the data are zero): the NA = Ia + 1 tiles of row Ia in acc[0 .. NA-1], the
// nb <= min(4, 9 - NA) tiles of row Ib behind them. FIRST: the accumulators start here (the first rows of a pass)
template<int NA, bool FIRST> __device__ __forceinline__
void spl_gram_mfma(const double* __restrict__ Jd, int LD, int nsteps, int r16, int kq, int offIa, int offIb, int nb, spl_d4 (&acc)[9])
{
    if(FIRST)
    {
#pragma unroll
        for(int u = 0; u < 9; u++) acc[u] = spl_d4{0.0, 0.0, 0.0, 0.0};
    }
    const double* __restrict__ rowp = Jd + kq*LD + r16;
    const int step = 4*LD;
    double b0[NA], b1[NA], aA0, aA1, aB0 = 0.0, aB1 = 0.0;
    auto load = [&](double (&bb)[NA], double& aA, double& aB, const double* __restrict__ rp)
    {
#pragma unroll
        for(int J = 0; J < NA; J++) bb[J] = rp[16*J];
        aA = rp[offIa];
        if(nb > 0) aB = rp[offIb];
    };
    auto mma = [&](const double (&bb)[NA], double aA, double aB)
    {
#pragma unroll
        for(int J = 0; J < NA; J++) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA, bb[J], acc[J], 0, 0, 0);
#define SPL_ROWB(J) if constexpr((J) < NA && NA + (J) < 9) { if((J) < nb) acc[NA + (J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB, bb[J], acc[NA + (J)], 0, 0, 0); }
        SPL_ROWB(0) SPL_ROWB(1) SPL_ROWB(2) SPL_ROWB(3)
#undef SPL_ROWB
    };
    load(b0, aA0, aB0, rowp);
    int s = 0;
#pragma unroll 1
    for(; s + 2 <= nsteps; s += 2)
    {
        load(b1, aA1, aB1, rowp + step);
        mma(b0, aA0, aB0);
        rowp += 2*step;
        // (past the end: the last rows once more, unused)
        load(b0, aA0, aB0, (s + 2 < nsteps) ? rowp : rowp - step);
        mma(b1, aA1, aB1);
    }
    if(s < nsteps) mma(b0, aA0, aB0);
}
// c / wx = (c spl_magic(wx)) >> 16 for 0 <= c < 2^16/wx: the local columns (< 128) of a box up to 128 wide. (An integer
// division is ~40 vector instructions; the gather made four for every pass that held its row, and its waves share a SIMD
// four at a time: those, and nine ds_bpermute, were most of the 18k cycles a batch of three passes took)
__host__ __device__ __forceinline__ int spl_magic(int wx) { return (65536 + wx - 1)/(wx > 0 ? wx : 1); }
// state index of local column c of a pass, -1: not a camera-block variable of this pass (a frame column, x,
// or a variable that is not being optimized)
__device__ __forceinline__
int spl_col_state(const DeviceProblem& P, const NormalDims& nd, int c, int K, int ix0, int iy0, int wx, int xy,
                  int i_state_intrinsics, int i_state_extrinsics, int wx_magic /* spl_magic(wx): c / wx without the division */)
{
    if(c < K)
    {
        const int cy = (c*wx_magic) >> 16, cx = c - cy*wx;
        return (i_state_intrinsics >= 0)
            ? i_state_intrinsics + P.Ncore_state + 2*((iy0 + cy)*P.cfg.spline_Nx + ix0 + cx) + xy : -1;
    }
    const int d = c - K;
    if(d < 4)           return (i_state_intrinsics >= 0 && d < P.Ncore_state) ? i_state_intrinsics + d : -1;
    if(d < 10)          return (i_state_extrinsics >= 0) ? i_state_extrinsics + (d - 4) : -1;
    if(d < SPL_NDENSE)  return nd.Nwarp ? nd.i_state_warp + (d - 10) : -1;
    return -1;
}
// (not inlined: what it needs in registers should not count against the kernel's usual path)
__device__ __noinline__
void spl_rows_fallback(const NormalDims& nd, const OpDev& O, int r_first, int r1,
                       const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    for(int r = r_first; r < r1; r += blockDim.x) rows_generic_row(nd, O, r, r1, Jp, Ji);
}
// can an observation of this problem take the fallback above? (The whole grid under one board is the largest box)
static bool spl_fallback_possible(const DeviceProblem& P)
{
    const int order = P.cfg.spline_order, T = SPL_SUB_MAX - order;
    const int nsx = std::max(1, ((int)P.cfg.spline_Nx - order + T - 1)/T), nsy = std::max(1, ((int)P.cfg.spline_Ny - order + T - 1)/T);
    return nsx*nsy > SPL_MAXSUB || P.W*P.H > 1024;
}
bool splined_needs_repro_rows(const DeviceProblem& P)
{
    if(P.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC) return false;
    // discrete points whose rows' patch columns move with every evaluation, or a board that can cover more than the
    // sub-boxes hold
    return (P.Nobs_point > 0 && P.Ndist_state > 0) || (P.Nobs_board > 0 && P.Nframes > 0 && spl_fallback_possible(P));
}
#ifndef SPL_WAVES_PER_EU
#define SPL_WAVES_PER_EU 2
#endif
__device__ __forceinline__ void rows_pairs_body(const NormalDims& nd, const OpDev& O, int row0, int row1, const int32_t* __restrict__ Jp,
                                                const int32_t* __restrict__ Ji, double* __restrict__ row_part, int block);
__device__ __forceinline__ void spl_compact_body(const DeviceProblem& P, const NormalDims& nd, const OpDev& O, int* __restrict__ lds_c, const int* __restrict__ nd_lim);
// Riding along behind the frames' workgroups (round 5; they were launches of their own on the side stream, behind a fork
// that cost the main stream 8 us): `npairs_extra` workgroups of rows_pairs_body() - the regularization rows from
// pairs_row0 on, which write the camera block's A and g: nothing this kernel's own workgroups touch (never where they can
// fall back to row-by-row atomics: the launcher sees to it) - and, if compact_extra, one of spl_compact_body()
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPL_WAVES_PER_EU)))
void assemble_splined_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan,
                             const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                             int npairs_extra, int pairs_row0, int compact_extra)
{
    if(opref_skip(R)) return;
    __shared__ __attribute__((aligned(16))) double Jd[SPL_LDS_DOUBLES];     // [rows][LD] of a pass
    if((int)blockIdx.x >= 2*P.Nframes)
    {
        const int e = (int)blockIdx.x - 2*P.Nframes;
        if(e < npairs_extra)   rows_pairs_body(nd, opref_get(R), pairs_row0, P.Nmeas, Jp, Ji, plan.row_part, e);
        else if(compact_extra) spl_compact_body(P, nd, opref_get(R), (int*)Jd, plan.nd_lim);
        return;
    }
    __shared__ double F[7*SPL_TW];          // the frame rows and the x row of a pass's Gram
    __shared__ double FD[7*6];              // the frame's own block and its part of the gradient, summed over the passes
    __shared__ unsigned char own[1024];     // sub-boxes: which of them a corner belongs to
    __shared__ unsigned short crow[1024];   // ... and its row among the corners of the sub-box being assembled (0xffff: another's)
    __shared__ int s_nown;
    __shared__ double FB[6*SPL_NDENSE];     // the frame rows against the core, the extrinsics and the warp: over an observation's two passes (the warp: over the frame)
    const OpDev& O = opref_get(R);
    const double* __restrict__ Jv = O.Jv;
    const double* __restrict__ x  = O.x;
    // A workgroup per frame and SURFACE (round 4; it was per frame, the two passes one after the other): the x rows touch
    // the x surface's control points only, the y rows the y surface's, so the two passes share nothing they write but
    // the frame's block, its part of the gradient and the core / extrinsics / warp columns of Bt - to each of which
    // each of the two adds ONE number, atomically, onto zero: a + b is b + a. Twice the workgroups of half the length:
    // nothing at BASELINE configuration 2 (800 frames on 512 places), and a calibration of 186 frames of close-ups,
    // eight passes each, no longer leaves a quarter of the CUs without a workgroup
    const int f = blockIdx.x >> 1, xy0 = blockIdx.x & 1, t = threadIdx.x;
    const int lane = t & 63, r16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int o0 = __builtin_amdgcn_readfirstlane(plan.frame_obs_begin[f]), o1 = __builtin_amdgcn_readfirstlane(plan.frame_obs_begin[f+1]);
    const int NPTS = P.W*P.H, Nx = P.cfg.spline_Nx;
    const int Ncs = P.Ncore_state;
    if(t < 42) FD[t] = 0.0;
    if(t < 6*SPL_NDENSE) FB[t] = 0.0;
    // (-DSPL_TS: cycles per phase, printed by three of the workgroups)
#ifdef SPL_TS
    long long ts_bbox = 0, ts_zero = 0, ts_scatter = 0, ts_gram = 0, ts_out = 0, ts0 = clock64(), ts1;
    const long long tw0 = wall_clock64(), tc0 = ts0;
#define SPL_TICK(what) { ts1 = clock64(); what += ts1 - ts0; ts0 = ts1; }
#else
#define SPL_TICK(what)
#endif

    for(int o = o0; o < o1; o++)
    {
        // (the observation's record and its box are the same for every lane, and said so: the tile's geometry, the
        //  waves' shares and the loops' bounds stay in scalar registers)
        const BoardObsMeta mv = P.board_meta[o];
        const int4 boxv = ((const int4*)O.spl_box)[o];
        const int m_isi = __builtin_amdgcn_readfirstlane(mv.i_state_intrinsics);
        const int m_ise = __builtin_amdgcn_readfirstlane(mv.i_state_extrinsics);
        const int r0 = __builtin_amdgcn_readfirstlane(mv.i_meas0), r1 = r0 + 2*NPTS;
        // (rowptr[i_meas0 + r] = i_nnz0 + r*nnz_per_row: board_splined_kernel. From the record: two loads fewer in line)
        const int p00 = __builtin_amdgcn_readfirstlane((int)mv.i_nnz0);
        const int L   = __builtin_amdgcn_readfirstlane(mv.nnz_per_row);     // entries per row, the same for all rows of the observation
        // the box of control points under this observation's inliers: board_splined_kernel left it with the rows
        const int4 box  = make_int4(__builtin_amdgcn_readfirstlane(boxv.x), __builtin_amdgcn_readfirstlane(boxv.y),
                                    __builtin_amdgcn_readfirstlane(boxv.z), __builtin_amdgcn_readfirstlane(boxv.w));
        const bool any = (P.Ndist_row > 0) && box.y >= 0;
        const int ox0 = any ? box.x : 0, oy0 = any ? box.z : 0;
        const int owx = any ? box.y - box.x + 1 : 0, owy = any ? box.w - box.z + 1 : 0;
        // sub-boxes (solver_kernels.hpp): one if the box fits the tile, else a grid of them, SPL_SUB_MAX wide and
        // high at most, each owning the corners whose patch starts in its first SPL_SUB_MAX - order columns and rows
        const int order = P.cfg.spline_order, T = SPL_SUB_MAX - order;
        int nsx = 1, nsy = 1;
        if(owx*owy + SPL_NEXTRA > SPL_TW)
        {
            nsx = max(1, (owx - order + T - 1)/T);
            nsy = max(1, (owy - order + T - 1)/T);
        }
        const int nsub = nsx*nsy;
        SPL_TICK(ts_bbox)
        if(nsub > SPL_MAXSUB || (nsub > 1 && NPTS > (int)sizeof(own)))
        {
            if(xy0 == 0)
            {
                if(t == 0) plan.spl_hdr[o] = SplHdr{ 0, 0, -1, -1 };
                // (round 5: its rows go through the pre-rounded sums behind this launch, repro_step_*: the same bits
                //  every time. Without their buffers - never, for a problem that can come here - row by row with atomics)
                if(plan.repro.lvl[0] == NULL) spl_rows_fallback(nd, O, r0 + t, r1, Jp, Ji);
            }
            continue;
        }
        if(nsub > 1)
        {
            // whose corner: from the first control point of its x row (an outlier's, outside the box: nobody's)
            for(int c = t; c < NPTS; c += blockDim.x)
            {
                const int rel  = Ji[p00 + 2*c*L + (Ncs ? 2 : 0)] - (m_isi + Ncs);
                const int knot = rel >> 1, px = knot % Nx - ox0, py = knot / Nx - oy0;
                const int gx = px / T, gy = py / T;
                own[c] = (px >= 0 && py >= 0 && gx < nsx && gy < nsy) ? (unsigned char)(gy*nsx + gx) : (unsigned char)255;
            }
            __syncthreads();
        }
#pragma unroll 1
        for(int isub = 0; isub < nsub; isub++)
        {
        const int sgx = isub % nsx, sgy = isub / nsx;
        const int ix0 = ox0 + ((nsub > 1) ? sgx*T : 0), iy0 = oy0 + ((nsub > 1) ? sgy*T : 0);
        const int wx  = (nsub > 1) ? min(T + order, ox0 + owx - ix0) : owx;
        const int wy  = (nsub > 1) ? min(T + order, oy0 + owy - iy0) : owy;
        const int K   = wx*wy;
        // (the first header says how many there are: wy | nsub << 16)
        // (wx with its reciprocal for the gather: wx | spl_magic(wx) << 8)
        const int wx_magic = spl_magic(wx);
        if(t == 0 && xy0 == 0) *spl_hdr_at(plan, o, isub) = SplHdr{ ix0, iy0, wx | (wx_magic << 8), (isub == 0) ? (wy | (nsub << 16)) : wy };
        // a sub-box's pass is over ITS corners only, packed: their rows in the tile, in corner order (a pass over all
        // the corners with the others' rows left zero is as long as the whole observation's: 0.29 ms more a step with
        // 2 x 2 sub-boxes under every board)
        if(nsub > 1)
        {
            __syncthreads();        // (the previous sub-box's passes are through with crow)
            if(wave == 0)
            {
                int nown = 0;
                for(int cb = 0; cb < NPTS; cb += 64)
                {
                    const int  c    = cb + lane;
                    const bool mine = c < NPTS && own[c] == isub;
                    const unsigned long long mm = __ballot(mine);
                    if(c < NPTS) crow[c] = mine ? (unsigned short)(nown + __popcll(mm & ((1ull << lane) - 1ull))) : (unsigned short)0xffff;
                    nown += __popcll(mm);
                }
                if(lane == 0) s_nown = nown;
            }
            __syncthreads();
        }
        const int nrows = (nsub > 1) ? s_nown : NPTS;   // the pass's rows
        const int NC = K + SPL_NEXTRA;                  // local columns in use
        const int NS = (NC + 15) >> 4;                  // 16-column tiles in use
        const int LD = 16*(NS + 1 - (NS & 1));          // row stride: an odd number of tiles
        const int rows_cap = min((NPTS + 3) & ~3, (SPL_LDS_DOUBLES / LD) & ~3);
        const int lx = K + SPL_NDENSE + 6;              // the x column
        const int fr0 = K + SPL_NDENSE;                 // the first frame column
        // this wave's tile rows
        // (every second workgroup deals the rows the other way round: two workgroups share a CU, and their waves w a SIMD)
        const int wv = ((((int)blockIdx.x >> 8) ^ (int)blockIdx.x) & 1) ? 3 - wave : wave;
        const int Ia = NS - 1 - wv, Ib = wv - (8 - NS);
        const int na = (Ia >= 0) ? Ia + 1 : 0, nb = (Ib >= 0 && Ia >= 0) ? Ib + 1 : 0;
        const unsigned L_magic = (unsigned)((0x100000000ull + (unsigned)L - 1)/(unsigned)L);      // e / L = e L_magic >> 32, e (L-1) < 2^32

        {
            const int xy = xy0;
            // state index -> local column of this pass
            auto local_of = [&](int col) -> int
            {
                if(P.do_optimize_frames && col >= nd.E_state0 && col < nd.E_state0 + nd.NE) return fr0 + (col - (nd.E_state0 + 6*f));
                if(m_isi >= 0 && col >= m_isi && col < m_isi + P.Nintr_state)
                {
                    const int rel = col - m_isi;
                    if(rel < Ncs) return K + rel;
                    const int knot = (rel - Ncs) >> 1;
                    return (knot / Nx - iy0)*wx + (knot % Nx - ix0);
                }
                if(m_ise >= 0 && col >= m_ise && col < m_ise + 6)
                    return K + 4 + (col - m_ise);
                return K + 10 + (col - P.i_state_warp);
            };
            // A board's rows mostly fit the tile at once. When they do not, every chunk of rows makes its own Gram and
            // ADDS it to what is staged (and to F): no accumulator is live while rows are fetched - held across the
            // loop they were spilled on every path, and a reload from scratch waits for every store in flight
            double* __restrict__ G = spl_slot(plan, o, xy, isub);
#pragma unroll 1
            for(int c0 = 0; c0 < max(nrows, 1); c0 += rows_cap)
            {
                constexpr bool first = true;
                spl_d4 acc[9];
                // (the lane's indices made opaque at the head of each phase: what is computed from them is computed in
                //  the phase, not in front of the loop over the observations and carried - spilled - through everything)
                int tq = t;
                asm volatile("" : "+v"(tq));
                const int nr = min(rows_cap, nrows - c0), nr4 = (nr + 3) & ~3;
                // (one box: the entries of the corners c0 .. c0 + nr; sub-boxes: of all the corners, kept if the corner's
                //  packed row is one of this chunk's)
                const int pbase = p00 + ((nsub == 1 ? 2*c0 : 0) + xy)*L;
                const int ne = (nsub == 1 ? nr : NPTS)*L;
                // Six entries per thread asked for together (one entry at a time, its column only when the value is
                // not zero, is two memory round trips per entry), the first six BEFORE the tile is cleared: their
                // trip to memory and the clearing overlap
                // (nothing is done with what a load returns before the batch is put away: a select on the spot is a wait on the spot)
                constexpr int EB = 4;
                double v0[EB], v1[EB]; int ci0[EB], ci1[EB], ii0[EB], ii1[EB];
                auto ask = [&](int e0, double (&v)[EB], int (&ci)[EB], int (&ii)[EB])
                {
#pragma unroll
                    for(int u = 0; u < EB; u++)
                    {
                        const int e = e0 + 256*u + tq;
                        const bool ok = e < ne;
                        const int i = ok ? (int)__umulhi((unsigned)e, L_magic) : 0, k = ok ? e - i*L : 0;
                        const int p = pbase + 2*i*L + k;        // (a valid entry either way)
                        ii[u] = ok ? i : -1;
                        v[u]  = Jv[p];
                        ci[u] = Ji[p];
                    }
                };
                auto put = [&](const double (&v)[EB], const int (&ci)[EB], const int (&ii)[EB])
                {
#pragma unroll
                    for(int u = 0; u < EB; u++)
                    {
                        if(ii[u] < 0 || v[u] == 0.0) continue;
                        int row = ii[u];
                        if(nsub > 1) { row = (int)crow[ii[u]] - c0; if(row < 0 || row >= nr) continue; }
                        Jd[row*LD + local_of(ci[u])] = v[u];
                    }
                };
                ask(0, v0, ci0, ii0);
                const double xv = x[r0 + 2*(((nsub == 1) ? c0 : 0) + max(0, min(tq, ((nsub == 1) ? nr : NPTS) - 1))) + xy];
                for(int i = tq; i < nr4*(LD/2); i += blockDim.x) ((double2*)Jd)[i] = make_double2(0.0, 0.0);
                spl_lds_barrier();
                SPL_TICK(ts_zero)
                for(int e0 = 0; e0 < ne; e0 += 2*EB*256)
                {
                    if(e0 + EB*256 < ne)   ask(e0 + EB*256, v1, ci1, ii1);
                    put(v0, ci0, ii0);
                    if(e0 + EB*256 >= ne)  break;
                    if(e0 + 2*EB*256 < ne) ask(e0 + 2*EB*256, v0, ci0, ii0);
                    put(v1, ci1, ii1);
                }
                if(nsub == 1)
                {
                    if(tq < nr) Jd[tq*LD + lx] = xv;
                    for(int i = tq + blockDim.x; i < nr; i += blockDim.x) Jd[i*LD + lx] = x[r0 + 2*(c0 + i) + xy];
                }
                else
                    for(int i = tq; i < NPTS; i += blockDim.x)
                    {
                        const int row = (int)crow[i] - c0;
                        if(row >= 0 && row < nr) Jd[row*LD + lx] = (i == tq) ? xv : x[r0 + 2*i + xy];
                    }
                spl_lds_barrier();
                SPL_TICK(ts_scatter)
                int r16g = r16, kqg = kq;
                asm volatile("" : "+v"(r16g), "+v"(kqg));
                switch(na)
                {
                case 0: if(first) { for(int u = 0; u < 9; u++) acc[u] = spl_d4{0.0, 0.0, 0.0, 0.0}; } break;
                case 1: spl_gram_mfma<1, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 2: spl_gram_mfma<2, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 3: spl_gram_mfma<3, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 4: spl_gram_mfma<4, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 5: spl_gram_mfma<5, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 6: spl_gram_mfma<6, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 7: spl_gram_mfma<7, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                default:spl_gram_mfma<8, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                }
                SPL_TICK(ts_gram)
                // out: the whole lower triangle is staged - the camera-block rows and the x row for the gather - with
                // stores nobody here waits for; the six frame rows and the x row also go to F, for the second half of this
                // (the lane's coordinates made opaque here: left alone, the compiler computes the 36 store addresses once,
                //  in front of the loop over the observations, and keeps them - the kernel spills)
                int kq_o = kq, r16_o = r16;
                asm volatile("" : "+v"(kq_o), "+v"(r16_o));
                const bool add = c0 > 0;
                auto out_tile = [&](const spl_d4& a4, int I, int J)
                {
                    const int col = 16*J + r16_o;
#pragma unroll
                    for(int vv = 0; vv < 4; vv++)
                    {
                        const int row = 16*I + kq_o + 4*vv;
                        if(row < NC && col <= row)
                        {
                            double* __restrict__ g = &G[spl_tri(row) + col];
                            *g = add ? *g + a4[vv] : a4[vv];
                            if(row >= fr0)
                            {
                                double* __restrict__ ff = &F[(row - fr0)*SPL_TW + col];
                                *ff = add ? *ff + a4[vv] : a4[vv];
                            }
                        }
                    }
                };
                if(!add)
                {
#pragma unroll
                    for(int u = 0; u < 9; u++)
                    {
                        if(u < na)           out_tile(acc[u], Ia, u);
                        else if(u - na < nb) out_tile(acc[u], Ib, u - na);
                    }
                }
                else
                {
#pragma unroll
                    for(int u = 0; u < 9; u++)
                    {
                        if(u < na)           out_tile(acc[u], Ia, u);
                        else if(u - na < nb) out_tile(acc[u], Ib, u - na);
                    }
                }
                // (the tile is cleared again only after everybody is through with it)
                if(c0 + rows_cap < nrows) spl_lds_barrier();
            }
            spl_lds_barrier();
            // what belongs to the frame: rows K+12 .. K+17 against the camera-block columns (Bt) and against each
            // other (D_f); the x row against the frame columns (g_f). Element (ia, col) is the same thread's in every
            // pass and observation: its read-modify-writes of one address follow each other in program order
            int tf = t;
            asm volatile("" : "+v"(tf));
            if(P.do_optimize_frames)
                for(int e = tf; e < 7*SPL_TW; e += blockDim.x)
                {
                    const int ia = e >> 7, col = e & (SPL_TW - 1);      // ia 6: the x row
                    const int row = fr0 + ia;
                    const int ib = col - fr0;
                    if(col > row || (ia == 6 && (ib < 0 || ib >= 6))) continue;
                    const double vv = F[ia*SPL_TW + col];
                    if(vv == 0.0) continue;
                    if(ib >= 0)        FD[ia*6 + ib] += vv;                     // (ia 6: g_f)
                    else if(col >= K)  FB[ia*SPL_NDENSE + (col - K)] += vv;
                    else
                    {
                        const int cs = spl_col_state(P, nd, col, K, ix0, iy0, wx > 0 ? wx : 1, xy, m_isi, m_ise, wx_magic);
                        if(cs < 0) continue;
                        // a knot's column is written by this pass of this observation and by nobody else - or, cut
                        // into sub-boxes, by the passes of those that hold it, one after the other (the barrier below)
                        double* __restrict__ dst = &O.Bt[(size_t)(6*f + ia)*nd.Nc + state_to_SE(nd, cs)];
                        if(nsub == 1) *dst = vv; else *dst += vv;
                    }
                }
            if(nsub > 1) __syncthreads();
            // (F is written again after the next pass's barriers)
            SPL_TICK(ts_out)
        }
        }   // isub
        // the observation's core and extrinsics columns of Bt: one addition each, nothing read back (an atomic one, as
        // below). The warp's columns wait for the frame's last observation
        spl_lds_barrier();
        if(P.do_optimize_frames && t < 6*SPL_NDENSE)
        {
            const int ia = t / SPL_NDENSE, d = t - ia*SPL_NDENSE;
            if(d < 10 && FB[t] != 0.0)
            {
                const int cs = spl_col_state(P, nd, d, 0, 0, 0, 1, 0, m_isi, m_ise, 0);   // (a column past the control points: their box does not matter)
                if(cs >= 0) atomicAdd(&O.Bt[(size_t)(6*f + ia)*nd.Nc + state_to_SE(nd, cs)], FB[t]);
                FB[t] = 0.0;
            }
        }
    }
    // the frame's block and gradient: one addition each (an atomic one: an observation that went row by row - too many
    // knots - adds to the same entries with atomics, possibly still in flight)
    __syncthreads();
    if(P.do_optimize_frames && t < 6*SPL_NDENSE)
    {
        const int ia = t / SPL_NDENSE, d = t - ia*SPL_NDENSE;
        if(d >= 10 && nd.Nwarp && FB[t] != 0.0) atomicAdd(&O.Bt[(size_t)(6*f + ia)*nd.Nc + state_to_SE(nd, nd.i_state_warp + (d - 10))], FB[t]);
    }
    if(P.do_optimize_frames && t < 42)
    {
        const int ia = t / 6, ib = t - 6*ia;
        if(ia == 6) { if(FD[t] != 0.0) atomicAdd(&O.g[nd.E_state0 + 6*f + ib], FD[t]); }
        else
        {
            const double vv = (ib <= ia) ? FD[ia*6 + ib] : FD[ib*6 + ia];
            if(vv != 0.0) atomicAdd(&O.D[(size_t)f*36 + ia*6 + ib], vv);
        }
    }
#ifdef SPL_TS
    if((f == 0 || f == 400 || f == 799) && (t == 0 || t == 255))
        printf("splined assembly f %d t %d: bbox %lld zero %lld scatter %lld gram %lld out %lld cycles\n", f, t, ts_bbox, ts_zero, ts_scatter, ts_gram, ts_out);
    // (the workgroup's place in time: the constant 100 MHz clock at its start and end, and its own cycles)
    if((f % 100 == 0 || f == 255 || f == 256 || f == 511 || f == 512 || f == 799) && t == 0)
        printf("splined assembly f %d: wall %lld .. %lld (x10 ns), %lld cycles\n", f, tw0, wall_clock64(), clock64() - tc0);
#endif
}

#define SPLG_WAVES 8
#define SPLG_BATCH 3      // passes whose loads are in flight together (4: 104 registers with the sub-boxes' loop, and two of these
                          // workgroups and a SYRK workgroup no longer share a CU's registers: the launch beside the SYRK 122 us instead of 86)
// rows of the camera block that are knots (a workgroup each), and the others + the x row (SPLG_E workgroups each)
__host__ __device__ inline int splg_nknotrows(const DeviceProblem& P) { return P.Nintr_state > 0 ? P.Ncameras_intrinsics*(P.Nintr_state - P.Ncore_state) : 0; }
__host__ __device__ inline int splg_ndense(const DeviceProblem& P, const NormalDims& nd) { return nd.Nc + 1 - splg_nknotrows(P); }
__global__ __launch_bounds__(64*SPLG_WAVES)
// window > 0 (a launch of the knots' rows alone): a wave's copy of the row is its last window+1 columns and the
// camera's core - a knot's row of A holds nothing else: two control points meet in a Gram only if some corner's
// (order+1)^2 patch holds both, so only within `order` knots of each other either way, and the lower triangle is the
// part at or before the row: window = 2 (order Nx + order) columns. (The whole row, 1207 doubles a wave, was 77 KB of
// LDS a workgroup - two workgroups a CU, 12 of its 20 us clearing and adding up zeros.) block0: the first row's number
__global__ __launch_bounds__(64*SPLG_WAVES)
void assemble_splined_gather_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nwaves, int block0, int window)
{
    if(opref_skip(R)) return;
    extern __shared__ double accs[];        // [nwaves][stride]
    const OpDev& O = opref_get(R);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int stride = (window > 0) ? window + 1 + 4 : nd.Nc + 1;
    const int Nx = P.cfg.spline_Nx, Ncs = P.Ncore_state;
    const int nknot = splg_nknotrows(P), nintr = P.Nintr_state > 0 ? P.Ncameras_intrinsics*P.Nintr_state : 0;
    const int blk = block0 + (int)blockIdx.x;
    // the row, and the observations this workgroup walks
    int r, obs0, obs1, dense = -1, chunk = 0;
    if(blk < nknot)
    {
        const int per = P.Nintr_state - Ncs, ic = blk / per;
        r = ic*P.Nintr_state + Ncs + (blk - ic*per);
        obs0 = 0; obs1 = P.Nobs_board;
    }
    else
    {
        dense = (blk - nknot) / SPLG_E; chunk = (blk - nknot) % SPLG_E;
        const int ncore = P.Nintr_state > 0 ? P.Ncameras_intrinsics*Ncs : 0;
        r = (dense < ncore) ? (dense / Ncs)*P.Nintr_state + dense % Ncs : nintr + (dense - ncore);
        const int per = (P.Nobs_board + SPLG_E - 1)/SPLG_E;
        obs0 = min(P.Nobs_board, chunk*per); obs1 = min(P.Nobs_board, obs0 + per);
    }
    for(int i = t; i < nwaves*stride; i += blockDim.x) accs[i] = 0.0;
    __syncthreads();
    const bool xrow = (r == nd.Nc);
    const int  sr   = xrow ? -1 : (S_to_state(nd, r));     // state index of the row
#ifdef SPLG_TS
    long long gts0 = clock64(), gts_match = 0, gtq; int gn_match = 0, gn_batch = 0;
#endif
    if(wave < nwaves)
    {
        double* __restrict__ acc = accs + wave*stride;
        const int per = (obs1 - obs0 + nwaves - 1)/nwaves;
        const int w0 = min(obs1, obs0 + wave*per), w1 = min(obs1, w0 + per);
        for(int ob = w0; ob < w1; ob += 64)
        {
            // lane l: does observation ob + l hold the row, and where (local row of the x pass, of the y pass)
            // (an observation cut into sub-boxes has a pass, a header and a staged triangle for each: first the first
            //  sub-box of the 64 observations, then the second of those that have one, ...)
            const int o = ob + lane;
            int isi = -1, ise = -1, nsub_l = 0;
            SplHdr h0 = { 0, 0, -1, -1 };
            if(o < w1)
            {
                h0 = plan.spl_hdr[o];
                isi = P.board_meta[o].i_state_intrinsics; ise = P.board_meta[o].i_state_extrinsics;
                if(h0.wx >= 0) { nsub_l = h0.wy >> 16; h0.wy &= 0xffff; }
            }
            auto passes_of = [&](const int isub, const SplHdr hraw) __attribute__((always_inline))
            {
            int lr0 = -1, lr1 = -1, kind = 2;       // kind 0: a knot's row, 1: a core row, 2: the others
            SplHdr h = hraw;
            int hmagic = 0;
            if(hraw.wx >= 0) { h.wx = hraw.wx & 0xff; hmagic = hraw.wx >> 8; }
            if(isub < nsub_l)
            {
                if(hraw.wx >= 0)
                {
                    const int K = h.wx*h.wy;
                    if(xrow) lr0 = lr1 = K + SPL_NDENSE + 6;
                    else if(r >= nd.S_split) lr0 = lr1 = K + 10 + (r - nd.S_split);       // (a warp term: the splined models keep the stationary partition)
                    else if(ise >= 0 && sr >= ise && sr < ise + 6) lr0 = lr1 = K + 4 + (sr - ise);
                    else if(isi >= 0 && sr >= isi && sr < isi + P.Nintr_state)
                    {
                        const int rel = sr - isi;
                        if(rel < Ncs) { lr0 = lr1 = K + rel; kind = 1; }
                        else
                        {
                            const int knot = (rel - Ncs) >> 1;
                            const int ax = knot % Nx - h.ix0, ay = knot / Nx - h.iy0;
                            kind = 0;
                            if(ax >= 0 && ax < h.wx && ay >= 0 && ay < h.wy)
                            {
                                if((rel - Ncs) & 1) lr1 = ay*h.wx + ax; else lr0 = ay*h.wx + ax;
                            }
                        }
                    }
                }
            }
            unsigned long long mm = __ballot(lr0 >= 0 || lr1 >= 0);
#ifdef SPLG_TS
            gn_match += __popcll(mm); gtq = clock64();
#endif
            while(mm)
            {
#ifdef SPLG_TS
                gn_batch++;
#endif
                // up to SPLG_BATCH observations: every load first, then the sums in order
                double v[SPLG_BATCH][2][2], vc[SPLG_BATCH][2];
                int    cs[SPLG_BATCH][2], csc[SPLG_BATCH], Kk[SPLG_BATCH];
                bool   on[SPLG_BATCH][2];
                // (nothing is done with what a load returns before all of a batch's loads are out - a select on the spot is a
                //  wait on the spot, and the 18 loads of three matches were most of 18 trips to memory: 20k cycles a batch.
                //  Which of the values count: a bit each)
                unsigned wanted = 0u;         // (six bits a match)
#pragma unroll
                for(int k = 0; k < SPLG_BATCH; k++)
                {
                    const bool have = mm != 0ull;
                    const int src = __builtin_amdgcn_readfirstlane(have ? __ffsll((long long)mm) - 1 : 0);
                    if(have) mm &= mm - 1;
                    const int oo  = ob + src;
                    // (one lane's values for the wave: v_readlane, not nine trips through the LDS crossbar)
                    const int ix0 = __builtin_amdgcn_readlane(h.ix0, src), iy0 = __builtin_amdgcn_readlane(h.iy0, src);
                    const int wx  = __builtin_amdgcn_readlane(h.wx, src),  wy  = __builtin_amdgcn_readlane(h.wy, src);
                    const int wxm = __builtin_amdgcn_readlane(hmagic, src);
                    const int si  = __builtin_amdgcn_readlane(isi, src), se = __builtin_amdgcn_readlane(ise, src), kd = __builtin_amdgcn_readlane(kind, src);
                    const int l0  = __builtin_amdgcn_readlane(lr0, src), l1 = __builtin_amdgcn_readlane(lr1, src);
                    const int K   = wx*wy;
                    Kk[k]  = K;
                    csc[k] = (have && kd == 0 && lane < Ncs && si >= 0) ? si + lane : -1;
#pragma unroll
                    for(int it = 0; it < 2; it++)
                    {
                        const int lc = lane + 64*it;
                        // the column's variable: the same in both passes but for the surface of a knot
                        int c = spl_col_state(P, nd, lc, K, ix0, iy0, wx > 0 ? wx : 1, 0, si, se, wxm);
                        if(kd == 1 && lc < K) c = -1;                      // a core row: the knots are above the diagonal
                        if(xrow && lc == K + SPL_NDENSE + 6) c = -2;        // |x|^2
                        cs[k][it] = have ? c : -1;
                    }
#pragma unroll
                    for(int xy = 0; xy < 2; xy++)
                    {
                        const int lr = xy ? l1 : l0;
                        on[k][xy] = have && lr >= 0;
                        const double* __restrict__ Gp = spl_slot(plan, oo, xy, isub);
#pragma unroll
                        for(int it = 0; it < 2; it++)
                        {
                            const int lc = lane + 64*it;
                            // (always a load, from an address that is always valid: a load under a condition is a branch and a wait)
                            const bool want = on[k][xy] && lc <= lr && cs[k][it] != -1;
                            v[k][xy][it] = Gp[want ? spl_tri(lr) + lc : 0];
                            if(want) wanted |= 1u << (6*k + 2*xy + it);
                        }
                        // a knot's row against the core: below it in local order
                        {
                            const bool want = on[k][xy] && csc[k] >= 0;
                            vc[k][xy] = Gp[want ? spl_tri(K + lane) + lr : 0];
                            if(want) wanted |= 1u << (6*k + 4 + xy);
                        }
                    }
                }
#pragma unroll
                for(int k = 0; k < SPLG_BATCH; k++)
#pragma unroll
                    for(int xy = 0; xy < 2; xy++)
                    {
                        if(!on[k][xy]) continue;
#pragma unroll
                        for(int it = 0; it < 2; it++)
                        {
                            const int c = cs[k][it];
                            const double vv = ((wanted >> (6*k + 2*xy + it)) & 1u) ? v[k][xy][it] : 0.0;
                            if(c == -1 || vv == 0.0) continue;
                            if(c == -2) { acc[nd.Nc] += vv; continue; }
                            const int se = state_to_SE(nd, c + ((lane + 64*it < Kk[k]) ? xy : 0));
                            if(window <= 0) acc[se] += vv;
                            else
                            {
                                // (a control point further away than a patch reaches: a structural zero that is not one)
                                const int pw = se - (r - window);
                                if(pw >= 0) acc[pw] += vv; else O.scalars[SC_BAD_STRUCTURE] = 1.0;
                            }
                        }
                        const double vcc = ((wanted >> (6*k + 4 + xy)) & 1u) ? vc[k][xy] : 0.0;
                        if(csc[k] >= 0 && vcc != 0.0)
                        {
                            if(window <= 0) acc[state_to_SE(nd, csc[k])] += vcc;
                            else            acc[window + 1 + lane] += vcc;        // (csc = the camera's core + lane)
                        }
                    }
            }
#ifdef SPLG_TS
            gts_match += clock64() - gtq;
#endif
            };
            // (the first sub-box - almost always the only one - exactly as before there were any)
            passes_of(0, h0);
#ifndef SPLG_NO_SUB
            if(__any(nsub_l > 1))
                for(int isub = 1; __any(isub < nsub_l); isub++)
                {
                    SplHdr h = { 0, 0, -1, -1 };
                    if(isub < nsub_l) h = *spl_hdr_at(plan, o, isub);
                    passes_of(isub, h);
                }
#endif
        }
    }
#ifdef SPLG_TS
    if(lane == 0 && (wave == 0 || wave == 5) && window > 0 && (blk % 149 == 0))
        printf("gather row %d wave %d: %lld cycles to the barrier, %lld of them in %d matches (%d batches)\n", blk, wave, clock64() - gts0, gts_match, gn_match, gn_batch);
#endif
    __syncthreads();
    // the waves' copies, in wave order
    for(int c = t; c < stride; c += blockDim.x)
    {
        double s = 0.0;
        for(int w = 0; w < nwaves; w++) s += accs[w*stride + c];
        if(dense >= 0) { plan.spl_part[((size_t)dense*SPLG_E + chunk)*stride + c] = s; continue; }
        int col = c;
        if(window > 0)
        {
            const int per = P.Nintr_state - Ncs, ic = blk / per;
            col = (c <= window) ? r - window + c : ic*P.Nintr_state + (c - window - 1);
        }
        if(s != 0.0 && col >= 0 && col <= r) O.A[(size_t)r*nd.Nc + col] += s;
    }
}
// The rows of the camera block that are control points, the other way round (round 4; the kernel above still takes the
// rows every pass holds). A control point's row of A's lower triangle has (order)(2 order + 1) + order + 1 places that
// can hold anything - the control points of its surface up to `order` back in either direction, 25 for order 3 - and the
// camera's core: one LANE per place, and a pass that holds the row is ONE load for the wave (half a wave: lanes 32..63
// take the next pass): G[tri(lr) + lr + dy wx + dx]. No LDS copy of the row, no search for what a column is, nothing
// read that is a structural zero; each lane adds its place's entries in a fixed order (its half's passes in order, the
// two halves, then the waves in order), the same bits every time. A wave scans 64 observations' headers at a time and
// leaves those that hold the row in an LDS list; SPLK_INFLIGHT entries a half are asked for together.
// (The row-per-workgroup gather above spent 18-20k cycles on a batch of three passes - 128 local columns looked up,
//  loaded and added through LDS for the 25 that count: 79 us at BASELINE configuration 2, 108 on a real calibration)
#define SPLK_WAVES    4
#define SPLK_INFLIGHT 4
struct SplkMatch { int slot; int lr; int wx; int ax_wy; };  // the pass's staged triangle (index of its slot), the row, the box (wx | which allocation << 16; ax | wy << 16)
__global__ __launch_bounds__(64*SPLK_WAVES)
void assemble_splined_gather_knots_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan)
{
    if(opref_skip(R)) return;
    __shared__ SplkMatch list[SPLK_WAVES][64];
    __shared__ double    part[SPLK_WAVES][32];
    const OpDev& O = opref_get(R);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int Nx = P.cfg.spline_Nx, Ncs = P.Ncore_state, order = P.cfg.spline_order;
    // the row: control point (kx, ky) of surface s of camera ic
    const int blk = blockIdx.x;
    const int per = P.Nintr_state - Ncs, ic = blk / per, rel = blk - ic*per;
    const int r   = ic*P.Nintr_state + Ncs + rel;                   // its index in the camera block
    const int sr  = S_to_state(nd, r);
    const int knot = rel >> 1, s = rel & 1, kx = knot % Nx, ky = knot / Nx;
    // this lane's place: (dx, dy), dy < 0 any dx, dy == 0 dx <= 0; or a core variable; or none
    const int ndx = 2*order + 1, nback = order*ndx, nplaces = nback + order + 1;
    int dx = 0, dy = 0, core = -1;
    bool place = false;
    if(j < nback)          { dy = -order + j / ndx; dx = -order + j % ndx; place = true; }
    else if(j < nplaces)   { dy = 0; dx = -order + (j - nback); place = true; }
    else if(j < nplaces + Ncs) core = j - nplaces;
    place = place && kx + dx >= 0 && kx + dx < Nx && ky + dy >= 0;
    double acc = 0.0;

    const int Nobs = P.Nobs_board;
    const int per_wave = (Nobs + SPLK_WAVES - 1)/SPLK_WAVES;
    const int w0 = min(Nobs, wave*per_wave), w1 = min(Nobs, w0 + per_wave);
    SplkMatch* __restrict__ mine = list[wave];
    for(int ob = w0; ob < w1; ob += 64)
    {
        const int o = ob + lane;
        int nsub_l = 0;
        SplHdr h0 = { 0, 0, -1, -1 };
        bool cam = false;
        if(o < w1)
        {
            h0 = plan.spl_hdr[o];
            const int isi = P.board_meta[o].i_state_intrinsics;
            cam = isi >= 0 && sr >= isi && sr < isi + P.Nintr_state;
            if(h0.wx >= 0) { nsub_l = h0.wy >> 16; h0.wy &= 0xffff; }
            if(!cam) nsub_l = 0;
        }
        for(int isub = 0; __any(isub < nsub_l); isub++)
        {
            SplHdr h = h0;
            if(isub > 0 && isub < nsub_l) h = *spl_hdr_at(plan, o, isub);
            bool hit = false;
            int  lr = 0, wx = 1, wy = 1;
            if(isub < nsub_l && h.wx >= 0)
            {
                wx = h.wx & 0xff; wy = h.wy;
                const int ax = kx - h.ix0, ay = ky - h.iy0;
                hit = ax >= 0 && ax < wx && ay >= 0 && ay < wy;
                lr  = ay*wx + ax;
            }
            // the passes that hold the row, in observation order, into the list
            const unsigned long long mm = __ballot(hit);
            const int n = __popcll(mm);
            if(hit)
            {
                const int at = __popcll(mm & ((1ull << lane) - 1ull));
                const double* __restrict__ G = spl_slot(plan, o, s, isub);
                mine[at] = SplkMatch{ (int)((G - ((isub > 0) ? plan.chunk_extra : plan.chunk_part))/SPL_TRI), lr, wx | ((isub > 0) ? 0x10000 : 0), (kx - h.ix0) | (wy << 16) };
            }
            // (the list is this wave's own: no barrier, the LDS keeps a wave's accesses in order)
            for(int i0 = 0; i0 < n; i0 += 2*SPLK_INFLIGHT)
            {
                double v[SPLK_INFLIGHT];
                bool   on[SPLK_INFLIGHT];
#pragma unroll
                for(int b = 0; b < SPLK_INFLIGHT; b++)
                {
                    const int i = i0 + 2*b + half;
                    on[b] = false; v[b] = 0.0;
                    if(i < n)
                    {
                        const SplkMatch m = mine[i];
                        const int mwx = m.wx & 0xffff;
                        // (a second, third.. sub-box's triangle is in the other allocation: spl_slot())
                        const double* __restrict__ G = ((m.wx & 0x10000) ? plan.chunk_extra : plan.chunk_part) + (size_t)m.slot*SPL_TRI;
                        const int ax = m.ax_wy & 0xffff;
                        if(place && ax + dx >= 0 && ax + dx < mwx && m.lr + dy*mwx >= 0)
                        {
                            on[b] = true;
                            v[b]  = G[spl_tri(m.lr) + m.lr + dy*mwx + dx];
                        }
                        else if(core >= 0)
                        {
                            on[b] = true;
                            v[b]  = G[spl_tri(mwx*(m.ax_wy >> 16) + core) + m.lr];
                        }
                    }
                }
#pragma unroll
                for(int b = 0; b < SPLK_INFLIGHT; b++) if(on[b]) acc += v[b];
            }
        }
    }
    // the halves, then the waves, in order
    acc += __shfl(acc, (lane + 32) & 63);
    if(lane < 32) part[wave][lane] = acc;
    __syncthreads();
    if(t < 32)
    {
        double total = 0.0;
        for(int w = 0; w < SPLK_WAVES; w++) total += part[w][t];
        if(total != 0.0)
        {
            const int col = (core >= 0) ? ic*P.Nintr_state + core : r + 2*(dy*Nx + dx);
            if((place || core >= 0) && col >= 0 && col <= r) O.A[(size_t)r*nd.Nc + col] += total;
        }
    }
}
// The regularization rows of a splined model (regularization_splined_kernel in kernels.hip): per knot a radial
// and a tangential row on the knot's two variables, then one row per centre-pixel variable, then unity_cam01.
// Rows 2 i and 2 i + 1 share their columns; no two PAIRS do. One lane per pair, the pair's rows one after the
// other, plain adds: the same bits every time. (Row by row with atomics, the two rows of a knot race.) |x|^2 of a
// workgroup's rows goes to row_part[blockIdx.x]; the combine kernel adds those in order. Lower triangle of A only
__device__ __forceinline__
void rows_pairs_body(const NormalDims& nd, const OpDev& O, int row0, int row1,
                     const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, double* __restrict__ row_part, int block)
{
    const double* __restrict__ Jv = O.Jv;
    const int i = block*blockDim.x + threadIdx.x;
    double n2 = 0.0;
    for(int k = 0; k < 2; k++)
    {
        const int r = row0 + 2*i + k;
        if(r >= row1) break;
        const double xr = O.x[r];
        n2 += xr*xr;
        const int p0 = Jp[r], p1 = Jp[r+1];
        for(int p = p0; p < p1; p++)
        {
            const int ci = Ji[p];
            const double vi = Jv[p];
            if((unsigned)ci >= (unsigned)nd.Nstate) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }
            const int si = state_to_SE(nd, ci);
            if(si < 0) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }       // a regularization row has camera-block variables only
            O.g[ci] += vi*xr;
            for(int q = p0; q < p1; q++)
            {
                const int cj = Ji[q];
                if((unsigned)cj >= (unsigned)nd.Nstate) continue;
                const int sj = state_to_SE(nd, cj);
                if(sj >= 0 && sj <= si) O.A[(size_t)si*nd.Nc + sj] += vi*Jv[q];
            }
        }
    }
    for(int off = 32; off > 0; off >>= 1) n2 += __shfl_down(n2, off);
    __shared__ double part[4];
    if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n2;
    __syncthreads();
    if(threadIdx.x == 0) row_part[block] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256)
void rows_pairs_kernel(NormalDims nd, OpRef R, int row0, int row1,
                       const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, double* __restrict__ row_part)
{
    if(opref_skip(R)) return;
    rows_pairs_body(nd, opref_get(R), row0, row1, Jp, Ji, row_part, blockIdx.x);
}
// Which control points does a board cover at this point? (Round 5.) The others have their regularization rows and
// nothing else: a 2 x 2 block of the camera block each, coupled to nothing - and in the Cholesky of the camera block
// every one of their columns is a pivot of the sequential chain all the same: at BASELINE configuration 2 (30 x 20
// control points over 150 degrees, boards 4 m away) 277 of the 600 control points, 554 of 1206 pivots. So the
// camera block is put in the order [coupled variables | isolated pairs] (each part in its own order), the reduction
// writes the coupled part as a dense matrix of its own and the pairs' blocks beside it (schur_reduce_body), and the
// factorization's launches past the coupled part's last panel find nothing to do (lchol_plan()).
// One workgroup: the observations' boxes (OpDev::spl_box, left by board_splined_kernel) marked in LDS, then a scan over
// the camera block's variables. cperm: [Nc] position -> variable | [Nc] variable -> position | [1] the coupled ones
#define SPLC_T 256
__device__ __forceinline__
void spl_compact_body(const DeviceProblem& P, const NormalDims& nd, const OpDev& O, int* __restrict__ lds_c /* [Nknots_all] used | [SPLC_T/64] wave totals | spl_compact_lds_ints(): the dissection's scratch */,
                      const int* __restrict__ nd_lim /* NdLimits on the device; NULL: no dissection */)
{
    if(O.cperm == NULL) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int Nx = P.cfg.spline_Nx, Ny = P.cfg.spline_Ny, NK = Nx*Ny;
    const int nknots = P.Ncameras_intrinsics*NK;
    int* __restrict__ used = lds_c;
    int* __restrict__ wtot = lds_c + nknots;
    // (the dissection's scratch behind the wave totals: [0] the widest box | [1..4] the plan | [8 ..] covered control points per grid column | 64-bit scan words)
    int* __restrict__ ndw = wtot + SPLC_T/64;
    for(int i = t; i < nknots; i += SPLC_T) used[i] = 0;
    if(t < 8 + Nx && t < SPLC_T) ndw[t] = 0;
    __syncthreads();
    if(P.Ndist_state > 0 && O.spl_box != NULL)
        for(int o = t; o < P.Nobs_board; o += SPLC_T)
        {
            const int4 box = ((const int4*)O.spl_box)[o];
            const int isi = P.board_meta[o].i_state_intrinsics;
            if(box.y < 0 || isi < 0) continue;                       // no inlier under this observation
            atomicMax(&ndw[0], box.y - box.x + 1);
            const int icam = (isi - P.i_state_intrinsics)/P.Nintr_state;
            for(int iy = box.z; iy <= box.w; iy++)
                for(int ix = box.x; ix <= box.y; ix++)
                    used[icam*NK + iy*Nx + ix] = 1;                  // (everybody writes the same 1)
        }
    __syncthreads();
    // the variables, a run of consecutive ones per thread: coupled unless it is a control point's that no box holds
    const int per = (nd.Nc + SPLC_T - 1)/SPLC_T;
    const int c0 = min(nd.Nc, t*per), c1 = min(nd.Nc, c0 + per);
    auto coupled = [&](int c) -> bool
    {
        const int st = S_to_state(nd, c);
        const int rel = st - P.i_state_intrinsics;
        if(P.Ndist_state <= 0 || rel < 0 || rel >= P.Ncameras_intrinsics*P.Nintr_state) return true;
        const int icam = rel / P.Nintr_state, k = rel - icam*P.Nintr_state - P.Ncore_state;
        if(k < 0) return true;                                       // the core
        return used[icam*NK + (k >> 1)] != 0;
    };
    int mine = 0;
    for(int c = c0; c < c1; c++) mine += coupled(c) ? 1 : 0;
    // exclusive scan of `mine` over the threads: within the wave, then the waves' totals
    int incl = mine;
    for(int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if(lane >= off) incl += v; }
    if(lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = incl - mine, total = 0;
    for(int w = 0; w < SPLC_T/64; w++) { const int v = wtot[w]; if(w < wave) before += v; total += v; }
    // (a camera block with nothing coupled in it cannot be: boards make rows. If it were, nothing is compacted)
    const int n1 = (total > 0) ? total : nd.Nc;
    int* __restrict__ perm = O.cperm;
    int* __restrict__ iperm = O.cperm + nd.Nc;
    int at_c = before, at_i = n1 + (c0 - before);
    for(int c = c0; c < c1; c++)
    {
        const bool cpl = (total > 0) ? coupled(c) : true;
        const int pos = cpl ? at_c++ : at_i++;
        perm[pos] = c; iperm[c] = pos;
    }
    if(t == 0) O.cperm[2*nd.Nc] = n1;

    // ---- the dissection (lchol_nd_*): a strip of grid columns as wide as the widest box less one; A the covered control
    // points left of it, B those right of it, S the strip's and every coupled variable that is no control point
    if(O.ndp == NULL) return;
    int* __restrict__ ndh   = O.ndp;
    int* __restrict__ npos  = ndh + NDH_WORDS;
    int* __restrict__ nperm = npos + nd.Nc;
    int* __restrict__ colcnt = ndw + 8;
    const bool eligible = nd_lim != NULL && P.Ncameras_intrinsics == 1 && total > 0 && P.Ndist_state > 0 && Nx + 8 <= SPLC_T;
    if(eligible && t < Nx)
    {
        int k = 0;
        for(int iy = 0; iy < Ny; iy++) k += used[iy*Nx + t];
        colcnt[t] = k;
    }
    __syncthreads();
    if(t == 0)
    {
        // the best strip there is (what learn_likely_size() provides launches for), and the best one that fits what was provided
        const int std_cost = (n1 + ND_PANEL - 1)/ND_PANEL;
        int ideal = 0, idA = 0, idB = 0, idS = n1, idcost = std_cost;
        int act = 0, bestA = 0, bestB = 0, bestS = n1, bestc0 = 0, bestcost = std_cost, ws = 0;
        if(eligible && ndw[0] >= 2)
        {
            ws = ndw[0] - 1;
            int left = 0, all = 0;
            for(int x = 0; x < Nx; x++) all += colcnt[x];
            int strip = 0;
            for(int x = 0; x < ws && x < Nx; x++) strip += colcnt[x];       // the strip [s0, s0 + ws), s0 = 0 to begin with
            for(int s0 = 0; s0 + ws < Nx; s0++)
            {
                if(s0 > 0) { left += colcnt[s0 - 1]; strip += colcnt[s0 + ws - 1] - colcnt[s0 - 1]; }
                const int ar = 2*left, br = 2*(all - left - strip), sr = n1 - ar - br;
                const int a = (ar + ND_PANEL - 1)/ND_PANEL, b = (br + ND_PANEL - 1)/ND_PANEL, s = (sr + ND_PANEL - 1)/ND_PANEL;
                if(a < 1 || b < 1 || sr < 1) continue;
                // launches on the chain: the rounds of the longer side, the junction, the separator's panels
                const int cost = max(a, b) + 1 + s;
                if(cost < idcost) { idcost = cost; idA = ar; idB = br; idS = sr; ideal = 1; }
                const bool fits = nd_lim[0] > 0 && a <= nd_lim[0] && b <= nd_lim[0] && sr <= nd_lim[1] && ND_PANEL*max(a, b) <= LCH_ND_WMAX;
                if(fits && cost < bestcost) { bestcost = cost; bestA = ar; bestB = br; bestS = sr; bestc0 = s0; act = 1; }
            }
        }
        ndh[NDH_IDEAL_A] = ideal ? idA : 0; ndh[NDH_IDEAL_B] = ideal ? idB : 0; ndh[NDH_IDEAL_NS] = ideal ? idS : 0;
        const int a = (bestA + ND_PANEL - 1)/ND_PANEL, b = (bestB + ND_PANEL - 1)/ND_PANEL;
        ndh[NDH_ACTIVE] = act;
        ndh[NDH_NA] = act ? ND_PANEL*a : 0; ndh[NDH_NB] = act ? ND_PANEL*b : 0; ndh[NDH_NS] = act ? bestS : n1;
        ndh[NDH_NSEFF] = act ? bestS : n1;
        ndh[NDH_ARAW] = act ? bestA : 0; ndh[NDH_BRAW] = act ? bestB : 0;
        ndw[1] = act; ndw[2] = bestc0; ndw[3] = ws; ndw[4] = bestA; ndw[5] = bestB;
    }
    __syncthreads();
    if(!ndw[1]) return;
    {
        const int sc0 = ndw[2], sws = ndw[3], arw = ndw[4], brw = ndw[5];
        const int nA = ndh[NDH_NA], nB = ndh[NDH_NB];
        // class of a coupled variable: 1 A, 2 B, 0 S
        auto cls_of = [&](int c) -> int
        {
            const int st = S_to_state(nd, c);
            const int rel = st - P.i_state_intrinsics;
            if(rel < 0 || rel >= P.Nintr_state) return 0;
            const int k = rel - P.Ncore_state;
            if(k < 0) return 0;
            const int x = (k >> 1) % Nx;
            return (x < sc0) ? 1 : ((x >= sc0 + sws) ? 2 : 0);
        };
        // exclusive scans of the three classes' counts over the threads' runs, in one 64-bit word (21 bits a count)
        unsigned long long mine3 = 0ull;
        for(int c = c0; c < c1; c++)
            if(coupled(c)) mine3 += 1ull << (21*cls_of(c));
        unsigned long long incl3 = mine3;
        for(int off = 1; off < 64; off <<= 1) { const unsigned long long v = __shfl_up(incl3, off); if(lane >= off) incl3 += v; }
        unsigned long long* __restrict__ wt3 = (unsigned long long*)(((size_t)(colcnt + Nx) + 7) & ~(size_t)7);
        __syncthreads();
        if(lane == 63) wt3[wave] = incl3;
        __syncthreads();
        unsigned long long before3 = incl3 - mine3;
        for(int w = 0; w < wave; w++) before3 += wt3[w];
        int at[3] = { (int)(before3 & 0x1fffff), (int)((before3 >> 21) & 0x1fffff), (int)((before3 >> 42) & 0x1fffff) };
        const int base[3] = { nA + nB, 0, nA };
        for(int c = c0; c < c1; c++)
        {
            if(!coupled(c)) { npos[c] = 3 << 28; continue; }
            const int k = cls_of(c), idx = at[k]++;
            npos[c] = (k << 28) | idx;
            nperm[base[k] + idx] = c;
        }
        // the pads: positions without a variable
        for(int p = arw + t; p < nA; p += SPLC_T) nperm[p] = -1;
        for(int p = nA + brw + t; p < nA + nB; p += SPLC_T) nperm[p] = -1;
    }
}
// the plans of both operating points put out of use (the host has changed what it provides launches for: plans made
// against the old limits may not fit the new grids; the points' next reductions go the ordinary way)
__global__ void nd_plans_off_kernel(const OpDev* __restrict__ ops, int Nc)
{
    const OpDev& O = ops[threadIdx.x];
    if(threadIdx.x >= 2 || O.ndp == NULL || O.cperm == NULL) return;
    O.ndp[NDH_ACTIVE] = 0; O.ndp[NDH_NA] = 0; O.ndp[NDH_NB] = 0; O.ndp[NDH_ARAW] = 0; O.ndp[NDH_BRAW] = 0;
    O.ndp[NDH_NS] = O.cperm[2*Nc]; O.ndp[NDH_NSEFF] = O.cperm[2*Nc];
}
hipError_t launch_nd_plans_off(const OpDev* ops, int Nc, hipStream_t stream)
{
    hipLaunchKernelGGL(nd_plans_off_kernel, dim3(1), dim3(64), 0, stream, ops, Nc);
    return hipGetLastError();
}
// ints of LDS spl_compact_body() needs
__host__ __device__ inline size_t spl_compact_lds_ints(int nknots_all, int Nx) { return (size_t)nknots_all + SPLC_T/64 + 8 + ((Nx + 1) & ~1) + 2*(SPLC_T/64) + 8; }
__global__ __launch_bounds__(SPLC_T)
void spl_compact_kernel(DeviceProblem P, NormalDims nd, OpRef R, const int* __restrict__ nd_lim)
{
    if(opref_skip(R)) return;
    extern __shared__ int lds_cc[];
    spl_compact_body(P, nd, opref_get(R), lds_cc, nd_lim);
}

// the SPLG_E parts of a row every pass holds, in order; and |x|^2 of the regularization rows
__global__ __launch_bounds__(256)
void assemble_splined_combine_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nrow_parts)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const int dense = blockIdx.x, stride = nd.Nc + 1;
    const int nintr = P.Nintr_state > 0 ? P.Ncameras_intrinsics*P.Nintr_state : 0;
    const int ncore = P.Nintr_state > 0 ? P.Ncameras_intrinsics*P.Ncore_state : 0;
    const int r = (dense < ncore) ? (dense / P.Ncore_state)*P.Nintr_state + dense % P.Ncore_state : nintr + (dense - ncore);
    for(int c = threadIdx.x; c < stride; c += blockDim.x)
    {
        double s = 0.0;
        for(int e = 0; e < SPLG_E; e++) s += plan.spl_part[((size_t)dense*SPLG_E + e)*stride + c];
        if(r == nd.Nc && c == nd.Nc)
            for(int b = 0; b < nrow_parts; b++) s += plan.row_part[b];
        if(s == 0.0) continue;
        if(r == nd.Nc)
        {
            if(c == nd.Nc) O.scalars[SC_NORM2_X] += s;
            else           O.g[S_to_state(nd, c)] += s;
        }
        else if(c <= r) O.A[(size_t)r*nd.Nc + c] += s;
    }
}

struct ReproAcc { double *A, *Bt, *D, *g, *n2; };      // (g, n2: the solver step's rows; NULL for a bare matrix)
// t, below 2^c in magnitude, added to the three levels at offset i
__device__ __forceinline__ void repro_add(const ReproAcc (&acc)[3], int which, size_t i, double t, int c, int N)
{
    auto pick = [&](const ReproAcc& a) { return which == 0 ? a.A : (which == 1 ? a.Bt : (which == 2 ? a.D : (which == 3 ? a.g : a.n2))); };
    double* const dst[3] = { pick(acc[0]), pick(acc[1]), pick(acc[2]) };
#pragma unroll
    for(int l = 0; l < 3; l++)
    {
        // M = 1.5 2^(c + N): an ulp of 2^(c + N - 52)
        const double M = __longlong_as_double(((long long)(1023 + c + N) << 52) | (1ll << 51));
        const double q = __dadd_rn(__dadd_rn(t, M), -M);
        if(q != 0.0) atomicAdd(&dst[l][i], q);
        t = __dadd_rn(t, -q);
        c += N - 52;
    }
}
// what a kernel that adds through repro_add() needs: the three levels, the columns' maxima, N
struct ReproCtx { ReproAcc acc[3]; const unsigned long long* cmax; int N; };
// one row, by one lane: the products of its entries, pair by pair, through repro_add()
__device__ __forceinline__
void rows_repro_row(const NormalDims& nd, const OpDev& O, int r, const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                    const ReproCtx& rc, int extra_bits, bool with_x = false /* g and |x|^2 too: x is column Nstate of cmax */)
{
    const double* __restrict__ Jv = O.Jv;
    const int N = rc.N;
    const int p0 = Jp[r], p1 = Jp[r+1];
    // the exponent above a column's largest |value|: biased exponent - 1023 + 1
    auto cexp = [&](int c) { return (int)((rc.cmax[c] >> 52) & 0x7ff) - 1022; };
    auto in_range = [&](int c) { return !(c + N > 900 || c + 3*N - 160 < -900); };
    const double xr = with_x ? O.x[r] : 0.0;
    const int    ex = with_x ? cexp(nd.Nstate) : 0;
    if(with_x && xr != 0.0)
    {
        if(in_range(2*ex)) repro_add(rc.acc, 4, 0, __dmul_rn(xr, xr), 2*ex, N); else O.scalars[SC_BAD_STRUCTURE] = 2.0;
    }
    for(int p = p0; p < p1; p++)
    {
        const int    ci = Ji[p];
        const double vi = Jv[p];
        if((unsigned)ci >= (unsigned)nd.Nstate) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }
        if(vi == 0.0) continue;
        const int si = state_to_SE(nd, ci), ei = cexp(ci);
        if(with_x && xr != 0.0)
        {
            if(in_range(ei + ex)) repro_add(rc.acc, 3, (size_t)ci, __dmul_rn(vi, xr), ei + ex, N); else O.scalars[SC_BAD_STRUCTURE] = 2.0;
        }
        for(int q = p0; q < p1; q++)
        {
            const int cj = Ji[q];
            if((unsigned)cj >= (unsigned)nd.Nstate) continue;
            const double t = __dmul_rn(vi, Jv[q]);
            if(t == 0.0) continue;
            const int sj = state_to_SE(nd, cj);
            const int c  = ei + cexp(cj) + extra_bits;
            // (the exponents the levels' constants are made of must exist: columns of ~1e+-100 and smaller are not served)
            if(c + N > 900 || c + 3*N - 160 < -900) { O.scalars[SC_BAD_STRUCTURE] = 2.0; continue; }
            // (the lower triangles of A and of the D blocks: repro_combine_kernel mirrors them)
            if(si >= 0 && sj >= 0)     { if(sj <= si) repro_add(rc.acc, 0, (size_t)si*nd.Nc + sj, t, c, N); }
            else if(si < 0 && sj >= 0) repro_add(rc.acc, 1, (size_t)(-si-1)*nd.Nc + sj, t, c, N);
            else if(si < 0 && sj < 0)
            {
                int bi, ai, di, e0i, bj, aj, dj, e0j;
                E_to_block(nd, -si-1, &bi, &ai, &di, &e0i);
                E_to_block(nd, -sj-1, &bj, &aj, &dj, &e0j);
                if(bi == bj) { if(aj <= ai) repro_add(rc.acc, 2, (size_t)bi*36 + ai*6 + aj, t, c, N); }
                else         O.scalars[SC_BAD_STRUCTURE] = 1.0;      // no row may touch two E blocks
            }
        }
    }
}
// sum over the 32 lanes of this lane's half of the wave, to all of them
__device__ __forceinline__ double half_wave_sum_f64(double v)
{
#pragma unroll
    for(int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// 64 consecutive rows per wave. The rows of a CSR Jacobian come in runs with the SAME
// columns: of a board observation's 2 W H rows, the x rows share one column set and the y
// rows another (fx, cx against fy, cy). Lanes 0..31 take the even rows, lanes 32..63 the
// odd ones. In each half, the rows with the columns of the half's first pending row form a
// group: their products are summed across the half and ONE lane adds them; then the next
// group (the rows past an observation boundary), until no row is pending. A run of 32 rows
// costs the atomics of one. (A bare CSR Jacobian handed to CHOLMOD_factorization(J) has no
// Grams to assemble from: at 1.6 M rows x 24 entries one lane per row is 922 M atomics, 97 ms)
// REPRO (round 5, CHOLMOD_factorization(J) of a bare matrix): the group sums - made across the half-wave in a fixed
// order, whatever the scheduling - go to memory through repro_add(), pre-rounded so that no addition of the atomics
// rounds (launch_assemble_rows); a group sum of up to 32 products of columns i, j is below 2^(c_i + c_j + 5). The
// rows outside runs go one lane a row through the same. x is not looked at (a bare matrix has none)
template<bool REPRO>
__device__ __forceinline__
void rows_generic_wave(const NormalDims& nd, const OpDev& O, int r_first, int row1,
                       const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, const ReproCtx* __restrict__ rc = NULL)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, first = half << 5;
    const int r = r_first + 2*(lane & 31) + half;
    const bool valid = r < row1;
    const int p0 = valid ? Jp[r] : 0, p1 = valid ? Jp[r+1] : 0;
    const int len = p1 - p0;
    const double* __restrict__ Jv = O.Jv;
    const double xr = (valid && !REPRO) ? O.x[r] : 0.0;
    auto cexp = [&](int c) { return (int)((rc->cmax[c] >> 52) & 0x7ff) - 1022; };   // (REPRO) the exponent above a column's largest |value|
    bool todo = valid;
    while(__any(todo))
    {
        const unsigned long long pending = __ballot(todo);
        const unsigned mine = (unsigned)(pending >> first);             // this half's 32 lanes
        const bool active = mine != 0u;
        const int  leader = first + (active ? __ffs(mine) - 1 : 0);
        const int  lp0 = __shfl(p0, leader), llen = active ? __shfl(len, leader) : 0;
        const int  lenmax = max(__shfl(llen, 0), __shfl(llen, 32));
        const int32_t* __restrict__ cols = Ji + lp0;                     // the group's columns
        bool member = todo && len == llen;
        for(int k = 0; k < lenmax; k++)
            if(member && k < llen) member = Ji[p0 + k] == cols[k];
        const bool adder = active && lane == leader;
        // no runs here (regularization rows: every row its own columns): the rows still pending go one lane per row
        if(__popcll(__ballot(member)) < 8)
        {
            if constexpr(REPRO) { if(todo) rows_repro_row(nd, O, r, Jp, Ji, *rc, 0); }
            else                { if(todo) rows_generic_row(nd, O, r, row1, Jp, Ji); }
            return;
        }

        if constexpr(!REPRO)
        {
            const double n2 = half_wave_sum_f64(member ? xr*xr : 0.0);
            if(adder) atomicAdd(&O.scalars[SC_NORM2_X], n2);
        }
        for(int p = 0; p < lenmax; p++)
        {
            const bool inp = member && p < llen;
            const double vi = inp ? Jv[p0 + p] : 0.0;
            bool addp = adder && p < llen;
            int  ci = addp ? cols[p] : 0;
            if((unsigned)ci >= (unsigned)nd.Nstate) { O.scalars[SC_BAD_STRUCTURE] = 1.0; addp = false; ci = 0; }
            const int  si = state_to_SE(nd, ci);
            if constexpr(!REPRO)
            {
                const double gs = half_wave_sum_f64(vi*xr);
                if(addp) atomicAdd(&O.g[ci], gs);
            }
            const int ei = (REPRO && addp) ? cexp(ci) : 0;
            for(int q = p; q < lenmax; q++)
            {
                // (REPRO: the products rounded one by one, then summed over the half-wave in the butterfly's fixed order)
                const double v = half_wave_sum_f64((inp && q < llen) ? __dmul_rn(vi, Jv[p0 + q]) : 0.0);
                if(!addp || q >= llen) continue;
                const int cj = cols[q];
                if((unsigned)cj >= (unsigned)nd.Nstate) continue;       // (flagged when it comes up as p)
                const int sj = state_to_SE(nd, cj);
                if constexpr(REPRO)
                {
                    // the lower triangles of A and of the D blocks only (repro_combine_kernel mirrors them); Bt whole
                    if(v == 0.0) continue;
                    const int c = ei + cexp(cj) + 5;
                    if(c + rc->N > 900 || c + 3*rc->N - 160 < -900) { O.scalars[SC_BAD_STRUCTURE] = 2.0; continue; }
                    const int sh = max(si, sj), sl = min(si, sj);       // (S indices >= 0, E indices < 0)
                    // a row that lists a column twice (p != q, the same variable): the cross term of (v_p + v_q)^2 is
                    // 2 v_p v_q and both halves land on the one diagonal entry - rows_repro_row, which walks the ordered
                    // pairs, adds it twice; so does this (the bound on N has room for it: launch_assemble_rows)
                    const int times = (p != q && si == sj) ? 2 : 1;
                    for(int t = 0; t < times; t++)
                    {
                        if(sl >= 0)                 repro_add(rc->acc, 0, (size_t)sh*nd.Nc + sl, v, c, rc->N);
                        else if(sh >= 0)            repro_add(rc->acc, 1, (size_t)(-sl-1)*nd.Nc + sh, v, c, rc->N);
                        else
                        {
                            int bi, ai, di, e0i, bj, aj, dj, e0j;
                            E_to_block(nd, -si-1, &bi, &ai, &di, &e0i);
                            E_to_block(nd, -sj-1, &bj, &aj, &dj, &e0j);
                            if(bi == bj) repro_add(rc->acc, 2, (size_t)bi*36 + max(ai, aj)*6 + min(ai, aj), v, c, rc->N);
                            else         O.scalars[SC_BAD_STRUCTURE] = 1.0;
                        }
                    }
                    continue;
                }
                // both orientations of the pair, as the row-by-row loop over (p,q) and (q,p) adds them
                for(int o = 0; o < ((p == q) ? 1 : 2); o++)
                {
                    const int s0 = o ? sj : si, s1 = o ? si : sj;
                    if(s0 >= 0 && s1 >= 0)      atomicAdd(&O.A[(size_t)s0*nd.Nc + s1], v);
                    else if(s0 < 0 && s1 >= 0)  atomicAdd(&O.Bt[(size_t)(-s0-1)*nd.Nc + s1], v);
                    else if(s0 < 0 && s1 < 0)
                    {
                        int bi, ai, di, e0i, bj, aj, dj, e0j;
                        E_to_block(nd, -s0-1, &bi, &ai, &di, &e0i);
                        E_to_block(nd, -s1-1, &bj, &aj, &dj, &e0j);
                        if(bi == bj) atomicAdd(&O.D[(size_t)bi*36 + ai*6 + aj], v);
                        else         O.scalars[SC_BAD_STRUCTURE] = 1.0;
                    }
                }
            }
        }
        todo = todo && !member;
    }
}
// ---- the rows of a splined problem that no plan covers, in the solver's step (ReproStep, solver_kernels.hpp) ----
// which rows: the board observations assemble_splined_kernel marked (SplHdr::wx < 0: more than SPL_MAXSUB sub-boxes),
// and the rows [rows_from, rows_to) (discrete points). Thread i: a board row for i < nboard, row rows_from + i - nboard
__device__ __forceinline__
int repro_step_row(const DeviceProblem& P, const AssemblyPlan& plan, int i, int nboard, int rows_from, int rows_to)
{
    if(i < nboard)
    {
        if(plan.spl_hdr == NULL || plan.spl_hdr[i/(2*P.W*P.H)].wx >= 0) return -1;
        return i;
    }
    const int r = rows_from + (i - nboard);
    return (r < rows_to) ? r : -1;
}
// pass 1: every column's largest |value| over those rows (the last column: x); any row at all -> *any
__global__ __launch_bounds__(256)
void repro_step_colmax_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nboard, int rows_from, int rows_to,
                              const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const int r = repro_step_row(P, plan, blockIdx.x*blockDim.x + threadIdx.x, nboard, rows_from, rows_to);
    if(r < 0) return;
    unsigned long long* __restrict__ cmax = plan.repro.cmax;
    *plan.repro.any = 1;
    for(int p = Jp[r]; p < Jp[r+1]; p++)
    {
        const int c = Ji[p];
        if((unsigned)c >= (unsigned)nd.Nstate) continue;
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(O.Jv[p]));
        if(b > cmax[c]) atomicMax(&cmax[c], b);
    }
    const unsigned long long bx = (unsigned long long)__double_as_longlong(fabs(O.x[r]));
    if(bx > cmax[nd.Nstate]) atomicMax(&cmax[nd.Nstate], bx);
}
__device__ __forceinline__ ReproCtx repro_step_ctx(const NormalDims& nd, const ReproStep& rs, int N)
{
    const size_t nA = (size_t)nd.Nc*nd.Nc, nB = (size_t)nd.NE*nd.Nc, nD = (size_t)nd.NEb*36;
    ReproCtx rc;
    for(int l = 0; l < 3; l++)
    {
        double* b = rs.lvl[l];
        rc.acc[l] = ReproAcc{ b, b + nA, b + nA + nB, b + nA + nB + nD, b + nA + nB + nD + nd.Nstate };
    }
    rc.cmax = rs.cmax; rc.N = N;
    return rc;
}
// pass 2: the products, pre-rounded, into the three levels (atomics whose order does not matter)
__global__ __launch_bounds__(64)
void repro_step_rows_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nboard, int rows_from, int rows_to,
                            const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, int N)
{
    if(opref_skip(R)) return;
    if(!*plan.repro.any) return;
    const int r = repro_step_row(P, plan, blockIdx.x*blockDim.x + threadIdx.x, nboard, rows_from, rows_to);
    if(r < 0) return;
    const ReproCtx rc = repro_step_ctx(nd, plan.repro, N);
    rows_repro_row(nd, opref_get(R), r, Jp, Ji, rc, 0, true);
}
// pass 3: entry += (level 1 + level 2) + level 3, the levels back to zero; the D blocks' lower triangles mirrored (the
// levels hold them alone, like A's - whose upper triangle nobody reads for these models). Block 0 clears the maxima
__global__ __launch_bounds__(256)
void repro_step_combine_kernel(NormalDims nd, OpRef R, ReproStep rs)
{
    if(opref_skip(R)) return;
    if(!*rs.any) return;
    const OpDev& O = opref_get(R);
    const size_t nA = (size_t)nd.Nc*nd.Nc, nB = (size_t)nd.NE*nd.Nc, nD = (size_t)nd.NEb*36;
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < rs.one; i += (size_t)gridDim.x*blockDim.x)
    {
        const double l1 = rs.lvl[0][i], l2 = rs.lvl[1][i], l3 = rs.lvl[2][i];
        if(l1 == 0.0 && l2 == 0.0 && l3 == 0.0) continue;
        rs.lvl[0][i] = 0.0; rs.lvl[1][i] = 0.0; rs.lvl[2][i] = 0.0;
        const double v = (l1 + l2) + l3;
        size_t j = i;
        if(j < nA) { O.A[j] += v; continue; }                     j -= nA;
        if(j < nB) { O.Bt[j] += v; continue; }                    j -= nB;
        if(j < nD)
        {
            O.D[j] += v;
            const size_t blk = j/36, e = j - blk*36, a = e/6, b = e - a*6;
            if(a != b) O.D[blk*36 + b*6 + a] += v;
            continue;
        }                                                         j -= nD;
        if(j < (size_t)nd.Nstate) { O.g[j] += v; continue; }
        O.scalars[SC_NORM2_X] += v;
    }
    if(blockIdx.x == 0)
        for(int i = threadIdx.x; i <= nd.Nstate; i += blockDim.x) rs.cmax[i] = 0ull;
}

__global__ __launch_bounds__(64)
void rows_generic_kernel(NormalDims nd, OpRef R, int row0, int row1,
                         const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    if(opref_skip(R)) return;
    rows_generic_wave<false>(nd, opref_get(R), row0 + blockIdx.x*blockDim.x, row1, Jp, Ji);
}

// The same for problems made of such rows (structure from motion: tens of
// thousands of triangulated pairs against a camera block of a few variables;
// discrete points only). Every row then adds to the SAME few entries of A and g,
// and one global atomic per product serializes on them: 5.4 M atomics on 324
// addresses took 7.8 ms at BASELINE configuration 4. Here a workgroup of 256
// rows sums its camera-block products in LDS first (per wave, to keep the waves
// off each other's addresses) and flushes Nc^2 values; the frame/point parts,
// which are spread out, go straight to memory as before. Nc <= ROWS_LDS_NC
#define ROWS_LDS_NC 40
__global__ __launch_bounds__(256)
void rows_generic_lds_kernel(NormalDims nd, OpRef R, int row0, int row1,
                             const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    if(opref_skip(R)) return;
    extern __shared__ double lds_r[];               // per wave: A[Nc][Nc] | g[Nc]
    const OpDev& O = opref_get(R);
    const double* __restrict__ Jv = O.Jv;
    const double* __restrict__ x  = O.x;
    const int Nc = nd.Nc, per_wave = Nc*Nc + Nc;
    const int t = threadIdx.x, wave = t >> 6;
    for(int i = t; i < 4*per_wave; i += 256) lds_r[i] = 0.0;
    __syncthreads();
    double* __restrict__ Aw = lds_r + wave*per_wave;
    double* __restrict__ gw = Aw + Nc*Nc;

    const int r = row0 + blockIdx.x*256 + t;
    double n2 = 0.0;
    if(r < row1)
    {
        const int p0 = Jp[r], p1 = Jp[r+1];
        const double xr = x[r];
        n2 = xr*xr;
        for(int p = p0; p < p1; p++)
        {
            const int    ci = Ji[p];
            const double vi = Jv[p];
            if(vi == 0.0) continue;
            const int si = state_to_SE(nd, ci);
            if(si >= 0) atomicAdd(&gw[si], vi*xr); else atomicAdd(&O.g[ci], vi*xr);
            for(int q = p0; q < p1; q++)
            {
                const int    cj = Ji[q];
                const double v  = vi*Jv[q];
                if(v == 0.0) continue;
                const int    sj = state_to_SE(nd, cj);
                if(si >= 0 && sj >= 0)
                    atomicAdd(&Aw[si*Nc + sj], v);
                else if(si < 0 && sj >= 0)
                    atomicAdd(&O.Bt[(size_t)(-si-1)*Nc + sj], v);
                else if(si < 0 && sj < 0)
                {
                    int bi, ai, di, e0i, bj, aj, dj, e0j;
                    E_to_block(nd, -si-1, &bi, &ai, &di, &e0i);
                    E_to_block(nd, -sj-1, &bj, &aj, &dj, &e0j);
                    if(bi == bj) atomicAdd(&O.D[(size_t)bi*36 + ai*6 + aj], v);
                }
            }
        }
    }
    for(int off=32; off>0; off>>=1) n2 += __shfl_down(n2, off);
    if((t & 63) == 0 && n2 != 0.0) atomicAdd(&O.scalars[SC_NORM2_X], n2);
    __syncthreads();
    for(int i = t; i < per_wave; i += 256)
    {
        const double v = (lds_r[i] + lds_r[per_wave + i]) + (lds_r[2*per_wave + i] + lds_r[3*per_wave + i]);
        if(v == 0.0) continue;
        if(i < Nc*Nc) atomicAdd(&O.A[i], v);
        else
        {
            const int sc = i - Nc*Nc;
            atomicAdd(&O.g[S_to_state(nd, sc)], v);
        }
    }
}

// ---- rows outside the Grams, in a fixed order (GenPlan, solver_kernels.hpp) ----
// One workgroup per chunk of a group's rows. The chunk's camera-block values (and x) are staged in LDS,
// then every output - a pair (p <= q), a sum s_p x, or sum x^2 - is summed over the rows in order by one thread
__device__ __forceinline__
void gen_chunk(const GenPlan& G, const OpDev& O, const int32_t* __restrict__ Jp, int ichunk, double* __restrict__ lds)
{
    const int g  = G.chunk_group[ichunk];
    const int k  = G.group_k[g];
    const int* __restrict__ spos = G.spos + G.group_off[g];
    const int c0 = G.chunk_begin[ichunk], nrows = G.chunk_begin[ichunk+1] - c0;
    const int ld = k + 1;
    for(int i = threadIdx.x; i < nrows*ld; i += blockDim.x)
    {
        const int ir = i / ld, c = i - ir*ld;
        const int r  = G.rows[c0 + ir];
        lds[i] = (c < k) ? O.Jv[Jp[r] + spos[c]] : O.x[r];
    }
    __syncthreads();
    const int npairs = (k*(k+1)) >> 1, nout = npairs + k + 1;
    double* __restrict__ out = G.part + (size_t)ichunk*G.stride;
    for(int o = threadIdx.x; o < nout; o += blockDim.x)
    {
        int p, q;
        if(o < npairs)
        {
            // o -> (p,q), p <= q, row-major over the upper triangle
            p = 0; int rem = o;
            while(rem >= k - p) { rem -= k - p; p++; }
            q = p + rem;
        }
        else if(o < npairs + k) { p = o - npairs; q = k; }
        else                    { p = k; q = k; }
        double acc = 0.0;
        for(int ir = 0; ir < nrows; ir++) acc = fma(lds[ir*ld + p], lds[ir*ld + q], acc);
        out[o] = acc;
    }
}
// One wave per eliminated block that such rows touch: its rows of Bt, its D block and its part of g, the
// rows applied one after the other. lds: [6][Nc] Bt rows, then [36] D, then [6] g
__device__ __forceinline__
void gen_eblock(const GenPlan& G, const NormalDims& nd, const OpDev& O, const int32_t* __restrict__ Jp, int ieb,
                double* __restrict__ lds)
{
    const int blk = G.eb_block[ieb];
    int de, e0;
    if(blk < nd.Nfb) { de = 6; e0 = 6*blk; } else { de = 3; e0 = 6*nd.Nfb + 3*(blk - nd.Nfb); }
    const int Nc = nd.Nc, nlds = 6*Nc + 42;
    for(int i = threadIdx.x; i < nlds; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    double* __restrict__ lD = lds + 6*Nc;
    double* __restrict__ lg = lD + 36;
    for(int ii = G.eb_begin[ieb]; ii < G.eb_begin[ieb+1]; ii++)
    {
        const int r = G.eb_rows[ii], g = G.eb_group[ii], k = G.group_k[g];
        const int* __restrict__ spos = G.spos + G.group_off[g];
        const int* __restrict__ scol = G.scol + G.group_off[g];
        const double* __restrict__ jr = O.Jv + Jp[r];
        const double* __restrict__ je = jr + G.eb_epos[ii];
        const double xr = O.x[r];
        // (within one row every entry below is touched by one thread: plain read-modify-writes)
        for(int i = threadIdx.x; i < de*k; i += blockDim.x)
        {
            const int a = i / k, c = i - a*k;
            lds[a*Nc + scol[c]] = fma(je[a], jr[spos[c]], lds[a*Nc + scol[c]]);
        }
        if((int)threadIdx.x < de*de)
        {
            const int a = threadIdx.x / de, b = threadIdx.x - a*de;
            lD[a*6 + b] = fma(je[a], je[b], lD[a*6 + b]);
        }
        if((int)threadIdx.x < de) lg[threadIdx.x] = fma(je[threadIdx.x], xr, lg[threadIdx.x]);
        __syncthreads();
    }
    for(int i = threadIdx.x; i < de*Nc; i += blockDim.x) O.Bt[(size_t)e0*Nc + i] = lds[i];
    if((int)threadIdx.x < 36) O.D[(size_t)blk*36 + threadIdx.x] = lD[threadIdx.x];
    if((int)threadIdx.x < de) O.g[nd.E_state0 + e0 + threadIdx.x] = lg[threadIdx.x];
}
static size_t gen_lds_bytes(const GenPlan& G, const NormalDims& nd)
{
    const size_t a = (size_t)GEN_CHUNK*(G.kmax + 1), b = (G.Neblocks > 0) ? (size_t)6*nd.Nc + 42 : 0;
    return (a > b ? a : b)*sizeof(double);
}
// workgroups [0, Nchunks): chunks; then the eliminated blocks
__global__ __launch_bounds__(256)
void gen_rows_kernel(NormalDims nd, OpRef R, GenPlan G, const int32_t* __restrict__ Jp)
{
    if(opref_skip(R)) return;
    extern __shared__ double lds_g[];
    const OpDev& O = opref_get(R);
    if((int)blockIdx.x < G.Nchunks) gen_chunk(G, O, Jp, blockIdx.x, lds_g);
    else                            gen_eblock(G, nd, O, Jp, blockIdx.x - G.Nchunks, lds_g);
}
// the regularization rows where there are no board Grams to ride with: every row has destinations of its own
// (A and g through one add each into the cleared buffers); |x|^2 summed in a fixed order. One workgroup
__global__ __launch_bounds__(256)
void rows_single_kernel(NormalDims nd, OpRef R, int row0, int row1, const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    double n2 = 0.0;
    for(int r = row0 + threadIdx.x; r < row1; r += blockDim.x)
    {
        double v = 0.0;
        rows_generic_row(nd, O, r, row1, Jp, Ji, &v);
        n2 += v;
    }
    for(int off=32; off>0; off>>=1) n2 += __shfl_down(n2, off);
    __shared__ double part[4];
    if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n2;
    __syncthreads();
    if(threadIdx.x == 0) O.scalars[SC_NORM2_X] += (part[0] + part[1]) + (part[2] + part[3]);
}

// The Gram assembly (+ elimination of the frame blocks) and the generic rows in
// ONE launch (all three kinds of work are independent): workgroups
// [0, nframe_blocks) take a frame each, the next Nchunks*slices 256 positions of a pair chunk each, the
// rest 256 generic rows each.
//   mode (device flag, or mode_host): 0 nothing; 1 the point *sel_eval was just
//   evaluated; 2 re-eliminate the point *sel_cur from its stored blocks
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5)))
void assemble_factor_kernel(DeviceProblem P, NormalDims nd, BlockRanges br, const OpDev* __restrict__ ops,
                            const int* __restrict__ sel_eval, const int* __restrict__ sel_cur,
                            const SolverCtl* __restrict__ ctl, const int* __restrict__ skip,
                            const int* __restrict__ mode_ptr, int mode_host,
                            int do_factor, double lambda_host,
                            AssemblyPlan plan, const double* __restrict__ gram, FactorBuffers F,
                            int nframe_blocks, int row0, int row1,
                            const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    // everything the workgroup needs to know before it can ask for data, asked for at once: one trip to memory.
    // (The chunk workgroups last: their loads then fall into the frame workgroups' arithmetic. First: 17.4 us
    //  against 16.5)
    const int nslices = assemble_chunk_slices(P), nchunk_blocks = plan.Nchunks*nslices;
    const int b = blockIdx.x;
    const bool frame_block = b < nframe_blocks;
    const int skipv = skip     ? *skip     : 0;
    const int mode  = mode_ptr ? *mode_ptr : mode_host;
    const int ie    = sel_eval ? *sel_eval : 0;
    const int ic    = sel_cur  ? *sel_cur  : 0;
    const double lambda = ctl ? ctl->lambda : lambda_host;
    const int o0 = frame_block ? plan.frame_obs_begin[br.frame_lo + b]     : 0;
    const int o1 = frame_block ? plan.frame_obs_begin[br.frame_lo + b + 1] : 0;
    if(skipv || mode == 0) return;
    extern __shared__ double lds_f[];
    const OpDev& O = ops[(mode == 1) ? ie : ic];
    if(frame_block)
        assemble_frame_block(P, nd, O, plan, gram, br.frame_lo + b, o0, o1, mode, do_factor != 0, lambda, F, lds_f);
    else if(mode != 1) return;
    else if(b < nframe_blocks + nchunk_blocks)
    {
        const int cb = b - nframe_blocks;
        reduce_pair_chunk(P, plan, gram, cb / nslices, cb - (cb / nslices)*nslices);
    }
    else
    {
        // rows that do not come from board observations; |x|^2 of each workgroup's
        // rows goes to its slot of row_part, summed in order by assemble_finalize()
        const int rb = b - nframe_blocks - nchunk_blocks;
        double n2 = 0.0;
        rows_generic_row(nd, O, row0 + rb*256 + threadIdx.x, row1, Jp, Ji, &n2);
        for(int off=32; off>0; off>>=1) n2 += __shfl_down(n2, off);
        __shared__ double part[4];
        if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n2;
        __syncthreads();
        if(threadIdx.x == 0) plan.row_part[rb] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

__global__ __launch_bounds__(64)
void assemble_finalize_kernel(int npos, NormalDims nd, const OpDev* __restrict__ ops, const int* __restrict__ sel,
                              const int* __restrict__ skip, AssemblyPlan plan)
{
    if(skip != NULL && *skip) return;
    assemble_finalize(npos, nd, ops[sel ? *sel : 0], plan, blockIdx.x*blockDim.x + threadIdx.x);
}
// the same, riding along in the SYRK launch (row nslices of its grid): the two are independent
struct FinalizeRide
{
    int           row0;      // first row (blockIdx.y) of the grid that is the ride's
    int           npos;      // 0: nothing rides along
    const OpDev*  ops;
    const int*    sel;
    const int*    skip;
    AssemblyPlan  plan;
};
__device__ __forceinline__ void finalize_ride(const FinalizeRide& fr, const NormalDims& nd)
{
    if(fr.skip != NULL && *fr.skip) return;
    const OpDev& O = fr.ops[fr.sel ? *fr.sel : 0];
    assemble_finalize(fr.npos, nd, O, fr.plan, ((blockIdx.y - fr.row0)*gridDim.x + blockIdx.x)*blockDim.x + threadIdx.x);
}

// A, Bt, D, g and the scalars of an operating point, zeroed in one launch
__global__ __launch_bounds__(256)
void zero_normal_kernel(NormalDims nd, OpRef R)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const size_t nA = (size_t)nd.Nc*nd.Nc, nB = (size_t)nd.NE*nd.Nc, nD = (size_t)nd.NEb*36;
    const size_t total = nA + nB + nD + nd.Nstate + NSCALARS;
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x*blockDim.x)
    {
        size_t j = i;
        if(j < nA) { O.A[j] = 0.0; continue; }   j -= nA;
        if(j < nB) { O.Bt[j] = 0.0; continue; }  j -= nB;
        if(j < nD) { O.D[j] = 0.0; continue; }   j -= nD;
        if(j < (size_t)nd.Nstate) { O.g[j] = 0.0; continue; }  j -= nd.Nstate;
        O.scalars[j] = 0.0;
    }
}

////////////////////////////////////////////////////////////////////////////////
// Schur complement
////////////////////////////////////////////////////////////////////////////////

// One workgroup per E block: L L^T = D_e + lambda I;  Wt_e = L^-1 Bt_e;  y_e = L^-1 g_e.
// status[0] is set to 1 if any block is not positive definite
__global__ __launch_bounds__(256)
void eblock_factor_kernel(NormalDims nd, BlockRanges br, int first, OpRef R, double lambda_host, const SolverCtl* ctl,
                          double* __restrict__ Wt, double* __restrict__ LD, double* __restrict__ y,
                          int* __restrict__ status, unsigned* __restrict__ occ, int nocc, double* __restrict__ Wtile)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const double* __restrict__ Bt = O.Bt;
    const double* __restrict__ D  = O.D;
    const double* __restrict__ g  = O.g;
    const double lambda = ctl ? ctl->lambda : lambda_host;

    __shared__ double L[36];
    __shared__ double rinv[6];
    __shared__ unsigned occ_s[8];           // which 16-column tiles of this block's Wt rows are not all zero (Nc <= 4096)
    if(threadIdx.x < 8) occ_s[threadIdx.x] = 0u;
    const int blk = br.block(first + blockIdx.x);
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    const int t   = threadIdx.x;

    // everything this workgroup reads is requested up front: the block, and this
    // lane's columns of Bt_e (column Nc is g_e). The 6x6 factorization below
    // then runs under the latency of the big loads
    const double dval = D[(size_t)blk*36 + min(t, 35)];
    // (one wave for camera blocks up to 255 columns, four for wider ones: the splined models' 1206 columns
    //  by one wave per block were 800 waves on the whole chip, 67 us of latency)
    constexpr int MAXC = 4;                // columns per lane held in registers
    const int nth = blockDim.x;
    double bt[MAXC][6];
#pragma unroll
    for(int cc = 0; cc < MAXC; cc++)
    {
        const int c = t + nth*cc;
#pragma unroll
        for(int i=0;i<6;i++)
            bt[cc][i] = (i < de && c <= nd.Nc) ? ((c < nd.Nc) ? Bt[(size_t)(e0+i)*nd.Nc + c] : g[nd.E_state0 + e0 + i]) : 0.0;
    }

    if(t < 36) L[t] = dval + (((t/6) == (t%6)) ? lambda : 0.0);
    __syncthreads();
    if(t == 0)
    {
        // the 6x6 (3x3) factorization in REGISTERS: one batch of LDS reads, fully
        // unrolled arithmetic, one batch of writes. (Working in LDS puts an LDS
        // round trip, ~100 cycles, on every one of the ~90 dependent accesses.)
        // A 3x3 point block is padded with the identity
        double M[6][6];
#pragma unroll
        for(int i=0;i<6;i++)
#pragma unroll
            for(int j=0;j<6;j++)
            {
                const double v = L[i*6+j];
                M[i][j] = (i < de && j < de) ? v : ((i == j) ? 1.0 : 0.0);
            }
        bool ok = true;
        double ri[6];
#pragma unroll
        for(int j=0;j<6;j++)
        {
            double d = M[j][j];
#pragma unroll
            for(int k=0;k<j;k++) d -= M[j][k]*M[j][k];
            if(!(d > 0.0)) { ok = false; d = 1.0; }
            d = sqrt(d);
            const double rd = 1.0/d;
            M[j][j] = d;
            ri[j]   = rd;
#pragma unroll
            for(int i=j+1;i<6;i++)
            {
                double v = M[i][j];
#pragma unroll
                for(int k=0;k<j;k++) v -= M[i][k]*M[j][k];
                M[i][j] = v*rd;
            }
        }
#pragma unroll
        for(int i=0;i<6;i++)
        {
            rinv[i] = ri[i];
#pragma unroll
            for(int j=0;j<=i;j++) if(i < de) L[i*6+j] = M[i][j];
        }
        if(!ok) atomicExch(status, 1);
    }
    __syncthreads();
    if(t < 36) LD[(size_t)blk*36 + t] = L[t];

    // forward substitution, one column of Bt_e per lane and pass
    double Lr[6][6], ri[6];
#pragma unroll
    for(int i=0;i<6;i++)
    {
        ri[i] = rinv[i < de ? i : 0];
#pragma unroll
        for(int k=0;k<6;k++) Lr[i][k] = L[i*6+k];
    }
#pragma unroll
    for(int cc = 0; cc < MAXC; cc++)
    {
        const int c = t + nth*cc;
        if(c > nd.Nc) break;
        double w[6];
#pragma unroll
        for(int i=0;i<6;i++)
        {
            double v = bt[cc][i];
#pragma unroll
            for(int k=0;k<i;k++) v -= Lr[i][k]*w[k];
            w[i] = v*ri[i];
        }
        if(c < nd.Nc) { for(int i=0;i<de;i++) Wt[(size_t)(e0+i)*nd.Nc + c] = w[i]; }
        else          { for(int i=0;i<de;i++) y[e0+i] = w[i]; }
        if(occ != NULL)
        {
            // this wave's 64 columns = 4 tiles, starting at a multiple of 64
            bool nz = false;
            for(int i=0;i<de;i++) nz = nz || (w[i] != 0.0);
            const unsigned long long m = __ballot(nz && c < nd.Nc);
            const unsigned bits = ((m & 0xffffull) ? 1u : 0u) | (((m >> 16) & 0xffffull) ? 2u : 0u) |
                                  (((m >> 32) & 0xffffull) ? 4u : 0u) | ((m >> 48) ? 8u : 0u);
            const int tile0 = (c - (t & 63)) >> 4;
            if((t & 63) == 0 && bits) atomicOr(&occ_s[tile0 >> 5], bits << (tile0 & 31));
            // the tiled copy, of the tiles that hold something
            if(Wtile != NULL && c < nd.Nc && ((bits >> ((t & 63) >> 4)) & 1u))
                for(int i=0;i<de;i++) Wtile[((size_t)(c >> 4)*nd.NE + e0 + i)*16 + (c & 15)] = w[i];
        }
    }
    // wider camera blocks: the remaining columns, plainly
    for(int c = t + nth*MAXC; c <= nd.Nc; c += nth)
    {
        double w[6];
        for(int i=0;i<de;i++)
        {
            double v = (c < nd.Nc) ? Bt[(size_t)(e0+i)*nd.Nc + c] : g[nd.E_state0 + e0 + i];
            for(int k=0;k<i;k++) v -= L[i*6+k]*w[k];
            w[i] = v*rinv[i];
        }
        if(c < nd.Nc) for(int i=0;i<de;i++) Wt[(size_t)(e0+i)*nd.Nc + c] = w[i];
        else          for(int i=0;i<de;i++) y[e0+i] = w[i];
        if(occ != NULL)
        {
            bool nz = false;
            for(int i=0;i<de;i++) nz = nz || (w[i] != 0.0);
            if(nz && c < nd.Nc) atomicOr(&occ_s[(c >> 4) >> 5], 1u << ((c >> 4) & 31));
            // (the 16 columns of a tile are 16 neighbouring lanes here too)
            const unsigned long long m = __ballot(nz && c < nd.Nc);
            if(Wtile != NULL && c < nd.Nc && ((m >> (t & 48)) & 0xffffull))
                for(int i=0;i<de;i++) Wtile[((size_t)(c >> 4)*nd.NE + e0 + i)*16 + (c & 15)] = w[i];
        }
    }
    if(occ != NULL)
    {
        __syncthreads();
        if(t < nocc) occ[(size_t)blk*nocc + t] = occ_s[t];
    }
}

// S = A + lambda I - sum_slots partial(Wt^T Wt) ;  r = g_S - sum_slots partial(Wt^T y).
// The SYRK below leaves, per slot (= slice of E rows), the 16x16 tiles of its
// part of Wt^T Wt in the MFMA's register order, Spart[slot][pair][v][lane] with
// element (i = 16 bi + lane/16 + 4 v, j = 16 bj + lane%16), and behind all of
// those its part of Wt^T y, rpart[slot][16 nb]. One thread per tile element
// sums over the slots (coalesced) and writes the LOWER triangle S[j][i], j >= i,
// which is what the Cholesky reads. (A lives in the full square; only its lower
// triangle is copied.) No atomics anywhere: the result does not depend on the
// order in which workgroups finish
#define SRED_SPLIT 4      // threads sharing one output element (adjacent lanes)
__device__ __forceinline__
void schur_reduce_body(const NormalDims& nd, const OpDev& O, double lambda, int add_g /* r starts from g_S (else from 0) */,
                       int nslots, const double* __restrict__ Spart,
                       double* __restrict__ S, double* __restrict__ r, int block,
                       const unsigned char* __restrict__ live = NULL /* [nslots][npairs]: the slot holds the tile (sparse SYRK); NULL: all do */,
                       double* __restrict__ iso = NULL /* given: S goes out COMPACTED by O.cperm (LcholCompact): the coupled
                                                         variables' n' x n' matrix with its rhs as row n', the isolated pairs' blocks here */,
                       int* __restrict__ err = NULL /* with iso: set to 3 if an entry that the compaction has no place for is not zero */,
                       double* __restrict__ ndMA = NULL, double* __restrict__ ndMB = NULL /* with iso and an active plan in O.ndp: the two sides' matrices (lchol_nd_*) */)
{
    // (the dissection: the coupled variables go to three matrices by their classes - the separator's to S)
    const bool ndact = (iso != NULL && ndMA != NULL && O.ndp != NULL && O.ndp[NDH_ACTIVE] != 0);
    const int  ndA = ndact ? O.ndp[NDH_NA] : 0, ndB = ndact ? O.ndp[NDH_NB] : 0, ndS = ndact ? O.ndp[NDH_NS] : 0;
    const int* __restrict__ npos = ndact ? O.ndp + NDH_WORDS : (const int*)NULL;
    const int nb = (nd.Nc + 15) >> 4, npairs = nb*(nb+1)/2;
    // SRED_SPLIT threads per element, each taking every SRED_SPLIT-th slot, 4
    // loads in flight; the 64-byte groups they read are still whole cache lines
    // across the wave (16 consecutive elements x SRED_SPLIT slots)
    // (the sparse SYRK's few slots with anything in them: a thread per element, the slots in order)
    const int gid = block*blockDim.x + threadIdx.x;
    const int split = (live != NULL) ? 1 : SRED_SPLIT;
    const int sub = (live != NULL) ? 0 : (gid >> 4) & (SRED_SPLIT-1);
    const int idx = (live != NULL) ? gid : ((gid >> 6) << 4) | (gid & 15);      // 16 elements per wave
    const int nS  = npairs*256, nTot = nS + nb*16;
    if(idx >= nTot) return;
    const bool is_r = idx >= nS;
    const double* __restrict__ base = is_r ? Spart + (size_t)nslots*nS + (idx - nS) : Spart + idx;
    const size_t stride = is_r ? (size_t)nb*16 : (size_t)nS;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int s = sub;
    if(live != NULL && !is_r)
    {
        // the sparse SYRK leaves most slots of most tiles unwritten (no block of the slice touches the tile): what is
        // not there is not read - the same sum, a zero added is a zero skipped
        const unsigned char* __restrict__ lv = live + (idx >> 8);
        for(; s + 7 < nslots; s += 8)
        {
            unsigned char f[8]; double v[8];
#pragma unroll
            for(int u = 0; u < 8; u++) f[u] = lv[(size_t)(s + u)*npairs];
#pragma unroll
            for(int u = 0; u < 8; u++) { v[u] = 0.0; if(f[u]) v[u] = base[(size_t)(s + u)*stride]; }
#pragma unroll
            for(int u = 0; u < 8; u++) a0 += v[u];
        }
        for(; s < nslots; s++) if(lv[(size_t)s*npairs]) a0 += base[(size_t)s*stride];
    }
    else
    {
    for(; s + 3*split < nslots; s += 4*split)
    {
        a0 += base[(size_t)(s          )*stride];
        a1 += base[(size_t)(s +   split)*stride];
        a2 += base[(size_t)(s + 2*split)*stride];
        a3 += base[(size_t)(s + 3*split)*stride];
    }
    for(; s < nslots; s += split) a0 += base[(size_t)s*stride];
    }
    double acc = (a0 + a1) + (a2 + a3);
    if(split > 1)
    {
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
    }
    if(sub != 0) return;
    if(!is_r)
    {
        // tile pair idx >> 8 -> (bi, bj), bi <= bj, row bi of the pairs starting at bi nb - bi (bi - 1)/2: the root of
        // the quadratic, put right by a step either way (walking the rows from 0 was up to 76 steps for every one of
        // the 3 M threads of a 1206-column camera block: most of this kernel's 25 us)
        const int pp = idx >> 8;
        int bi = (int)(0.5*((double)(2*nb + 1) - sqrt((double)(2*nb + 1)*(double)(2*nb + 1) - 8.0*(double)pp)));
        bi = max(0, min(nb - 1, bi));
        while(bi > 0 && bi*nb - ((bi*(bi - 1)) >> 1) > pp) bi--;
        while(bi + 1 < nb && (bi + 1)*nb - (((bi + 1)*bi) >> 1) <= pp) bi++;
        const int bj = bi + (pp - (bi*nb - ((bi*(bi - 1)) >> 1)));
        const int v = (idx >> 6) & 3, lane = idx & 63;
        const int i = 16*bi + (lane >> 4) + 4*v, j = 16*bj + (lane & 15);
        if(i < nd.Nc && j < nd.Nc && j >= i)
        {
            const double v = O.A[(size_t)j*nd.Nc + i] + ((i==j) ? lambda : 0.0) - acc;
            if(iso == NULL) S[(size_t)j*nd.Nc + i] = v;
            else
            {
                const int* __restrict__ ip = O.cperm + nd.Nc;
                const int n1 = O.cperm[2*nd.Nc], pi = ip[i], pj = ip[j];
                if(pi < n1 && pj < n1)
                {
                    if(!ndact) S[(size_t)max(pi, pj)*n1 + min(pi, pj)] = v;
                    else
                    {
                        const int ci = npos[i], cj = npos[j], ki = ci >> 28, kj = cj >> 28, xi = ci & 0xfffffff, xj = cj & 0xfffffff;
                        if(ki == kj)
                        {
                            double* __restrict__ Mk = (ki == 0) ? S : ((ki == 1) ? ndMA : ndMB);
                            const int ldk = (ki == 0) ? ndS : ((ki == 1) ? ndA + ndS : ndB + ndS);
                            Mk[(size_t)max(xi, xj)*ldk + min(xi, xj)] = v;
                        }
                        else if(ki == 0 || kj == 0)
                        {
                            // a side's variable against the separator's: the side's border rows
                            const int kx = ki ? ki : kj, xs = ki ? xj : xi, xx = ki ? xi : xj;
                            double* __restrict__ Mk = (kx == 1) ? ndMA : ndMB;
                            const int nx = (kx == 1) ? ndA : ndB;
                            Mk[(size_t)(nx + xs)*(nx + ndS) + xx] = v;
                        }
                        else if(v != 0.0 && err != NULL) *err = 3;          // the two sides are coupled after all
                    }
                }
                else if(pi >= n1 && pj >= n1 && ((pi - n1) >> 1) == ((pj - n1) >> 1))
                    iso[4*((pi - n1) >> 1) + ((pi - n1) & 1) + ((pj - n1) & 1)] = v;      // (0,0) -> 0, (1,0) -> 1, (1,1) -> 2
                else if(v != 0.0 && err != NULL) *err = 3;                  // an isolated variable that is coupled after all: the solve fails, loudly
            }
        }
    }
    else
    {
        const int i = idx - nS;
        if(i < nd.Nc)
        {
            const double v = (add_g ? O.g[S_to_state(nd, i)] : 0.0) - acc;
            if(iso == NULL) r[i] = v;
            else
            {
                const int n1 = O.cperm[2*nd.Nc], pi = O.cperm[nd.Nc + i];
                if(pi < n1)
                {
                    if(!ndact) S[(size_t)n1*n1 + pi] = v;
                    else
                    {
                        const int ci = npos[i], ki = ci >> 28, xi = ci & 0xfffffff;
                        if(ki == 0)      S[(size_t)ndS*ndS + xi] = v;
                        else if(ki == 1) ndMA[(size_t)(ndA + ndS)*(ndA + ndS) + xi] = v;
                        else             ndMB[(size_t)(ndB + ndS)*(ndB + ndS) + xi] = v;
                    }
                }
                else        iso[4*(nd.Nc/2 + 1) + (pi - n1)] = v;
            }
        }
    }
}
__global__ __launch_bounds__(256)
void schur_reduce_kernel(NormalDims nd, OpRef R, double lambda_host, const SolverCtl* ctl, int is_leader,
                         int nslots, const double* __restrict__ Spart,
                         double* __restrict__ S, double* __restrict__ r, const unsigned char* __restrict__ live)
{
    if(opref_skip(R)) return;
    const double lambda = is_leader ? (ctl ? ctl->lambda : lambda_host) : 0.0;
    schur_reduce_body(nd, opref_get(R), lambda, is_leader, nslots, Spart, S, r, blockIdx.x, live);
}

// Wt^T Wt and Wt^T y on the FP64 matrix cores: one wave per (16x16 tile of S,
// slice of E rows), v_mfma_f64_16x16x4: with lane l holding
// Wt[k0 + l/16][c0 + l%16], one 8-byte load per lane is a whole A operand
// (A[i][k] = Wt[k][i0+i]) and, for another column block, a whole B operand:
// 2 loads feed 1024 multiply-adds, no LDS, no barriers. Each wave stores its
// accumulators as they are (coalesced) into its slot of Spart;
// schur_reduce_kernel sums the slots. The products of the diagonal tiles with
// y give r. Result layout of the instruction (measured): register v of lane l
// holds D[l/16 + 4 v][l%16]
typedef double syrk_d4 __attribute__((ext_vector_type(4)));
#ifndef SYRK_UNROLL
#define SYRK_UNROLL 16
#endif
// Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), and every XCD has its own
// L2. In the natural order every XCD ends up reading ALL of Wt (each tile pair needs its two column
// strips over all rows): 8 x |Wt| from the Infinity Cache into the L2s. Here the slices (row ranges
// of Wt) are dealt to the XCDs instead: XCD k takes slices k, k+8, ... of every tile pair and reads
// an eighth of the rows. (nslices is a multiple of 8: syrk_slicing)
__device__ __forceinline__ void syrk_xcd_map(int nslices, int* px, int* sy)
{
    *px = blockIdx.x; *sy = blockIdx.y;
    if((nslices & 7) == 0)
    {
        const int L = blockIdx.x + gridDim.x*blockIdx.y;
        const int j = L >> 3;
        *sy = (L & 7) + 8*(j / (int)gridDim.x);
        *px = j % (int)gridDim.x;
    }
}
__global__ __launch_bounds__(64)
void schur_syrk_mfma_kernel(NormalDims nd, const int* __restrict__ skip, int e_lo, int e_hi, int e_per_slice,
                            int slot0, int nslots_total,
                            const double* __restrict__ Wt, const double* __restrict__ y,
                            double* __restrict__ Spart, int nslices, FinalizeRide fr)
{
    if((int)blockIdx.y >= nslices) { finalize_ride(fr, nd); return; }
    if(skip != NULL && *skip) return;
    // tile pair p -> (bi <= bj)
    const int nb = (nd.Nc + 15) >> 4, npairs = nb*(nb+1)/2;
    int px, sy;
    syrk_xcd_map(nslices, &px, &sy);
    int bi = 0, p = px;
    while(p >= nb - bi) { p -= nb - bi; bi++; }
    const int bj = bi + p;
    const int e_begin = e_lo + sy*e_per_slice;
    const int e_end   = min(e_hi, e_begin + e_per_slice);   // may be empty: the slot is then written as zeros

    const int lane = threadIdx.x;
    const int kk = lane >> 4, cc = lane & 15;
    const int ci = 16*bi + cc, cj = 16*bj + cc;
    const bool oki = ci < nd.Nc, okj = cj < nd.Nc;
    const double* __restrict__ pi = Wt + (oki ? ci : 0);
    const double* __restrict__ pj = Wt + (okj ? cj : 0);
    const bool diag = (bi == bj);

    syrk_d4 acc  = {0.0, 0.0, 0.0, 0.0};
    syrk_d4 accr = {0.0, 0.0, 0.0, 0.0};
#ifdef ASM_TS
    const long long sts0 = clock64();
#endif
    for(int e0 = e_begin; e0 < e_end; e0 += 4*SYRK_UNROLL)
    {
        double a[SYRK_UNROLL], b[SYRK_UNROLL], yy[SYRK_UNROLL];
#pragma unroll
        for(int u=0;u<SYRK_UNROLL;u++)
        {
            const int  e  = e0 + 4*u + kk;
            const bool ok = e < e_end;
            const size_t row = (size_t)(ok ? e : e_begin)*nd.Nc;
            a[u] = pi[row];
            b[u] = diag ? 0.0 : pj[row];
            yy[u] = (diag && cc == 0) ? y[ok ? e : e_begin] : 0.0;
            if(!ok || !oki) a[u] = 0.0;
            if(!ok || !okj) b[u] = 0.0;
            if(!ok) yy[u] = 0.0;
        }
#pragma unroll
        for(int u=0;u<SYRK_UNROLL;u++)
        {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], diag ? a[u] : b[u], acc, 0, 0, 0);
            // (Wt^T y as a matrix instruction for ONE useful column doubles the diagonal tiles' matrix work; on
            //  the vector unit instead: measured, no change - the launch is as long as its loads' round trips)
            if(diag) accr = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], yy[u], accr, 0, 0, 0);
        }
    }
    const int slot = slot0 + sy;
    double* __restrict__ o = Spart + ((size_t)slot*npairs + px)*256;
#pragma unroll
    for(int v=0;v<4;v++) o[64*v + lane] = acc[v];
    if(diag && cc == 0)
    {
        // column 0 of the y product: D[i][0] = sum_k Wt[k][i0+i] y[k]
        double* __restrict__ rpart = Spart + (size_t)nslots_total*npairs*256 + (size_t)slot*nb*16 + 16*bi;
#pragma unroll
        for(int v=0;v<4;v++) rpart[kk + 4*v] = accr[v];
    }
#ifdef ASM_TS
    if(lane == 0 && (px == 1 || px == 44) && (sy == 0 || sy == 40)) printf("syrk ts px=%d sy=%d block=(%d,%d) rows=%d dur=%lld\n", px, sy, blockIdx.x, blockIdx.y, e_end-e_begin, clock64()-sts0);
#endif
}

// The same for big camera blocks (splined models: 76 x 76 tiles), where the
// kernel above is bound by the L2: every wave loads two operands per MFMA and
// an element of Wt is loaded Nc/16 times. Here a wave takes a STRIP of up to
// four tiles (bi, bj0 .. bj0+3): one A operand serves four B operands, 5 loads
// per 4 MFMAs instead of 8. Same slots, same reduction
#define SYRK_STRIP 4
#ifndef SYRK_DEPTH
#define SYRK_DEPTH 2      // blocks whose operands are in flight together (schur_syrk_sparse_kernel)
#endif
#define SYRK_STRIP_UNROLL 8
__global__ __launch_bounds__(64)
void schur_syrk_strip_kernel(NormalDims nd, const int* __restrict__ skip, int e_lo, int e_hi, int e_per_slice,
                             int slot0, int nslots_total,
                             const double* __restrict__ Wt, const double* __restrict__ y,
                             double* __restrict__ Spart, int nslices, FinalizeRide fr)
{
    if((int)blockIdx.y >= nslices) { finalize_ride(fr, nd); return; }
    if(skip != NULL && *skip) return;
    const int nb = (nd.Nc + 15) >> 4, npairs = nb*(nb+1)/2;
    // strip -> (bi, first bj)
    int px, sy;
    syrk_xcd_map(nslices, &px, &sy);
    int bi = 0, sidx = px;
    for(;;) { const int ng = (nb - bi + SYRK_STRIP - 1)/SYRK_STRIP; if(sidx < ng) break; sidx -= ng; bi++; }
    const int bj0 = bi + SYRK_STRIP*sidx;
    const int ntile = min(SYRK_STRIP, nb - bj0);
    const int pair0 = bi*nb - (bi*(bi-1))/2 + (bj0 - bi);          // pair index of (bi, bj0); the strip's follow
    const int e_begin = e_lo + sy*e_per_slice;
    const int e_end   = min(e_hi, e_begin + e_per_slice);

    const int lane = threadIdx.x;
    const int kk = lane >> 4, cc = lane & 15;
    const int ci = 16*bi + cc;
    const bool oki = ci < nd.Nc;
    const double* __restrict__ pi = Wt + (oki ? ci : 0);
    const double* __restrict__ pj[SYRK_STRIP];
    bool okj[SYRK_STRIP];
#pragma unroll
    for(int q = 0; q < SYRK_STRIP; q++)
    {
        const int cj = 16*(bj0 + q) + cc;
        okj[q] = (q < ntile) && cj < nd.Nc;
        pj[q]  = Wt + (okj[q] ? cj : 0);
    }
    const bool diag = (bj0 == bi);                                  // tile 0 of the strip is a diagonal tile

    syrk_d4 acc[SYRK_STRIP];
#pragma unroll
    for(int q = 0; q < SYRK_STRIP; q++) acc[q] = syrk_d4{0.0, 0.0, 0.0, 0.0};
    syrk_d4 accr = {0.0, 0.0, 0.0, 0.0};
    for(int e0 = e_begin; e0 < e_end; e0 += 4*SYRK_STRIP_UNROLL)
    {
        double a[SYRK_STRIP_UNROLL], b[SYRK_STRIP][SYRK_STRIP_UNROLL], yy[SYRK_STRIP_UNROLL];
#pragma unroll
        for(int u=0;u<SYRK_STRIP_UNROLL;u++)
        {
            const int  e  = e0 + 4*u + kk;
            const bool ok = e < e_end;
            const size_t row = (size_t)(ok ? e : e_begin)*nd.Nc;
            a[u] = pi[row];
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++) b[q][u] = pj[q][row];
            yy[u] = (diag && cc == 0) ? y[ok ? e : e_begin] : 0.0;
            if(!ok || !oki) a[u] = 0.0;
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++) if(!ok || !okj[q]) b[q][u] = 0.0;
            if(!ok) yy[u] = 0.0;
        }
#pragma unroll
        for(int u=0;u<SYRK_STRIP_UNROLL;u++)
        {
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[q][u], acc[q], 0, 0, 0);
            if(diag) accr = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], yy[u], accr, 0, 0, 0);
        }
    }
    const int slot = slot0 + sy;
#pragma unroll
    for(int q = 0; q < SYRK_STRIP; q++)
    {
        if(q >= ntile) break;
        double* __restrict__ o = Spart + ((size_t)slot*npairs + pair0 + q)*256;
#pragma unroll
        for(int v=0;v<4;v++) o[64*v + lane] = acc[q][v];
    }
    if(diag && cc == 0)
    {
        double* __restrict__ rpart = Spart + (size_t)nslots_total*npairs*256 + (size_t)slot*nb*16 + 16*bi;
#pragma unroll
        for(int v=0;v<4;v++) rpart[kk + 4*v] = accr[v];
    }
}

// The same where Wt is SPARSE by tiles (the splined models: a frame's rows of Wt are nonzero only under the
// knots its board covers, about a third of the 76 column tiles, so about a ninth of the tile pairs).
// eblock_factor_kernel leaves a bit per (block, 16-column tile); here a wave walks the BLOCKS of its slice:
// a block that does not touch tile bi is skipped (scalar test), of the others the A operand is loaded and
// only the B tiles the block touches are. A block's 6 (3) rows are two (one) k-steps of 4, the missing
// rows zero: a third more matrix instructions per processed block, for an eighth of the blocks x tiles.
// A slice is every nslices-th block of the range, whole (the operands come from the tiled copy of Wt, Wtile).
// The blocks that count are few but unevenly spread: nine strips in ten have none or two in a slice, the strips over
// the middle of the imager sixty, and ~1000 matrix instructions one after the other on ONE wave were the kernel's
// 62 us (with either the loads or the matrix instructions compiled out: ~50; with neither: 8). So a workgroup is
// SYRK_SPARSE_WAVES waves on the one strip, wave w taking every SYRK_SPARSE_WAVES-th block of the slice; their sums
// are added in wave order (LDS) - the same bits every time. (Tried instead: four STRIPS a workgroup, 104 us against
// 89; 16 and 32 slices, 53 and 60 us alone against 67 - and the reduction pays for the slots)
#define SYRK_SPARSE_WAVES 4
__global__ __launch_bounds__(64*SYRK_SPARSE_WAVES)
void schur_syrk_sparse_kernel(NormalDims nd, const int* __restrict__ skip, int e_lo, int e_hi,
                              int slot0, int nslots_total,
                              const double* __restrict__ Wt, const double* __restrict__ y,
                              double* __restrict__ Spart, int nslices, FinalizeRide fr,
                              const unsigned* __restrict__ occ, int nocc, unsigned char* __restrict__ live_out)
{
    if((int)blockIdx.y >= nslices) { finalize_ride(fr, nd); return; }
    if(skip != NULL && *skip) return;
    const int nb = (nd.Nc + 15) >> 4, npairs = nb*(nb+1)/2;
    int px, sy;
    syrk_xcd_map(nslices, &px, &sy);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int bi = 0, sidx = px;
    for(;;) { const int ng = (nb - bi + SYRK_STRIP - 1)/SYRK_STRIP; if(sidx < ng) break; sidx -= ng; bi++; }
    const int bj0 = bi + SYRK_STRIP*sidx;
    const int ntile = min(SYRK_STRIP, nb - bj0);
    const int pair0 = bi*nb - (bi*(bi-1))/2 + (bj0 - bi);
    // A slice is every nslices-th BLOCK of the range (not a run of rows): eblock_factor_kernel's workgroup i runs on
    // XCD i % 8 and leaves block i's rows of Wt in THAT XCD's L2, and syrk_xcd_map() puts slice s on XCD s % 8 - with
    // 8 slices a workgroup here finds what it reads in its own L2 (a trip of ~700 cycles instead of ~4000 to another
    // XCD's data; the walk is nothing but such trips)
    const int e_begin = e_lo, e_end = e_hi;

    const int lane = threadIdx.x & 63;
    const int kk = lane >> 4, cc = lane & 15;
    const int ci = 16*bi + cc;
    const bool oki = ci < nd.Nc;
    // (Wt here is eblock_factor_kernel's tiled copy, [tile][row][16]: a block's rows of a tile are contiguous)
    const double* __restrict__ pi = Wt + (size_t)bi*nd.NE*16 + cc;
    const double* __restrict__ pj[SYRK_STRIP];
    bool okj[SYRK_STRIP];
#pragma unroll
    for(int q = 0; q < SYRK_STRIP; q++)
    {
        const int cj = 16*(bj0 + q) + cc;
        okj[q] = (q < ntile) && cj < nd.Nc;
        pj[q]  = Wt + (size_t)min(bj0 + q, nb - 1)*nd.NE*16 + cc;
    }
    const bool diag = (bj0 == bi);

    syrk_d4 acc[SYRK_STRIP];
#pragma unroll
    for(int q = 0; q < SYRK_STRIP; q++) acc[q] = syrk_d4{0.0, 0.0, 0.0, 0.0};
    syrk_d4 accr = {0.0, 0.0, 0.0, 0.0};
    unsigned touched = 0;       // bit q: a block of this slice touched B tile q of the strip (and the A tile)
#ifdef SYRK_TS
    long long sts_scan = 0, sts_blocks = 0, sts_t0 = clock64(), sts_t1; int sts_live = 0, sts_mma = 0;
    const long long sts_begin = sts_t0, sts_wall0 = wall_clock64();
#define SYRK_TICK(w) { sts_t1 = clock64(); w += sts_t1 - sts_t0; sts_t0 = sts_t1; }
#else
#define SYRK_TICK(w)
#endif
    if(e_begin < e_end)
    {
        auto block_of   = [&](int e) { return (e < 6*nd.Nfb) ? e/6 : nd.Nfb + (e - 6*nd.Nfb)/3; };
        const int b_lo = block_of(e_begin), b_hi = block_of(e_end - 1) + 1;
        const int b_first = b_lo + sy;
        const int nblk = (b_first < b_hi) ? (b_hi - b_first + nslices - 1)/nslices : 0;
        const unsigned wi = bi >> 5, mi = 1u << (bi & 31);
        // The occupancy bits of 64 blocks at a time, one block per lane: ONE round trip for the lot, then the
        // wave goes through the blocks that touch its tiles. (Read block by block - a dependent load in front
        // of every block, skipped or not - the walk over a slice's 100 blocks was most of the kernel's 104 us.)
        // And the operands of the NEXT block that counts are asked for before the products of the current one
        // (nothing is done with what a load returns before the block is applied: a select on the spot is a wait on the
        //  spot, and the ten loads of a block were ten trips to memory one after the other)
        // A block is six rows (a frame) or three (a point), whole (the slices are made of blocks): two k-steps of 4, or
        // one. What a lane's row and column are worth is decided once, per kind of block, not per block: the walk is a
        // chain of trips to memory with ~30 instructions between them (it was ~200, and those were half of its time)
        struct Operands { double a0, a1, b0[SYRK_STRIP], b1[SYRK_STRIP], y0, y1; unsigned mb; bool six; };
        const int  ka6 = kk, kb6 = 4 + (kk & 1), ka3 = (kk < 3) ? kk : 0;          // the lane's rows of the block (valid rows always)
        const bool ma6 = oki, mb6 = oki && kk < 2, ma3 = oki && kk < 3;             // ... and whether they count
        auto fetch = [&](int b, unsigned mb, Operands& o)
        {
            o.six = b < nd.Nfb; o.mb = mb;
            const int e0 = o.six ? 6*b : 6*nd.Nfb + 3*(b - nd.Nfb);
            const int ea = e0 + (o.six ? ka6 : ka3), eb = e0 + (o.six ? kb6 : 0);
            const size_t rowa = (size_t)ea*16, rowb = (size_t)eb*16;
            o.a0 = pi[rowa]; o.a1 = pi[rowb];
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++)
                if((mb >> q) & 1u) { o.b0[q] = pj[q][rowa]; o.b1[q] = pj[q][rowb]; }
            if(diag) { o.y0 = y[ea]; o.y1 = y[eb]; }
        };
        auto apply = [&](const Operands& o)
        {
            const double a0 = (o.six ? ma6 : ma3) ? o.a0 : 0.0, a1 = mb6 ? o.a1 : 0.0;
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++)
                if((o.mb >> q) & 1u)
                {
                    // (the columns past the matrix, in its last tile, were never written: not even a zero times them)
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, okj[q] ? o.b0[q] : 0.0, acc[q], 0, 0, 0);
                    if(o.six) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, okj[q] ? o.b1[q] : 0.0, acc[q], 0, 0, 0);
                }
            if(diag)
            {
                accr = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, (cc == 0) ? o.y0 : 0.0, accr, 0, 0, 0);
                if(o.six) accr = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, (cc == 0) ? o.y1 : 0.0, accr, 0, 0, 0);
            }
        };
        // (round 4: SYRK_DEPTH blocks' operands in flight at once - the walk was one block ahead, and a block's ten
        //  loads are ten cache lines 9.6 KB apart: a trip to memory per block that counts, ~15 of them a workgroup)
        Operands ring[SYRK_DEPTH];
        int nring = 0;
        // (wave w: the blocks w, w + SYRK_SPARSE_WAVES, ... of the slice, 64 of them at a time)
        for(int base = 0; SYRK_SPARSE_WAVES*base + wave < nblk; base += 64)
        {
            unsigned m = 0;         // bits 0..3: the block touches B tile q of the strip; bit 4: it touches the A tile
            const int jblk = SYRK_SPARSE_WAVES*(base + lane) + wave;
            if(jblk < nblk)
            {
                const unsigned* __restrict__ ob = occ + (size_t)(b_first + nslices*jblk)*nocc;
                const unsigned wa = ob[wi];
                const int w0i = bj0 >> 5, w1i = min((bj0 + SYRK_STRIP - 1) >> 5, nocc - 1);
                const unsigned w0 = ob[w0i], w1 = ob[w1i];
#pragma unroll
                for(int q = 0; q < SYRK_STRIP; q++)
                {
                    const int tile = bj0 + q;
                    const unsigned w = ((tile >> 5) == w0i) ? w0 : w1;
                    if(q < ntile && ((w >> (tile & 31)) & 1u)) m |= 1u << q;
                }
                if(wa & mi) m |= 16u;
            }
            unsigned long long live = __ballot((m & 16u) && (diag || (m & 15u)));
            SYRK_TICK(sts_scan)
#ifdef SYRK_TS
            sts_live += __popcll(live);
#endif
            while(live)
            {
                // fill the ring, then spend it
#pragma unroll
                for(int d = 0; d < SYRK_DEPTH; d++)
                    if(d >= nring && live)
                    {
                        const int bit = __ffsll((long long)live) - 1;
                        live &= live - 1;
                        const unsigned mb = (unsigned)__builtin_amdgcn_readlane((int)m, bit);
                        fetch(b_first + nslices*(SYRK_SPARSE_WAVES*(base + bit) + wave), mb, ring[d]);
                        nring = d + 1;
                        touched |= mb;
#ifdef SYRK_TS
                        sts_mma += __popc(mb & 15u);
#endif
                    }
                if(nring == SYRK_DEPTH)
                {
#pragma unroll
                    for(int d = 0; d < SYRK_DEPTH; d++) apply(ring[d]);
                    nring = 0;
                }
            }
            SYRK_TICK(sts_blocks)
        }
#pragma unroll
        for(int d = 0; d < SYRK_DEPTH; d++) if(d < nring) apply(ring[d]);
        SYRK_TICK(sts_blocks)
    }
    // the waves' sums, in wave order
    if(SYRK_SPARSE_WAVES > 1)
    {
        __shared__ double red[SYRK_SPARSE_WAVES - 1][(SYRK_STRIP + 1)*256];
        __shared__ unsigned s_touched[SYRK_SPARSE_WAVES];
        if(lane == 0) s_touched[wave] = touched;
        if(wave > 0 && touched)
        {
            double* __restrict__ o = red[wave - 1];
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++)
#pragma unroll
                for(int v=0;v<4;v++) o[q*256 + 64*v + lane] = acc[q][v];
#pragma unroll
            for(int v=0;v<4;v++) o[SYRK_STRIP*256 + 64*v + lane] = accr[v];
        }
        __syncthreads();
        if(wave > 0) return;
        for(int w = 1; w < SYRK_SPARSE_WAVES; w++)
        {
            const unsigned tw = s_touched[w];
            if(!tw) continue;       // (nothing but zeros)
            touched |= tw;
            const double* __restrict__ o = red[w - 1];
#pragma unroll
            for(int q = 0; q < SYRK_STRIP; q++)
#pragma unroll
                for(int v=0;v<4;v++) acc[q][v] += o[q*256 + 64*v + lane];
#pragma unroll
            for(int v=0;v<4;v++) accr[v] += o[SYRK_STRIP*256 + 64*v + lane];
        }
    }
    // A tile no block of the slice touched is not written: its flag says so and the reduction does not read it
    // (7 of 8 slots at BASELINE configuration 2: 49 MB of zeros written and read back, before)
    const int slot = slot0 + sy;
#pragma unroll
    for(int q = 0; q < SYRK_STRIP; q++)
    {
        if(q >= ntile) break;
        const bool any = (touched >> q) & 1u;
        if(lane == 0) live_out[(size_t)slot*npairs + pair0 + q] = any ? 1 : 0;
        if(!any) continue;
        double* __restrict__ o = Spart + ((size_t)slot*npairs + pair0 + q)*256;
#pragma unroll
        for(int v=0;v<4;v++) o[64*v + lane] = acc[q][v];
    }
    if(diag && cc == 0)
    {
        double* __restrict__ rpart = Spart + (size_t)nslots_total*npairs*256 + (size_t)slot*nb*16 + 16*bi;
#pragma unroll
        for(int v=0;v<4;v++) rpart[kk + 4*v] = accr[v];
    }
#ifdef SYRK_TS
    if(lane == 0 && sy == 3 && (px % 97 == 0))
        printf("syrk strip %d (tile row %d, from tile %d) slice %d: wall %lld, whole %lld cycles: scan %lld, blocks %lld (%d live, %d B tiles)\n",
               px, bi, bj0, sy, sts_wall0, clock64() - sts_begin, sts_scan, sts_blocks, sts_live, sts_mma);
#endif
}

// Dense Cholesky of S (lower triangle valid on input) and the solve S d = -r,
// one workgroup of 1024 (16 waves). On output r holds d and, if keep_factor,
// the lower triangle of S holds L.
//
// Blocked, panels of 16 columns, 2 workgroup barriers per panel:
//   (a) wave 0 factors the 16x16 diagonal block IN REGISTERS, one matrix row per
//       lane, with 16 identity rows appended (lanes 16..31) that come out as
//       X = L_pp^-T: chol_factor_diag16()
//   (b) the rows below the panel: L21 = A21 X, a 16-row tile per wave on
//       v_mfma_f64_16x16x4 (it was a forward substitution, 16 lanes per row with a
//       16-step DPP chain each: 3.3k cycles per panel, now 1.2-1.8k)
//   (c) rank-16 update of the trailing matrix with the same MFMA, one 16x16 tile
//       at a time per wave. Wave 0 takes the next diagonal tile first and factors
//       it while the others finish (look-ahead)
// The right-hand side rides along as an extra matrix row n, so L z = r is
// solved by the factorization itself; L^T d = z then goes panel by panel,
// backwards, the block's own solve being the product d_p = X w.
//
// Storage: packed lower triangle in LDS, (n+1)(n+2)/2 doubles, + the X blocks:
// n <= 178 (chol_fits_lds). Larger camera blocks use launch_cholesky_large() below
#define CHOL_PB 16
template<int N>
__device__ __forceinline__ double row_share_f64(double v)   // lane N of each 16-lane row, to the row
{
    union { double d; int i[2]; } u; u.d = v;
    // bound_ctrl with full masks: every lane is written, no need to initialize the destination
    u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x150 + N, 0xf, 0xf, true);
    u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x150 + N, 0xf, 0xf, true);
    return u.d;
}
typedef double chol_double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int lane);

// The fused step (see "dog-leg control" below) puts the end of a trial - accept
// or reject, the trust region, does the current point need its Gauss-Newton
// step - in front of the factorization, in the same launch (FINISH)
struct SolverCtlFlags;
struct Step2Dev
{
    NormalDims nd; const OpDev* ops; SolverCtl* ctl; SolverCtlFlags* fl;
    int initial; const double* comm1_tail;       // [g_S (Nc) | |x|^2 | status] behind S and r
};
__device__ bool step2_finish(const Step2Dev& sd, int* chol_status, bool no_unpack = false);        // one workgroup; true: factor
__device__ void step2_chol_done(const Step2Dev& sd, bool not_positive_definite);   // one thread

// LDS of the kernel: the packed triangle (n+1 rows), the inverse diagonal blocks, factor_diag's scratch
#define CHOL_XLD 17              // row stride of an inverse diagonal block: odd, so that 16 lanes reading a column hit 16 banks
__host__ __device__ inline int    chol_tri_doubles(int n) { return ((((n+1)*(n+2)) >> 1) + 1) & ~1; }
__host__ __device__ inline size_t chol_lds_bytes(int n)
{
    const int npanels = (n + CHOL_PB - 1)/CHOL_PB;
    return ((size_t)chol_tri_doubles(n) + (size_t)npanels*CHOL_PB*CHOL_XLD + 3*64)*sizeof(double);
}
static inline bool chol_fits_lds(int n) { return n <= 200 && chol_lds_bytes(n) <= 160*1024 - 4096; }

// The 16x16 diagonal block of a Cholesky panel, one wave, in registers. Lanes 0..15
// hold the rows of the block (lanes jb..15, beyond the end of the matrix: rows of the
// identity), lanes 16..31 the rows of an identity matrix appended below it, which
// leave as X = L^-T (what the column operations do to appended rows is to multiply
// them by L^-T from the right).
//   rowL: lane r < jb: its row of the block in LDS, 16 entries readable (the others: any readable row)
//   X:    [16][CHOL_XLD] in LDS;  cbuf: [2][64] doubles of LDS scratch
//   dstL(c): where lane r < 16 stores entry c of its row of L (a sink for what must not be stored)
// Returns true if a pivot was not positive (the factor is then garbage: NaNs).
// Per column j:
//   pivot: one v_readlane pair (the value is wave-uniform)
//   1/sqrt: hardware estimate (2^-24) + two Newton steps = 2.6e-16 relative
//     (tools/exp/rsq_f64_probe.hip; one step is 4e-15 and measures no faster here).
//     (Not positive: flagged; no select in the chain)
//   multipliers L[c][j], c > j: the next column's by readlane, at once (the next
//     pivot waits for nothing else); the others through LDS - each block lane
//     stores its entry, every lane reads the column back as broadcasts. LDS answers
//     after ~100 cycles and a wave issues in order: a wait for them would stop the
//     pivot chain too. So they are asked for as soon as column j is scaled and
//     applied at the END of column j+1, from two alternating register sets. (The
//     version before moved every multiplier with two DPP instructions: 50 VALU
//     instructions per column, issue-bound; this one has 27)
template<class DstL>
__device__ __forceinline__
bool chol_factor_diag16(const int lane, const int jb, const double* __restrict__ rowL,
                        double* __restrict__ X, double* __restrict__ cbuf, DstL dstL)
{
    const int  r16  = lane & 15;
    const bool mine = lane < 16 && r16 < jb;
#ifdef DIAG16_TS
    const long long dts_in = clock64();
#endif
    double row[CHOL_PB];
    {
        // unconditional loads, then select: no branches
        double tmp[CHOL_PB];
#pragma unroll
        for(int c = 0; c < CHOL_PB; c++) tmp[c] = rowL[c];
#pragma unroll
        for(int c = 0; c < CHOL_PB; c++)
            row[c] = mine ? ((c <= r16) ? tmp[c] : 0.0) : ((lane < 32 && c == r16) ? 1.0 : 0.0);
    }
    bool   bad = false;
    double lp[2] = {0.0, 0.0};       // this lane's scaled entry of columns j-1, j-2 (by parity)
    double Lc[2][CHOL_PB];           // those columns of the block: L[c][j-1], L[c][j-2]
#pragma unroll
    for(int c = 0; c < CHOL_PB; c++) Lc[0][c] = Lc[1][c] = 0.0;
    double* __restrict__ mycb = cbuf + lane;
#define IC(v) std::integral_constant<int,(v)>{}
    auto column = [&](auto J)
    {
        constexpr int j = decltype(J)::value;
        const double piv = readlane_f64(row[j], j);
        bad = bad || !(piv > 0.0);
        const double rd0 = __builtin_amdgcn_rsq(piv);
        const double hp  = -0.5*piv;
        const double sq  = rd0*rd0;
        const double lr  = row[j]*rd0;          // beside the chain
        const double u   = fma(hp, sq, 1.5);
        const double rd1 = rd0*u;
        const double u2  = fma(hp, rd1*rd1, 1.5);
        const double l   = (lr*u)*u2;            // block lane j: piv/sqrt(piv)
        row[j] = l;
        if constexpr(j + 2 < CHOL_PB)
        {
            // (every lane stores: no exec juggling; lanes 0..15 are the block)
            mycb[64*(j & 1)] = l;
            const double* __restrict__ cb = cbuf + 64*(j & 1);
#pragma unroll
            for(int c = j + 2; c < CHOL_PB; c++) Lc[j & 1][c] = cb[c];
            lp[j & 1] = l;
        }
        // the next pivot's column first
        if constexpr(j + 1 < CHOL_PB) row[j+1] = fma(-l, readlane_f64(l, j+1), row[j+1]);
        // what column j-1 does to the columns right of j (asked for a column ago): row[c] -= L[i][j-1] L[c][j-1]
        if constexpr(j >= 1)
        {
#pragma unroll
            for(int c = j + 1; c < CHOL_PB; c++) row[c] = fma(-lp[(j-1) & 1], Lc[(j-1) & 1][c], row[c]);
        }
    };
#ifdef DIAG16_TS
    long long dts[17];
#define CHOL_COL(j) dts[j] = clock64(); __builtin_amdgcn_sched_barrier(0); column(IC(j));
#else
    // A scheduling barrier between the columns: left alone, the compiler's scheduler treats the sixteen unrolled
    // columns as one block and interleaves them (sinking LDS reads to their uses, hoisting multiply-adds): 6160
    // cycles per call. With the columns kept apart, in the order written: see tools/exp/diag16_bench.hip
#ifndef CHOL_NO_COLUMN_BARRIER
#define CHOL_COL(j) __builtin_amdgcn_sched_barrier(0); column(IC(j));
#else
#define CHOL_COL(j) column(IC(j));
#endif
#endif
    CHOL_COL(0)  CHOL_COL(1)  CHOL_COL(2)  CHOL_COL(3)  CHOL_COL(4)  CHOL_COL(5)  CHOL_COL(6)  CHOL_COL(7)
    CHOL_COL(8)  CHOL_COL(9)  CHOL_COL(10) CHOL_COL(11) CHOL_COL(12) CHOL_COL(13) CHOL_COL(14) CHOL_COL(15)
#undef CHOL_COL
#undef IC
#ifdef DIAG16_TS
    dts[16] = clock64();
    if(lane == 0 && blockIdx.x == 0 && rowL != NULL && dts[0] % 997 == 0)
        printf("diag16 entry to column 0: %lld; per column: %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld\n",
               dts[0]-dts_in, dts[1]-dts[0], dts[2]-dts[1], dts[3]-dts[2], dts[4]-dts[3], dts[5]-dts[4], dts[6]-dts[5], dts[7]-dts[6], dts[8]-dts[7],
               dts[9]-dts[8], dts[10]-dts[9], dts[11]-dts[10], dts[12]-dts[11], dts[13]-dts[12], dts[14]-dts[13], dts[15]-dts[14], dts[16]-dts[15]);
#endif
    // L through dstL (lanes 0..15), X into its block (lanes 16..31): one store per column for the whole wave
    {
        const bool isid = (lane >= 16 && lane < 32);
        double* __restrict__ dstX = X + r16*CHOL_XLD;
#pragma unroll
        for(int c = 0; c < CHOL_PB; c++)
        {
            double* dst = isid ? dstX + c : dstL(c);
            *dst = row[c];
        }
    }
#ifdef DIAG16_TS
    { const long long dts_out = clock64(); if(lane == 0 && blockIdx.x == 0 && dts_in % 997 == 0) printf("diag16 whole call %lld cycles\n", dts_out - dts_in); }
#endif
    return bad;
}

// FINISH: 0 a factorization and solve and nothing else | 1 the end of the trial step in front, the verdict behind (sharded:
// the end-of-trial logic needs the tail summed over the ranks, which is complete only now) | 2 the verdict behind alone:
// the end-of-trial logic has run in the reduction's launch (single GPU, round 5: step2_reduce_kernel) and `skip` is its word
template<int FINISH>
__global__ __launch_bounds__(1024)
void schur_cholesky_solve_kernel(int n, const int* __restrict__ skip, int keep_factor,
                                 double* __restrict__ S, double* __restrict__ r,
                                 int* __restrict__ status, Step2Dev sd)
{
    if(skip != NULL && *skip) return;
    if constexpr(FINISH == 1) { if(!step2_finish(sd, status)) return; }
    extern __shared__ __attribute__((aligned(16))) double Mp[];
    const int t    = threadIdx.x;
    const int nt   = blockDim.x;
    const int lane = t & 63, wave = t >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // the same, known to be wave-uniform
    const int npanels = (n + CHOL_PB - 1)/CHOL_PB;
    const int r16 = lane & 15, kq = lane >> 4;

    // element (i,j), j <= i <= n, packed. Row n is the right-hand side
    auto rowptr = [&](int i) -> double* { return Mp + ((i*(i+1)) >> 1); };
    double* __restrict__ Xs   = Mp + chol_tri_doubles(n);               // [npanels][16][CHOL_XLD]: L_pp^-T of every panel
    double* __restrict__ cbuf = Xs + npanels*CHOL_PB*CHOL_XLD;          // [3][64]: factor_diag's column exchange + a sink

    __shared__ int notpd;
    if(t == 0) notpd = 0;
#ifdef CHOL_TS
    long long cts[64]; int ncts = 0;
#define CTS() do { if(t == 0 && ncts < 64) cts[ncts++] = clock64(); } while(0)
#else
#define CTS()
#endif
    CTS();

    // The lower triangle into LDS. Wave w takes rows w, w+16, ..., 64 columns per
    // lane pass, and asks for ALL of it before it stores anything: S was written
    // by other CUs a launch ago and every load is a trip across the chip (three
    // batches of 12 loads were three such trips: 3.9 us of the kernel's 48)
    {
        double v[13][4];
#pragma unroll
        for(int a = 0; a < 13; a++)
#pragma unroll
            for(int b = 0; b < 4; b++)
                if(b <= a/4)
                {
                    const int  i = wave_u + 16*a, j = lane + 64*b;
                    const bool ok = (i < n && j <= i);
                    v[a][b] = S[ok ? (size_t)i*n + j : 0];          // always a valid address: no branch around the load
                }
#pragma unroll
        for(int a = 0; a < 13; a++)
#pragma unroll
            for(int b = 0; b < 4; b++)
                if(b <= a/4)
                {
                    const int i = wave_u + 16*a, j = lane + 64*b;
                    if(i < n && j <= i) rowptr(i)[j] = v[a][b];
                }
    }
    for(int j = t; j < n; j += nt) rowptr(n)[j] = r[j];
    __syncthreads();

    // (a) diagonal block of panel p, wave 0: chol_factor_diag16() above. L back into the triangle
    // (entries up to the diagonal; the rest into a per-lane sink), X = L_pp^-T into its block
    auto factor_diag = [&](int p) __attribute__((always_inline))
    {
        const int j0 = p*CHOL_PB;
        const int jb = min(CHOL_PB, n - j0);
        const bool mine = lane < 16 && r16 < jb;
        double* __restrict__ rowL = rowptr(j0 + (mine ? r16 : 0)) + j0;
        double* __restrict__ sink = cbuf + 128 + lane;
        const bool bad = chol_factor_diag16(lane, jb, rowL, Xs + p*CHOL_PB*CHOL_XLD, cbuf,
                                            [&](int c) -> double* { return (mine && c <= r16) ? rowL + c : sink; });
        if(bad && lane == 0) notpd = 1;
    };

    // Look-ahead: the diagonal block of panel p+1 is final as soon as ONE tile of
    // panel p's trailing update is done. Wave 0 does that tile first and factors
    // the block (a long dependent chain) while waves 1..15 do the rest of the
    // update: the chain is off the critical path of everything but itself
    // (p = -1: nothing but the factorization of the first diagonal block, so
    // that there is ONE copy of that long inlined code, not a cold one for the
    // first block and another for the rest)
    CTS();
    for(int p = -1; p < npanels; p++)
    {
        const int j0 = (p < 0) ? 0 : p*CHOL_PB;
        const int jb = min(CHOL_PB, n - j0);
        const int m0 = j0 + jb;

        // (b) rows below (and the rhs row): L[i][j0..] <- A[i][j0..] X, X = L_pp^-T, on the
        // matrix cores: a tile of 16 rows per wave, 4 x v_mfma_f64_16x16x4 (A lane = A[i=l%16][k=l/16],
        // B lane = B[k=l/16][j=l%16], D register v of lane l = D[l/16 + 4v][l%16]). In place: a wave
        // has read its tile before it writes it. (As a forward substitution, 16 lanes per row with
        // a 16-step DPP chain each, this phase was 3.3k cycles of every panel's ~11k)
        if(p >= 0)
        {
            const double* __restrict__ X = Xs + p*CHOL_PB*CHOL_XLD;
            const int ntile_b = (n + 1 - m0 + 15) >> 4;
            for(int ti = wave_u; ti < ntile_b; ti += 16)
            {
                const int  ia = m0 + 16*ti + r16;
                const bool va = ia <= n;
                const double* __restrict__ pa = rowptr(va ? ia : n) + j0;
                chol_double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for(int s4 = 0; s4 < 4; s4++)
                {
                    const int  k  = 4*s4 + kq;
                    const bool vk = k < jb;
                    const double al = pa[vk ? k : 0];
                    const double av = (va && vk) ? al : 0.0;
                    const double bv = X[k*CHOL_XLD + r16];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for(int v = 0; v < 4; v++)
                {
                    const int ri = m0 + 16*ti + kq + 4*v;
                    if(ri <= n && r16 < jb) rowptr(ri)[j0 + r16] = acc[v];
                }
            }
        }
        if(p >= 0) __syncthreads();
        CTS();

        // (c) trailing update with MFMA: C[i][c] -= sum_k L[i][k] L[c][k], k in the
        //     panel; rows m0..n (incl. the rhs row), columns m0..n-1, c <= i.
        //     16x16 tiles (ta,tb), tb <= ta, dealt round-robin to the 16 waves
        {
            const int nrows = n + 1 - m0, ncols = n - m0;
            const int ntr = (p < 0) ? 0 : (nrows + 15) >> 4, ntc = (ncols + 15) >> 4;     // p = -1: no tiles
            // tile (0,0) = the next diagonal block: wave 0; tiles 1.. : waves 1..15
            // round-robin. Wave-uniform (scalar) bookkeeping: no wave walks
            // through the other waves' tiles
            int ntiles = 0;
            for(int ta = 0; ta < ntr; ta++) ntiles += min(ta + 1, ntc);
            for(int tix = wave_u; tix < ntiles; tix += (wave_u == 0 ? ntiles : 15))
            {
                int ta = 0, tb = tix;
                for(;;) { const int ntb = min(ta + 1, ntc); if(tb < ntb) break; tb -= ntb; ta++; }
                {
                    const int  ia = 16*ta + r16, ib = 16*tb + r16;
                    const bool va = ia < nrows, vb = ib < ncols;
                    const double* __restrict__ pa = rowptr(m0 + (va ? ia : 0)) + j0;
                    const double* __restrict__ pb = rowptr(m0 + (vb ? ib : 0)) + j0;
                    chol_double4_t acc;
                    double* cp[4]; bool cv[4];
#pragma unroll
                    for(int v = 0; v < 4; v++)
                    {
                        const int ri = 16*ta + kq + 4*v, cj = 16*tb + r16;
                        cv[v] = (ri < nrows) && (cj < ncols) && (cj <= ri);
                        cp[v] = rowptr(m0 + (cv[v] ? ri : 0)) + m0 + (cv[v] ? cj : 0);
                        const double cval = *cp[v];
                        acc[v] = cv[v] ? cval : 0.0;
                    }
                    // (batching several tiles per wave - all LDS reads first, MFMA
                    // chains interleaved - was measured and is slower: 12k vs 8.5k cycles)
#pragma unroll
                    for(int s4 = 0; s4 < 4; s4++)
                    {
                        const int  k  = 4*s4 + kq;
                        const bool vk = k < jb;
                        const double al = pa[k], bl = pb[k];     // k < 16: inside the row
                        const double av = (va && vk) ? -al : 0.0;
                        const double bv = (vb && vk) ?  bl : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for(int v = 0; v < 4; v++) if(cv[v]) *cp[v] = acc[v];
                }
            }
            // (wave 0 shares its SIMD with three of the updating waves: the dependent chain of the
            //  diagonal block gets the issue slots first)
            if(wave == 0 && p + 1 < npanels)
            {
                __builtin_amdgcn_s_setprio(3);
                factor_diag(p + 1);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        __syncthreads();
        CTS();
    }
    if(t == 0 && notpd) atomicExch(status, 1);


    // row n now holds z = L^-1 r. Solve L^T d = z backwards, panel by panel.
    // The same look-ahead as in the factorization: once panel p is solved, wave 0
    // updates the 16 entries of panel p-1 and solves that panel while waves 1..15
    // update everything above it. The block's own solve is d_p = X w, X = L_pp^-T
    // (upper triangular): a product, not a 16-step substitution
    double* __restrict__ z = rowptr(n);
    auto back_diag = [&](int p) __attribute__((always_inline))
    {
        const int j0 = p*CHOL_PB;
        const int jb = min(CHOL_PB, n - j0);
        const double* __restrict__ X = Xs + p*CHOL_PB*CHOL_XLD + ((lane < CHOL_PB) ? lane : 0)*CHOL_XLD;
        double xv[CHOL_PB], wv[CHOL_PB];
#pragma unroll
        for(int k = 0; k < CHOL_PB; k++) { xv[k] = X[k]; wv[k] = z[j0 + ((k < jb) ? k : 0)]; }
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for(int k = 0; k < CHOL_PB; k += 2)
        {
            acc0 += (k   < jb && k   >= lane) ? xv[k]  *wv[k]   : 0.0;
            acc1 += (k+1 < jb && k+1 >= lane) ? xv[k+1]*wv[k+1] : 0.0;
        }
        if(lane < jb) z[j0 + lane] = acc0 + acc1;
    };
    // z[i] -= sum_c L[j0+c][i] d[c]
    auto back_update = [&](int i, int j0, int jb) __attribute__((always_inline))
    {
        // Full panels (all but possibly the last): 32 unconditional LDS reads in
        // flight together, row bases wave-uniform. (A branch per term, as the
        // generic form below has, serializes the reads: one LDS latency each)
        if(jb == CHOL_PB)
        {
            const double* __restrict__ zp = z + j0;
            double lv[CHOL_PB], zv[CHOL_PB];
#pragma unroll
            for(int c = 0; c < CHOL_PB; c++) { lv[c] = rowptr(j0+c)[i]; zv[c] = zp[c]; }
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for(int c = 0; c < CHOL_PB; c += 2) { acc0 += lv[c]*zv[c]; acc1 += lv[c+1]*zv[c+1]; }
            z[i] -= acc0 + acc1;
        }
        else
        {
            double acc = 0.0;
            for(int c = 0; c < jb; c++) acc += rowptr(j0+c)[i]*z[j0+c];
            z[i] -= acc;
        }
    };
    CTS();
    // (n = 0 - a solve without camera variables - has no panel: back_diag(-1) would read and WRITE 16 doubles in
    //  front of the triangle, i.e. this kernel's own flags in LDS; found when a change of the LDS layout made
    //  such solves report "not positive definite" at random.
    //  Measured and dropped: the whole sweep by wave 0 alone, no barriers - 40.3 us against 39.0: what a panel of
    //  the sweep costs is its chain of LDS round trips, not the barrier)
    if(wave == 0 && npanels > 0) back_diag(npanels-1);
    __syncthreads();
    CTS();
    for(int p = npanels-1; p >= 1; p--)
    {
        const int j0 = p*CHOL_PB;
        const int jb = min(CHOL_PB, n - j0);
        if(wave == 0)
        {
            if(lane < CHOL_PB) back_update(j0 - CHOL_PB + lane, j0, jb);     // panel p-1 is full
            back_diag(p - 1);
        }
        else
            for(int i = t - 64; i < j0 - CHOL_PB; i += nt - 64) back_update(i, j0, jb);
        __syncthreads();
        CTS();
    }
    // r <- -d ; keep the factor for later solves (uncertainty, solve_xt_JtJ_bt)
    for(int i = t; i < n; i += nt) r[i] = -z[i];
    if(keep_factor)
        for(int idx = t; idx < n*n; idx += nt)
        {
            const int i = idx / n, j = idx - i*n;
            if(j <= i) S[(size_t)i*n + j] = rowptr(i)[j];
        }
    CTS();
    if constexpr(FINISH != 0) { if(t == 0) step2_chol_done(sd, notpd != 0); }
#ifdef CHOL_TS
    if(t == 0) { printf("chol ts (load | diag0 | b,c per panel ... | backward | store):"); for(int i=1;i<ncts;i++) printf(" %lld", cts[i]-cts[i-1]); printf("\n"); }
#endif
}

// The same in place in global memory, row-major, one workgroup: the plain
// right-looking blocked algorithm. Only a fallback for callers without the
// panel workspace; camera blocks that do not fit the LDS normally go through
// launch_cholesky_large() below (this kernel takes 100 ms at 1206 variables,
// that path 1.7 ms)
__global__ __launch_bounds__(1024)
void schur_cholesky_solve_global_kernel(int n, const int* __restrict__ skip,
                                        double* __restrict__ S, double* __restrict__ r,
                                        int* __restrict__ status)
{
    if(skip != NULL && *skip) return;
    const int t  = threadIdx.x;
    const int nt = blockDim.x;
    auto at = [&](int i, int j) -> double& { return (i == n) ? r[j] : S[(size_t)i*n + j]; };
    __shared__ int notpd;
    if(t == 0) notpd = 0;
    __syncthreads();
    constexpr int PB = 8;
    for(int j0 = 0; j0 < n; j0 += PB)
    {
        const int jb = min(PB, n - j0);
        for(int jj = 0; jj < jb; jj++)
        {
            const int j = j0 + jj;
            if(t == 0)
            {
                double d = at(j,j);
                if(!(d > 0.0)) { notpd = 1; d = 1.0; }
                at(j,j) = sqrt(d);
            }
            __syncthreads();
            const double djj = at(j,j);
            for(int i = j + 1 + t; i <= n; i += nt) at(i,j) /= djj;
            __syncthreads();
            const int ncols = jb - jj - 1;
            for(int idx = t; idx < ncols*(n - j); idx += nt)
            {
                const int cc = idx % ncols, ii = idx / ncols;
                const int c = j + 1 + cc, i = j + 1 + ii;
                if(i >= c) at(i,c) -= at(i,j)*at(c,j);
            }
            __syncthreads();
        }
        const int m0 = j0 + jb;
        const int nm = n - m0;
        for(int idx = t; idx < (nm+1)*nm; idx += nt)
        {
            const int ii = idx / nm, cc = idx - ii*nm;
            if(cc > ii) continue;
            const int i = m0 + ii, c = m0 + cc;
            double acc = 0.0;
#pragma unroll
            for(int kk = 0; kk < PB; kk++)
                if(kk < jb) acc += at(i, j0+kk)*at(c, j0+kk);
            at(i,c) -= acc;
        }
        __syncthreads();
    }
    if(t == 0 && notpd) atomicExch(status, 1);
    // r = z. L^T d = z, column-oriented
    __shared__ double piv;
    for(int j=n-1;j>=0;j--)
    {
        if(t == 0) { piv = r[j]/at(j,j); r[j] = piv; }
        __syncthreads();
        const double pj = piv;
        for(int i=t;i<j;i+=nt) r[i] -= at(j,i)*pj;
        __syncthreads();
    }
    for(int i=t;i<n;i+=nt) r[i] = -r[i];
}

////////////////////////////////////////////////////////////////////////////////
// Large camera blocks (the splined models: Nc = 4 + 2 Nx Ny + ..., ~1200): the
// same factorization and solve as a sequence of launches over the WHOLE device.
// M = [S ; r] is (n+1) x n row-major (r is stored right behind S: the right-hand
// side is row n and rides through the factorization, as in the LDS kernel).
// Right-looking, panels of LCH_NB = 64 columns; per panel three launches:
//   diag   1 workgroup   L11 = chol(M11) in LDS, and its inverse (kept: the
//                        backward solve needs it again)
//   trsm   1 workgroup / 64 rows below    L21 = M21 L11^-T   (a small GEMM with the inverse)
//   syrk   1 workgroup / 32x32 tile of the trailing matrix, M22 -= L21 L21^T, v_mfma_f64_16x16x4
// then ONE launch solves L^T d = z panel by panel, backwards (z = row n), r <- -d.
// ~20 x 3 launches and ~0.6 GFLOP for n = 1200: about a millisecond, where the
// one-workgroup fallback above takes 160 ms
////////////////////////////////////////////////////////////////////////////////
#define LCH_NB 64
template<int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr(I < N) { f(std::integral_constant<int,I>{}); static_for<I+1,N>(f); }
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
// The 64x64 diagonal block of a panel: L11 L11^T = M11 and X = L11^-1.
// One workgroup of 1024 factors the 128x64 matrix [M11; I] by columns, blocked
// by 16: what the factorization does to rows appended below the matrix is to
// multiply them by L^-T from the right (the way schur_cholesky_solve_kernel gets
// L^-1 r out of its rhs row), so the identity comes out as L^-T = X^T. The same
// three phases per 16 columns as in schur_cholesky_solve_kernel, on a square
// LDS array:
//   (a) wave 0: chol_factor_diag16() - the 16x16 block in registers, and its own
//       inverse transpose X_pp from the identity lanes
//   (b) the rows below (to row 127): A[i][panel] <- A[i][panel] X_pp, a 16-row tile
//       per wave on v_mfma_f64_16x16x4
//   (c) rank-16 update of the columns to the right with the same MFMA; wave 0 takes
//       the next diagonal tile first and factors it while the others finish
// (Measured history of this kernel: 256 threads with a barrier per column 160 us;
// one wave with the whole block in registers 65 us; one wave blocked by 16 53 us;
// 1024 threads with a readlane factorization of the 16x16 block, a substitution
// chain per row for (b) and scalar FMAs for (c): 20-39 us. This: see DESIGN.md)
#define LCH_PB 16
// 16 x 16 tile (wi, wc) of A B^T over k < kmax (a multiple of 16), A and B 64 x 64 in LDS with row stride 65.
// Register v of lane l: row 16 wi + l/16 + 4 v, column 16 wc + l%16.
// B = X, lower triangular: tile column wc has nothing beyond k = 16 wc + 15. A v_mfma_f64_16x16x4 is ~110 cycles
// and the four waves of a SIMD take turns: a workgroup's 64^3 product is 7000 cycles of ONE CU. So the callers
// deal the tiles so that every SIMD (wave % 4) gets every tile column: 4+8+12+16 instructions instead of 4 x 16
__device__ __forceinline__
syrk_d4 lch_tile_ABt(const double* __restrict__ A, const double* __restrict__ B, int wi, int wc, int r16, int kq, bool negate,
                     int kmax = LCH_NB)
{
    syrk_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for(int k1 = 0; k1 < kmax; k1 += 16)
#pragma unroll
        for(int k0 = k1; k0 < k1 + 16; k0 += 4)
        {
            const double av = A[(16*wi + r16)*(LCH_NB+1) + k0 + kq];
            const double bv = B[(16*wc + r16)*(LCH_NB+1) + k0 + kq];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -av : av, bv, acc, 0, 0, 0);
        }
    return acc;
}
// tile (wi, wc) of a wave for a product with X: the SIMD is wave % 4 = wi, so each SIMD has one tile of every column
__device__ __forceinline__ void lch_tile_for_X(int wave, int* wi, int* wc) { *wi = wave & 3; *wc = wave >> 2; }
// With Xprev: the block is first brought up to date with the PREVIOUS panel (columns jprev .. jprev+63), whose
// trailing update runs beside this workgroup in the same launch (lchol_update_kernel, which leaves this block
// alone):  Lb = M[block rows][previous panel] Xprev^T,  block -= Lb Lb^T.  (Lb is not stored: the tile workgroups
// of this launch read the same rows of the previous panel as they are. lchol_trsm of the next launch stores it)
#define LCH_LDS_DOUBLES (2*LCH_NB*(LCH_NB+1) + CHOL_PB*CHOL_XLD + 3*64 + LCH_NB*(LCH_NB+1))
// threads of the panel kernels' workgroups. What they do is bound by ONE CU's matrix pipes and by wave 0's pivot
// chain, not by the number of waves; with 1024 threads a wave has 128 registers and chol_factor_diag16() spills
#ifndef LCH_THREADS
#define LCH_THREADS 512
#endif
#define LCH_NW  (LCH_THREADS/64)     // waves
#define LCH_TPW (16/LCH_NW)          // 16 x 16 tiles of a 64 x 64 block per wave
__device__ __forceinline__
void lchol_diag_block(int n, double* __restrict__ M, int j0,
                      double* __restrict__ Linv /* [LCH_NB][LCH_NB] of this panel */, int* __restrict__ status,
                      const double* __restrict__ Xprev, int jprev, double* __restrict__ lds /* LCH_LDS_DOUBLES, 16-byte aligned */,
                      const double* __restrict__ E1 = NULL, int ld1 = 0, const double* __restrict__ E2 = NULL, int ld2 = 0
                      /* (round 5, the separator's first block: what the two sides' chains left for it in their borders
                          - element (i, j) of the block at E[i ld + j] - is added as the block is loaded: (M + E1) + E2) */)
{
    constexpr int NB = LCH_NB, NR = 2*LCH_NB, LD = LCH_NB + 1;
    static_assert(LCH_PB == CHOL_PB, "chol_factor_diag16() is the block factorization");
    double* __restrict__ A  = lds;                       // rows 0..63: the block; rows 64..127: the identity -> L^-T
    double* __restrict__ Xb = A + NR*LD;                 // [CHOL_PB][CHOL_XLD] L_pp^-T of the current 16 columns
    double* __restrict__ cb = Xb + CHOL_PB*CHOL_XLD;     // [3*64] chol_factor_diag16's exchange + a sink
    double* __restrict__ Pm = cb + 3*64;                 // [NB][LD] this block's rows of the previous panel
    __shared__ int    notpd;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, kq = lane >> 4;
    const int nb = min(NB, n - j0);
#ifdef LCH_TS
    long long ts[6]; ts[0] = clock64();
#endif
    if(t == 0) notpd = 0;
    // the block, padded with the identity (so that a short last panel factors too); with Xprev, this block's rows
    // of the previous panel and Xprev too (rows 64.. of A are free until the factorization starts: the identity is
    // written there afterwards). EVERY load first, from addresses that are always valid, then the selects and the
    // stores: loads under a condition compile to branches with a wait each, three memory round trips instead of one
    {
        constexpr int NIT = NB*NB/LCH_THREADS;
        double va[NIT], vp[NIT], vx[NIT];
        double* __restrict__ Xp = A + NB*LD;
#pragma unroll
        for(int u = 0; u < NIT; u++)
        {
            const int idx = t + LCH_THREADS*u;
            const int i = idx / NB, j = idx - i*NB;
            const bool ina = (i < nb && j < nb && j <= i);
            va[u] = M[ina ? (size_t)(j0+i)*n + j0 + j : (size_t)0];
            if(E1 != NULL) va[u] = (va[u] + E1[ina ? (size_t)i*ld1 + j : (size_t)0]) + E2[ina ? (size_t)i*ld2 + j : (size_t)0];
            if(Xprev != NULL)
            {
                vp[u] = M[(i < nb) ? (size_t)(j0+i)*n + jprev + j : (size_t)0];
                vx[u] = Xprev[idx];
            }
        }
#pragma unroll
        for(int u = 0; u < NIT; u++)
        {
            const int idx = t + LCH_THREADS*u;
            const int i = idx / NB, j = idx - i*NB;
            const bool ina = (i < nb && j < nb && j <= i);
            A[i*LD + j] = ina ? va[u] : ((i == j) ? 1.0 : 0.0);
            if(Xprev != NULL) { Pm[i*LD + j] = (i < nb) ? vp[u] : 0.0; Xp[i*LD + j] = vx[u]; }
            else              A[(NB + i)*LD + j] = (i == j) ? 1.0 : 0.0;
        }
    }
#ifdef LCH_TS
    ts[1] = clock64();
#endif
    if(Xprev != NULL)
    {
        double* __restrict__ Xp = A + NB*LD;
        __syncthreads();
        syrk_d4 lb[LCH_TPW];
#pragma unroll
        for(int u = 0; u < LCH_TPW; u++)
        {
            int wi, wc;
            lch_tile_for_X(wave_u + LCH_NW*u, &wi, &wc);
            lb[u] = lch_tile_ABt(Pm, Xp, wi, wc, r16, kq, false, 16*(wc + 1));
        }
        __syncthreads();
#pragma unroll
        for(int u = 0; u < LCH_TPW; u++)
        {
            int wi, wc;
            lch_tile_for_X(wave_u + LCH_NW*u, &wi, &wc);
#pragma unroll
            for(int v = 0; v < 4; v++) Pm[(16*wi + kq + 4*v)*LD + 16*wc + r16] = lb[u][v];
        }
        __syncthreads();
        // the ten tiles of the lower triangle: wave 0 takes the first alone - it is all the first 16 x 16 block
        // factorization needs, which then starts without waiting for the others, who share the other nine
        for(int tix = wave_u; tix < (wave_u == 0 ? 1 : 10); tix += LCH_NW - 1)
        {
            int ti = 0, tj = tix;
            while(tj > ti) { tj -= ti + 1; ti++; }
            const syrk_d4 d = lch_tile_ABt(Pm, Pm, ti, tj, r16, kq, true);
#pragma unroll
            for(int v = 0; v < 4; v++) A[(16*ti + kq + 4*v)*LD + 16*tj + r16] += d[v];
        }
        // (rows 64.. held Xprev, read for the last time two barriers ago; nobody reads them before the next one)
        for(int idx = t; idx < NB*NB; idx += LCH_THREADS)
        {
            const int i = idx / NB, j = idx - i*NB;
            A[(NB + i)*LD + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    else __syncthreads();

#ifdef LCH_TS
    ts[2] = clock64();
#endif
    auto diag = [&](int base) __attribute__((always_inline))
    {
        double* __restrict__ rowL = &A[(base + r16)*LD + base];
        double* __restrict__ sink = cb + 128 + lane;
        // (the entries right of the diagonal are stored too, as zeros: nothing reads them)
        const bool bad = chol_factor_diag16(lane, CHOL_PB, rowL, Xb, cb,
                                            [&](int c) -> double* { return (lane < 16) ? rowL + c : sink; });
        if(bad && lane == 0) notpd = 1;
    };
#ifdef LCH_TS
    long long tsd = 0, tsb = 0, tsc = 0, tsy = 0, tq = clock64(), tq1;
#define LCH_TICK(w) { tq1 = clock64(); w += tq1 - tq; tq = tq1; }
#else
#define LCH_TICK(w)
#endif
    if(wave == 0) diag(0);
    LCH_TICK(tsd)
    __syncthreads();
    LCH_TICK(tsy)

#pragma unroll 1
    for(int base = 0; base < NB; base += LCH_PB)
    {
        const int m0 = base + LCH_PB;
        // (b) rows m0 .. 127
        for(int ti = wave_u; ti < (NR - m0)/16; ti += LCH_NW)
        {
            double* __restrict__ pa = &A[(m0 + 16*ti + r16)*LD + base];
            chol_double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for(int s4 = 0; s4 < 4; s4++)
            {
                const int k = 4*s4 + kq;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k], Xb[k*CHOL_XLD + r16], acc, 0, 0, 0);
            }
#pragma unroll
            for(int v = 0; v < 4; v++) A[(m0 + 16*ti + kq + 4*v)*LD + base + r16] = acc[v];
        }
        LCH_TICK(tsb)
        __syncthreads();
        LCH_TICK(tsy)
        // (c) tiles (ta, tb), tb <= ta, of rows m0.. x columns m0..63; (0,0) = the next diagonal block: wave 0
        {
            const int ntr = (NR - m0)/16, ntc = (NB - m0)/16;
            int ntiles = 0;
            for(int ta = 0; ta < ntr; ta++) ntiles += min(ta + 1, ntc);
            for(int tix = wave_u; tix < ntiles; tix += (wave_u == 0 ? ntiles : LCH_NW - 1))
            {
                int ta = 0, tb = tix;
                for(;;) { const int ntb = min(ta + 1, ntc); if(tb < ntb) break; tb -= ntb; ta++; }
                const double* __restrict__ pa = &A[(m0 + 16*ta + r16)*LD + base];
                const double* __restrict__ pb = &A[(m0 + 16*tb + r16)*LD + base];
                double* __restrict__ pc = &A[(m0 + 16*ta + kq)*LD + m0 + 16*tb + r16];
                chol_double4_t acc;
#pragma unroll
                for(int v = 0; v < 4; v++) acc[v] = pc[4*v*LD];
#pragma unroll
                for(int s4 = 0; s4 < 4; s4++)
                {
                    const int k = 4*s4 + kq;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[k], pb[k], acc, 0, 0, 0);
                }
#pragma unroll
                for(int v = 0; v < 4; v++) pc[4*v*LD] = acc[v];
            }
            LCH_TICK(tsc)
            if(wave == 0 && m0 < NB) diag(m0);
            LCH_TICK(tsd)
        }
        __syncthreads();
        LCH_TICK(tsy)
    }
#ifdef LCH_TS
    ts[3] = clock64();
#endif
    // L back into the matrix; X[i][k] = (L^-T)[k][i] = row 64+k, column i
    for(int idx = t; idx < NB*NB; idx += LCH_THREADS)
    {
        const int i = idx / NB, j = idx - i*NB;
        if(i < nb && j < nb && j <= i) M[(size_t)(j0+i)*n + j0 + j] = A[i*LD + j];
        Linv[idx] = (j <= i) ? A[(NB + j)*LD + i] : 0.0;
    }
    if(t == 0 && notpd) atomicExch(status, 1);
#ifdef LCH_TS
    ts[4] = clock64();
    if((t == 0 || t == 64*5) && (j0 == 128 || j0 == 640)) printf("lchol diag j0 %d t %d: issue loads %lld, pre-update %lld, factor %lld (diag16 %lld, b %lld, c %lld, barriers %lld), store %lld cycles\n", j0, t, ts[1]-ts[0], ts[2]-ts[1], ts[3]-ts[2], tsd, tsb, tsc, tsy, ts[4]-ts[3]);
#endif
}
// with_finish (round 5): the end of the trial step (step2_finish: one workgroup anyway) opens this launch instead of
// being a launch of its own in front of it - 4.8 us of pure launch on every trial step of a big camera block; what it
// decides (fl->skip_chol, which is what `skip` points at) is what the panel launches behind this one read
__device__ __forceinline__ int lchol_n(const int* __restrict__ n_dev, int n_host);
__global__ __launch_bounds__(LCH_THREADS)
void lchol_diag_kernel(const int* __restrict__ n_dev, int n_host, const int* __restrict__ skip, double* __restrict__ M, int j0,
                       double* __restrict__ Linv, int* __restrict__ status, int with_finish, Step2Dev sd,
                       const double* __restrict__ iso, int iso_Nc, unsigned* __restrict__ tail_counter)
{
    // (lchol_tail_kernel's barrier counts from zero: cleared here, launches ahead of it, instead of by a memset of its own)
    if(tail_counter != NULL && threadIdx.x == 0) *tail_counter = 0u;
    if(with_finish) { if(!step2_finish(sd, status)) return; }
    else if(skip != NULL && *skip) return;
    __shared__ __attribute__((aligned(16))) double lds[LCH_LDS_DOUBLES];
    const int n = lchol_n(n_dev, n_host);
    // (the isolated pairs of a compacted camera block, LcholCompact: a pair that is not positive definite is this
    //  factorization's failure like a pivot of the big matrix; found here, a thread a pair, so that the status is final
    //  when the last launch reads it)
    if(iso != NULL)
        for(int p0 = n + 2*(int)threadIdx.x; p0 < iso_Nc; p0 += 2*LCH_THREADS)
        {
            const double* __restrict__ b = iso + (size_t)4*((p0 - n) >> 1);
            const bool two = p0 + 1 < iso_Nc;
            const double s00 = b[0], s10 = two ? b[1] : 0.0, s11 = two ? b[2] : 1.0;
            if(!(s00 > 0.0) || !(s11 - s10*(s10/s00) > 0.0)) atomicExch(status, 1);
        }
    lchol_diag_block(n, M, j0, Linv, status, NULL, 0, lds);
}

// One launch per panel (lchol_panel_kernel), three kinds of workgroup side by side:
//   [0]  lchol_diag_kernel's work for the NEXT panel: its diagonal block, updated with this panel, factored
//   [1 .. ntiles]  the trailing update of THIS panel by 64 x 64 tiles - each tile workgroup makes the rows of
//        L21 = M21 X^T it needs itself (two 64 x 64 x 64 products on the MFMA, a few hundred ns) instead of
//        waiting for a panel-solve launch;  M21 is left as it is while they read it
//   [.. + ntrsm]  the panel solve of the PREVIOUS panel, in place: nobody reads those columns any more
// The chain per panel was diagonal block -> panel solve -> trailing update, three dependent launches of 13 + 9 +
// 8 us and the gaps between them; it is one launch as long as the longest of the three kinds
// rows of the trailing matrix come in blocks of 64; the rhs row n is a block of its own (the last)
__device__ __forceinline__
void lch_tile_of(int q, int nbt, int* bi, int* bj, int incl00 = 0)
{
    // q = 0 ..: the pairs (bi, bj), bj <= bi < nbt, but (0, 0) [the diagonal workgroup's; incl00: with it - the last panel
    // of a chain that stops in front of its border, LcholPlan::own]; then (nbt, bj), bj < nbt
    const int skip = incl00 ? 0 : 1;
    const int ntri = nbt*(nbt + 1)/2 - skip;
    if(q >= ntri) { *bi = nbt; *bj = q - ntri; return; }
    int i = 0, p = q + skip;
    while(p > i) { p -= i + 1; i++; }
    *bi = i; *bj = p;
}
__device__ __forceinline__
void lchol_update_tile(int n, double* __restrict__ M, int j0, const double* __restrict__ X, int q,
                       double* __restrict__ MI, double* __restrict__ MC, double* __restrict__ Xs, int incl00 = 0)
{
    constexpr int NB = LCH_NB, LD = LCH_NB + 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    // the wave's tiles u = 0 .. LCH_TPW-1: (wi, wc) of the 64 x 64 tile
    auto tile_of = [&](int u, int* wi, int* wc) { const int w = wave_u + LCH_NW*u; *wi = w >> 2; *wc = w & 3; };
    const int m0 = j0 + NB;
    const int nbt = (n - m0 + NB - 1)/NB;
    int bi, bj;
    lch_tile_of(q, nbt, &bi, &bj, incl00);
#ifdef LCH_TS
    const long long tt0 = clock64();
#endif
    const bool rhs  = (bi == nbt), diag = (bi == bj);
    const int  i0   = rhs ? n : m0 + NB*bi, c0 = m0 + NB*bj;
    const int  ni   = rhs ? 1 : min(NB, n - i0), nc = min(NB, n - c0);     // rows of M in the two blocks
    // the tile itself, asked for first
    double tile[LCH_TPW][4];
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int wi, wc; tile_of(u, &wi, &wc);
#pragma unroll
        for(int v = 0; v < 4; v++)
        {
            const int i = 16*wi + kq + 4*v, c = 16*wc + r16;
            const bool ok = i < ni && c < nc && (!diag || c <= i);
            const double mv = M[ok ? (size_t)(i0 + i)*n + c0 + c : (size_t)0];
            tile[u][v] = ok ? mv : 0.0;
        }
    }
    {
        // (all loads, from addresses that are always valid; then the stores: see lchol_diag_block)
        constexpr int NIT = NB*NB/LCH_THREADS;
        double vi[NIT], vc[NIT], vx[NIT];
#pragma unroll
        for(int u = 0; u < NIT; u++)
        {
            const int idx = t + LCH_THREADS*u;
            const int i = idx / NB, k = idx - i*NB;
            vi[u] = M[(size_t)(i0 + (i < ni ? i : 0))*n + j0 + k];
            vc[u] = diag ? 0.0 : M[(size_t)(c0 + (i < nc ? i : 0))*n + j0 + k];
            vx[u] = X[idx];
        }
#pragma unroll
        for(int u = 0; u < NIT; u++)
        {
            const int idx = t + LCH_THREADS*u;
            const int i = idx / NB, k = idx - i*NB;
            MI[i*LD + k] = (i < ni) ? vi[u] : 0.0;
            if(!diag) MC[i*LD + k] = (i < nc) ? vc[u] : 0.0;
            Xs[i*LD + k] = vx[u];
        }
    }
    __syncthreads();
    syrk_d4 li[LCH_TPW], lc[LCH_TPW];
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int xi, xc;
        lch_tile_for_X(wave_u + LCH_NW*u, &xi, &xc);
        li[u] = lch_tile_ABt(MI, Xs, xi, xc, r16, kq, false, 16*(xc + 1));
        lc[u] = li[u];
        if(!diag) lc[u] = lch_tile_ABt(MC, Xs, xi, xc, r16, kq, false, 16*(xc + 1));
    }
    __syncthreads();
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int xi, xc;
        lch_tile_for_X(wave_u + LCH_NW*u, &xi, &xc);
#pragma unroll
        for(int v = 0; v < 4; v++)
        {
            MI[(16*xi + kq + 4*v)*LD + 16*xc + r16] = li[u][v];
            if(!diag) MC[(16*xi + kq + 4*v)*LD + 16*xc + r16] = lc[u][v];
        }
    }
    __syncthreads();
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int wi, wc; tile_of(u, &wi, &wc);
        const syrk_d4 d = lch_tile_ABt(MI, diag ? MI : MC, wi, wc, r16, kq, true);
#pragma unroll
        for(int v = 0; v < 4; v++)
        {
            const int i = 16*wi + kq + 4*v, c = 16*wc + r16;
            if(i < ni && c < nc && (!diag || c <= i)) M[(size_t)(i0 + i)*n + c0 + c] = tile[u][v] + d[v];
        }
    }
#ifdef LCH_TS
    if(t == 0 && (j0 == 64 || j0 == 576) && (q == 0 || q == 40)) printf("lchol tile j0 %d q %d: %lld cycles\n", j0, q, clock64() - tt0);
#endif
}
// rows r0 .. r0+63 (up to the rhs row n) of a panel:  L21 = M21 L11^-T, in place
__device__ __forceinline__
void lchol_trsm_block(int n, double* __restrict__ M, int j0, const double* __restrict__ X, int b,
                      double* __restrict__ MI, double* __restrict__ Xs, double* __restrict__ zc)
{
    constexpr int NB = LCH_NB, LD = LCH_NB + 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int nb = min(NB, n - j0);
    const int r0 = j0 + nb + b*NB;
    for(int idx = t; idx < NB*NB; idx += LCH_THREADS)
    {
        const int i = idx / NB, k = idx - i*NB;
        MI[i*LD + k] = (r0 + i <= n && k < nb) ? M[(size_t)(r0 + i)*n + j0 + k] : 0.0;
        Xs[i*LD + k] = X[idx];
    }
    __syncthreads();
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int wi, wc;
        lch_tile_for_X(wave_u + LCH_NW*u, &wi, &wc);
        const syrk_d4 l = lch_tile_ABt(MI, Xs, wi, wc, r16, kq, false, 16*(wc + 1));
#pragma unroll
        for(int v = 0; v < 4; v++)
        {
            const int i = 16*wi + kq + 4*v, j = 16*wc + r16;
            if(r0 + i <= n && j < nb) M[(size_t)(r0 + i)*n + j0 + j] = l[v];
            // (the solved right-hand side z = L^-1 r once more, outside the matrix: lchol_apply_inverse_kernel reads
            //  it while it writes the solution over row n)
            if(r0 + i == n && j < nb && zc != NULL) zc[j0 + j] = l[v];
        }
    }
}

// ---- L^-1 on the side (round 4), so that the solve needs no backward sweep.
// The sweep L^T d = z is a chain of panels again, and every link wants the whole block column of L below it: one
// workgroup streams it at ~50 GB/s, so it went in four groups of panels - seven launches, 103 us of configuration 2's
// 930. Instead Y = L^-1 (lower triangular, 64 x 64 blocks Y_pq, p >= q, Y_pp = X_p) is built WHILE the panels are
// factored, in workgroups of the same launches on CUs that have nothing to do, and the solve ends with one product
// d = -Y^T z. Column block q of Y is a forward substitution of its own:
//     Y_pq = -X_p ( T_pq + L_{p,p-1} Y_{p-1,q} ),      T_pq = sum_{k=q}^{p-2} L_pk Y_kq
//   chain workgroup (p, q), in the launch after panel p-1 was solved (launch p+1): the two products above; T_pq is
//     complete by then and lives where Y_pq goes
//   tile workgroup (p, q, k), p >= k+2, in the launch after row k of Y was made (launch k+2): T_pq (+)= L_pk Y_kq;
//     the first contribution (k == q) writes, the others add - one launch apart each, and never in the launch in which
//     the chain reads T_pq (its last tile contribution, k = p-2, is one launch earlier)
// Twice the flops of the factorization, none of them on its critical path: a chain workgroup is two 64^3 products
// (14k cycles of a 43k-cycle launch), a launch has at most ~100 of the tiles
// Yb: [npad][npad] row-major, npad = 64 npanels. Blocks of L come from M (rows >= n: zero), X_p from Linv
__device__ __forceinline__
void lchol_inverse_block(int n, int npad, const double* __restrict__ M, const double* __restrict__ Linv, double* __restrict__ Yb,
                         int p, int q, int k, bool chain, double* __restrict__ LA, double* __restrict__ LB, double* __restrict__ LX)
{
    constexpr int NB = LCH_NB, LD = LCH_NB + 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    auto tile_of = [&](int u, int* wi, int* wc) { const int w = wave_u + LCH_NW*u; *wi = w >> 2; *wc = w & 3; };
    // chain: k = p-1. A = L_pk (rows of block p, columns of block k), B = Y_kq, transposed into LDS for A B
    const double* __restrict__ Ysrc = (k == q) ? Linv + (size_t)k*NB*NB : Yb + (size_t)k*NB*npad + (size_t)q*NB;
    const int ldy = (k == q) ? NB : npad;
    double* __restrict__ Tdst = Yb + (size_t)p*NB*npad + (size_t)q*NB;
    const bool have_T = chain ? (q <= p - 2) : (k > q);
    {
        constexpr int NIT = NB*NB/LCH_THREADS;
        double va[NIT], vb[NIT], vx[NIT];
#pragma unroll
        for(int u = 0; u < NIT; u++)
        {
            const int idx = t + LCH_THREADS*u;
            const int i = idx / NB, c = idx - i*NB;
            const int row = p*NB + i;
            va[u] = M[(size_t)(row < n ? row : 0)*n + k*NB + c];         // (block k is a full block: k < p)
            if(row >= n) va[u] = 0.0;
            vb[u] = Ysrc[(size_t)i*ldy + c];
            vx[u] = chain ? Linv[(size_t)p*NB*NB + idx] : 0.0;
        }
#pragma unroll
        for(int u = 0; u < NIT; u++)
        {
            const int idx = t + LCH_THREADS*u;
            const int i = idx / NB, c = idx - i*NB;
            LA[i*LD + c] = va[u];
            // (X_k as Y_kk: its upper triangle is not kept clean by the factorization: only j <= i counts)
            LB[c*LD + i] = (k == q && c > i) ? 0.0 : vb[u];
            if(chain) LX[i*LD + c] = (c <= i) ? vx[u] : 0.0;
        }
    }
    __syncthreads();
    syrk_d4 acc[LCH_TPW];
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int wi, wc; tile_of(u, &wi, &wc);
        acc[u] = lch_tile_ABt(LA, LB, wi, wc, r16, kq, false);
        if(have_T)
        {
#pragma unroll
            for(int v = 0; v < 4; v++) acc[u][v] += Tdst[(size_t)(16*wi + kq + 4*v)*npad + 16*wc + r16];
        }
    }
    if(!chain)
    {
#pragma unroll
        for(int u = 0; u < LCH_TPW; u++)
        {
            int wi, wc; tile_of(u, &wi, &wc);
#pragma unroll
            for(int v = 0; v < 4; v++) Tdst[(size_t)(16*wi + kq + 4*v)*npad + 16*wc + r16] = acc[u][v];
        }
        return;
    }
    // Y_pq = -X_p (T + L Y): the sum, transposed, where A was
    __syncthreads();
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int wi, wc; tile_of(u, &wi, &wc);
#pragma unroll
        for(int v = 0; v < 4; v++) LA[(16*wc + r16)*LD + 16*wi + kq + 4*v] = acc[u][v];
    }
    __syncthreads();
#pragma unroll
    for(int u = 0; u < LCH_TPW; u++)
    {
        int wi, wc; tile_of(u, &wi, &wc);
        // (X_p is lower triangular: row block wi has nothing beyond k = 16 wi + 15)
        const syrk_d4 y = lch_tile_ABt(LX, LA, wi, wc, r16, kq, true, 16*(wi + 1));
#pragma unroll
        for(int v = 0; v < 4; v++) Tdst[(size_t)(16*wi + kq + 4*v)*npad + 16*wc + r16] = y[v];
    }
}
// block 0: the next panel's diagonal block (if there is a next panel); then the tiles; then the previous panel's solve
// what a launch does for Y = L^-1 (lchol_inverse_block): the chain of row block prow (prow workgroups, q < prow) and the
// tiles fed by row block krow (targets p = krow+2 .. npanels-1, q <= krow); -1: none
struct LcholInverseWork { double* Yb; double* zc; const double* Linv; int npad, npanels, prow, krow, nchain, ntile; };
// What launch l = 0 .. npanels of the factorization of an n x n matrix does (round 5: one function for the host, which
// sizes the grid with it, and for the kernel, which may learn n only on the device - the splined models' camera block
// without the control points no board covers, launch_cholesky_large(n_dev) - and then finds itself to be a panel's
// launch, the closing one (l == npanels: the last panel's solve and the last row block of L^-1) or nothing (l > npanels)):
//   panel l < npanels:  [0] the next diagonal block | the trailing update's tiles | the previous panel's solve | L^-1
// own (round 5, the nested-dissection chains): the first `own` panels alone are factored - the matrix's other rows and
// columns are a BORDER that takes the panels' updates and is somebody else's to factor; L^-1 is made for those panels
// alone (its workspace: [own][64][64] | Yb [64 own][64 own] | zc) and launch l = own is the closing one. -1: all of them
struct LcholPlan
{
    int npanels, npad;  // (the panels that are factored, and 64 x that)
    int j0;             // first column of panel l
    int has_next;       // there is a diagonal block behind panel l (workgroup 0 factors it)
    int incl00;         // no next block of its own, but a border: the tile behind the panel is a tile like the others
    int ntiles, ntrsm, jprev, pprev;
    int prow, krow, nchain, ntile;
    int nblocks;        // workgroups of the launch (0: nothing to do)
};
__host__ __device__ inline LcholPlan lchol_plan(int n, int l, bool with_inverse, int own = -1)
{
    LcholPlan q;
    q.npanels = (n + LCH_NB - 1)/LCH_NB;
    if(own >= 0 && own < q.npanels) q.npanels = own;
    q.npad    = q.npanels*LCH_NB;
    q.j0 = 0; q.has_next = 0; q.incl00 = 0; q.ntiles = 0; q.ntrsm = 0; q.jprev = 0; q.pprev = 0;
    q.prow = l - 1; q.krow = l - 2; q.nchain = 0; q.ntile = 0; q.nblocks = 0;
    if(l > q.npanels || n <= 0) return q;
    // rows below panel p, the rhs row included, in blocks of 64 (the panel solve's)
    auto ntrsm_of = [&](int p) { const int m0 = (n < (p + 1)*LCH_NB) ? n : (p + 1)*LCH_NB; return (n + 1 - m0 + LCH_NB - 1)/LCH_NB; };
    if(with_inverse)
    {
        // the chain of row block l-1 of Y, the tiles fed by row block l-2
        if(q.prow >= 1 && q.prow < q.npanels) q.nchain = q.prow;
        if(q.krow >= 0 && q.krow + 2 < q.npanels) q.ntile = (q.npanels - q.krow - 2)*(q.krow + 1);
    }
    if(l < q.npanels)
    {
        q.j0 = l*LCH_NB;
        const int m0 = (n < q.j0 + LCH_NB) ? n : q.j0 + LCH_NB;
        q.has_next = (l + 1 < q.npanels) ? 1 : 0;
        // tiles: the 64-row blocks of the trailing matrix by pairs, without the next diagonal block, and the rhs row against each
        const int nbt = (n - m0 + LCH_NB - 1)/LCH_NB;
        q.incl00 = (!q.has_next && nbt > 0) ? 1 : 0;
        q.ntiles = (nbt > 0) ? nbt*(nbt + 1)/2 - (q.has_next ? 1 : 0) + nbt : 0;
        q.ntrsm  = (l > 0) ? ntrsm_of(l - 1) : 0;
        q.pprev  = (l > 0) ? l - 1 : 0;
    }
    else
    {
        // the last panel's solve: the rhs row alone (and the last row block of Y)
        q.ntrsm = ntrsm_of(q.npanels - 1);
        q.pprev = q.npanels - 1;
    }
    q.jprev = q.pprev*LCH_NB;
    const int work = q.ntiles + q.ntrsm + q.nchain + q.ntile;
    q.nblocks = (q.has_next || work > 0) ? 1 + work : 0;
    return q;
}
// n_dev (optional): the size of the matrix, on the device (<= n_host, which the grids were sized for; NULL: n_host)
__device__ __forceinline__ int lchol_n(const int* __restrict__ n_dev, int n_host)
{
    if(n_dev == NULL) return n_host;
    const int n = *n_dev;
    return (n > 0 && n <= n_host) ? n : n_host;
}
// workgroup bk of launch l (plan q)
__device__ __forceinline__
void lchol_panel_body(int n, int l, const LcholPlan& q, int bk, double* __restrict__ M, double* __restrict__ Linv,
                      int* __restrict__ status, bool with_inverse, double* __restrict__ lds)
{
    static_assert(LCH_LDS_DOUBLES >= 3*LCH_NB*(LCH_NB+1), "the tile workgroups take three 64 x 65 arrays");
    double* __restrict__ MI = lds;
    double* __restrict__ MC = MI + LCH_NB*(LCH_NB+1);
    double* __restrict__ Xs = MC + LCH_NB*(LCH_NB+1);
    // the workspace: [npanels][64][64] inverse diagonal blocks | Yb [npad][npad] | zc [npad]  (of THIS n)
    double* __restrict__ Yb = Linv + (size_t)q.npanels*LCH_NB*LCH_NB;
    double* __restrict__ zc = with_inverse ? Yb + (size_t)q.npad*q.npad : (double*)NULL;
    const double* __restrict__ X     = Linv + (size_t)((l < q.npanels) ? l : 0)*LCH_NB*LCH_NB;
    const double* __restrict__ Xprev = Linv + (size_t)q.pprev*LCH_NB*LCH_NB;
    const int b = bk - 1;
    if(b < 0)
    {
        if(q.has_next) lchol_diag_block(n, M, q.j0 + LCH_NB, Linv + (size_t)(l + 1)*LCH_NB*LCH_NB, status, X, q.j0, lds);
        return;
    }
    // The diagonal workgroup is the long one (24 us against 9), and it starts with a cold read of 96 KB. With
    // two hundred workgroups asking for theirs at the same moment that read took 6 us; the others wait 3 first
#ifndef LCH_NO_SLEEP
    if(q.has_next) __builtin_amdgcn_s_sleep(127);
#endif
    if(b < q.ntiles) lchol_update_tile(n, M, q.j0, X, b, MI, MC, Xs, q.incl00);
    else if(b < q.ntiles + q.ntrsm) lchol_trsm_block(n, M, q.jprev, Xprev, b - q.ntiles, MI, Xs, zc);
    else if(b < q.ntiles + q.ntrsm + q.nchain)
        lchol_inverse_block(n, q.npad, M, Linv, Yb, q.prow, b - q.ntiles - q.ntrsm, q.prow - 1, true, MI, MC, Xs);
    else if(b < q.ntiles + q.ntrsm + q.nchain + q.ntile)
    {
        const int w = b - q.ntiles - q.ntrsm - q.nchain;
        const int nq = q.krow + 1;
        lchol_inverse_block(n, q.npad, M, Linv, Yb, q.krow + 2 + w/nq, w % nq, q.krow, false, MI, MC, Xs);
    }
}
__global__ __launch_bounds__(LCH_THREADS)
void lchol_panel_kernel(const int* __restrict__ n_dev, int n_host, const int* __restrict__ skip, double* __restrict__ M,
                        int l, double* __restrict__ Linv, int* __restrict__ status, int with_inverse)
{
    if(skip != NULL && *skip) return;
    const int n = lchol_n(n_dev, n_host);
    const LcholPlan q = lchol_plan(n, l, with_inverse != 0);
    if((int)blockIdx.x >= q.nblocks) return;
    __shared__ __attribute__((aligned(16))) double lds[LCH_LDS_DOUBLES];
    lchol_panel_body(n, l, q, blockIdx.x, M, Linv, status, with_inverse != 0, lds);
}
// The launches past the ones the host provided one by one, in ONE (round 5): with the size of the matrix decided on
// the device (LcholCompact) the host provides launches for the size it finds likely - the coupled variables of the
// solve's first point and a panel to spare - and this kernel for whatever is left: usually nothing (it returns), else
// panel by panel with a barrier over its workgroups where a launch boundary would be. A step's result does not depend
// on which of the two ways a panel was done. (Nineteen launches for a matrix that needs eleven cost 5 us apiece -
// workgroups of 100 KB of LDS that are dispatched to find out they have nothing to do)
#ifndef LCH_TAIL_WGS
#define LCH_TAIL_WGS 96
#endif
// Returns false if a workgroup never came (the grid is sized so that all of them are resident - launch_cholesky_large -, so
// this is a surprise: a CU mask changed under the process, a debugger): *status = LCH_STATUS_BARRIER_TIMEOUT, which is
// NOT "not positive definite" - lchol_apply_inverse_kernel turns it into SolverCtl::error 4 and the solve fails saying so
//  - and the caller leaves the kernel instead of factoring on unsynchronized data
#define LCH_STATUS_BARRIER_TIMEOUT 0x7fff0003
__device__ __forceinline__ bool lchol_grid_barrier(unsigned* __restrict__ counter, unsigned nwg, unsigned epoch, int* __restrict__ status)
{
    __shared__ int barrier_ok;
    __syncthreads();
    if(threadIdx.x == 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch*nwg;
        int spins = 0, ok = 1;
        while(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        {
            __builtin_amdgcn_s_sleep(8);
            if(++spins > (1 << 23)) { atomicExch(status, LCH_STATUS_BARRIER_TIMEOUT); ok = 0; break; }
            // (somebody else gave up: so do we)
            if((spins & 1023) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == LCH_STATUS_BARRIER_TIMEOUT) { ok = 0; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        barrier_ok = ok;
    }
    __syncthreads();
    return barrier_ok != 0;
}
__global__ __launch_bounds__(LCH_THREADS)
void lchol_tail_kernel(const int* __restrict__ n_dev, int n_host, const int* __restrict__ skip, double* __restrict__ M,
                       int l_first, double* __restrict__ Linv, int* __restrict__ status, int with_inverse, unsigned* __restrict__ counter)
{
    if(skip != NULL && *skip) return;
    const int n = lchol_n(n_dev, n_host);
    const int npanels = (n + LCH_NB - 1)/LCH_NB;
    if(l_first > npanels) return;                       // (every workgroup finds the same)
    __shared__ __attribute__((aligned(16))) double lds[LCH_LDS_DOUBLES];
    unsigned epoch = 0;
    for(int l = l_first; l <= npanels; l++)
    {
        const LcholPlan q = lchol_plan(n, l, with_inverse != 0);
        for(int bk = blockIdx.x; bk < q.nblocks; bk += gridDim.x)
        {
            lchol_panel_body(n, l, q, bk, M, Linv, status, with_inverse != 0, lds);
            __syncthreads();                            // (the LDS is the next block's)
        }
        if(!lchol_grid_barrier(counter, gridDim.x, ++epoch, status)) return;
    }
}

// d = -Y^T z with Y = L^-1 in 64 x 64 blocks (diagonal blocks: X_p in Linv; the others: Yb), z = zc; the result over row n
// of M (which held z). One workgroup of 256 per 16 columns (a row of 16 doubles is one 128-byte line): 16 slices of the
// rows, added in slice order. (A workgroup per 64 columns - 19 of them at 1206 variables, the first one walking 617 KB
// alone - took 16 us; 76 of these take the same 5.8 MB through four times as many CUs)
#define LCH_AI_COLS 16
// Round 5, the splined models: the matrix that was factored may be the camera block WITHOUT its isolated variables (the
// control points no board covers: all they have is their regularization, a 2 x 2 block a control point, coupled to
// nothing: LcholCompact below). Then column c of the factored matrix is camera-block variable perm[c], the solution goes
// to dout[perm[c]], and the workgroups past the columns' solve the 2 x 2 blocks: d = -S2^-1 r2 by the closed form of a
// 2 x 2 Cholesky (not positive definite: status, like a pivot of the big matrix)
struct LcholCompact
{
    const int*    cperm;     // [Nc] position -> camera-block variable | [Nc] variable -> position | [1] n = the coupled ones (they come first); NULL: none of this
    const double* iso;       // [Nc/2][4]: per isolated pair (positions n + 2 q, n + 2 q + 1) s00, s10, s11, then [Nc] their rhs behind all blocks
    double*       dout;      // [Nc] the solution in the camera block's own order
    int           Nc;
    // the dissection (lchol_nd_*; NULL: none): the plan of the point that was reduced (FactorBuffers::ndp_cur). Where it is
    // active the matrix factored here is the SEPARATOR's: its column c is position nA + nB + c of the plan's map
    const int*    ndh;
    // ... and then every workgroup, with its 16 entries of d_S at hand, leaves the sides' share of them behind:
    // ndpart[column block][position i of A | B] = sum over its 16 columns s of L_SX[s][i] d_S[s]  (lchol_nd_apply_kernel adds the
    // blocks' shares in block order: w = z + L_SX^T d_S)
    const double* ndMA; const double* ndMB;
    double*       ndpart;    // [ceil(Nc/16)][2 LCH_ND_WMAX]
};
__global__ __launch_bounds__(256)
void lchol_apply_inverse_kernel(const int* __restrict__ n_dev, int n_host, const int* __restrict__ skip, double* __restrict__ M,
                                const double* __restrict__ Linv, int with_post, Step2Dev sd, int* __restrict__ chol_status,
                                LcholCompact cp)
{
    if(skip != NULL && *skip) return;
    const int n = lchol_n(n_dev, n_host);
    constexpr int NB = LCH_NB;
    const int npanels = (n + NB - 1)/NB, npad = npanels*NB;
    const double* __restrict__ Yb = Linv + (size_t)npanels*NB*NB;
    const double* __restrict__ zc = Yb + (size_t)npad*npad;
    const int ncolblocks = (n + LCH_AI_COLS - 1)/LCH_AI_COLS;
    const bool ndact = cp.ndh != NULL && cp.ndh[NDH_ACTIVE] != 0;
    const int* __restrict__ ndmap = ndact ? cp.ndh + NDH_WORDS + cp.Nc + cp.ndh[NDH_NA] + cp.ndh[NDH_NB] : (const int*)NULL;
    if((int)blockIdx.x >= ncolblocks)
    {
        // the isolated pairs, a thread each (their positions: behind ALL the coupled variables, cperm's count)
        if(cp.cperm == NULL) return;
        const int pair = ((int)blockIdx.x - ncolblocks)*blockDim.x + threadIdx.x;
        const int p0 = cp.cperm[2*cp.Nc] + 2*pair;
        if(p0 >= cp.Nc) return;
        const double* __restrict__ b = cp.iso + (size_t)4*pair;
        const double* __restrict__ rr = cp.iso + (size_t)4*(cp.Nc/2 + 1) + 2*pair;
        const bool two = p0 + 1 < cp.Nc;
        const double s00 = b[0], s10 = two ? b[1] : 0.0, s11 = two ? b[2] : 1.0;
        const double r0 = rr[0], r1 = two ? rr[1] : 0.0;
        (void)n;
        // L = [l00 0; l10 l11]
        bool bad = !(s00 > 0.0);
        const double l00 = sqrt(bad ? 1.0 : s00), l10 = s10/l00;
        const double t11 = s11 - l10*l10;
        bad = bad || !(t11 > 0.0);
        const double l11 = sqrt(bad ? 1.0 : t11);
        const double z0 = r0/l00, z1 = (r1 - l10*z0)/l11;
        const double x1 = z1/l11, x0 = (z0 - l10*x1)/l00;
        cp.dout[cp.cperm[p0]] = -x0;
        if(two) cp.dout[cp.cperm[p0 + 1]] = -x1;
        // (a pair that is not positive definite was reported by the factorization's first launch: lchol_diag_kernel)
        (void)bad;
        return;
    }
    // with_post (round 5): what step2_post_kernel did in a launch of its own - the factorization's verdict into the
    // control block (every panel's diagonal workgroup has run: the status is final) - by one thread of this launch
    if(with_post && blockIdx.x == 0 && threadIdx.x == 0)
    {
        if(*chol_status == LCH_STATUS_BARRIER_TIMEOUT) { sd.ctl->error = 4; sd.ctl->done = 1; }      // (not a verdict on the matrix)
        step2_chol_done(sd, *chol_status != 0);
    }
    __shared__ double part[16][LCH_AI_COLS];
    const int t = threadIdx.x, j16 = t & (LCH_AI_COLS - 1), slice = t >> 4;
    const int c = blockIdx.x*LCH_AI_COLS + j16;
    const int q = c / NB, j = c - q*NB;
    double a0 = 0.0, a1 = 0.0;
    if(c < n)
    {
        // the diagonal block: Y[i][c] = X_q[i - 64 q][j], i >= c
        const double* __restrict__ X = Linv + (size_t)q*NB*NB;
        for(int i = j + slice; i < NB && q*NB + i < n; i += 16)
            a0 = fma(X[(size_t)i*NB + j], zc[q*NB + i], a0);
        const double* __restrict__ col = Yb + c;
        int i = (q + 1)*NB + slice;
        constexpr int UB = 8;
        for(; i + 16*(UB-1) < n; i += 16*UB)
        {
            double v[UB], z[UB];
#pragma unroll
            for(int u = 0; u < UB; u++) { v[u] = col[(size_t)(i + 16*u)*npad]; z[u] = zc[i + 16*u]; }
#pragma unroll
            for(int u = 0; u < UB; u += 2) { a0 = fma(v[u], z[u], a0); a1 = fma(v[u+1], z[u+1], a1); }
        }
        for(; i < n; i += 16) a0 = fma(col[(size_t)i*npad], zc[i], a0);
    }
    part[slice][j16] = a0 + a1;
    __syncthreads();
    if(t < LCH_AI_COLS && c < n)
    {
        double sacc = 0.0;
        for(int k = 0; k < 16; k++) sacc += part[k][t];
        if(ndact)                 cp.dout[ndmap[c]] = -sacc;
        else if(cp.cperm != NULL) cp.dout[cp.cperm[c]] = -sacc;
        else                      M[(size_t)n*n + c] = -sacc;
        part[0][t] = -sacc;
    }
    if(ndact && cp.ndpart != NULL)
    {
        if(t < LCH_AI_COLS && c >= n) part[0][t] = 0.0;
        __syncthreads();
        const int nA = cp.ndh[NDH_NA], nB = cp.ndh[NDH_NB], s0 = blockIdx.x*LCH_AI_COLS;
        for(int i = t; i < nA + nB; i += blockDim.x)
        {
            const bool inA = i < nA;
            const int nx = inA ? nA : nB, ii = inA ? i : i - nA, N = nx + n;
            const double* __restrict__ colp = (inA ? cp.ndMA : cp.ndMB) + (size_t)(nx + s0)*N + ii;
            double m[LCH_AI_COLS];
#pragma unroll
            for(int k = 0; k < LCH_AI_COLS; k++) m[k] = colp[(size_t)((s0 + k < n) ? k : 0)*N];
            double acc = 0.0;
#pragma unroll
            for(int k = 0; k < LCH_AI_COLS; k++) acc = fma(m[k], part[0][k], acc);
            cp.ndpart[(size_t)blockIdx.x*(2*LCH_ND_WMAX) + i] = acc;
        }
    }
}

////////////////////////////////////////////////////////////////////////////////
// Round 5: a nested-dissection order of the splined models' camera block.
// With the frames eliminated every board couples ALL control points of the box under it; a strip of grid columns as wide
// as the widest box less one is a SEPARATOR: every box lies in the strip and ONE side of it. The coupled variables in
// the order [side A | side B | separator S (and whatever is no control point)] have S_BA = 0, so A's panels and B's
// panels are factored side by side - two chains in the same launches instead of one after the other:
//   M_A = [ S_AA      ;     M_B likewise;     M_S = [ S_SS ; r_S ]
//           S_SA  0   ;
//           r_A   0 ]     (nA + nS + 1 rows, nA + nS columns: the last nS rows and columns are the BORDER, zero at first)
// * lchol_nd_first_kernel: the two first diagonal blocks.  * lchol_nd_pair_kernel, launch l = 0 .. R-1: launch l of the
//   standard factorization (lchol_panel_body) of each chain, which stops behind its own panels (LcholPlan::own): the
//   trailing updates reach into the chain's border, the panel solves give L_SA (L_SB) and z_A (z_B), L_AA^-1 (L_BB^-1)
//   is made on the side.  * lchol_nd_junction_kernel: the separator's first diagonal block - what the reduction left of it
//   plus the two borders' - factored; the other tiles of the borders added to M_S; the chains' closing launches.
//   * then M_S is a matrix like any other: lchol_panel_kernel / lchol_tail_kernel / lchol_apply_inverse_kernel (d_S).
//   * lchol_nd_apply_kernel: d_A = -Y_A^T (z_A + L_SA^T d_S), likewise B.
// Sizes are the device's (spl_compact_body plans after every evaluation: the boxes move with the state); nA, nB are
// padded to whole panels with identity rows (positions without a variable). The host provides R rounds and grids for a
// border of NSprov (what the solve's first point needs, NdLimits); a plan that does not fit is not used (everything is
// "separator": the launches of the chains find nothing to do).
////////////////////////////////////////////////////////////////////////////////
struct LcholChain
{
    double*    M;        // [(nx + ns + 1)][nx + ns]
    double*    Linv;     // [own][64][64] | Yb [64 own][64 own] | zc [64 own],  own = nx/64
    const int* nx_dev;   // the chain's own columns (a multiple of 64)
    const int* ns_dev;   // its border
};
__global__ __launch_bounds__(LCH_THREADS)
void lchol_nd_first_kernel(LcholChain A, LcholChain B, const int* __restrict__ ndh, const int* __restrict__ skip, int* __restrict__ status)
{
    if(skip != NULL && *skip) return;
    if(!ndh[NDH_ACTIVE]) return;
    __shared__ __attribute__((aligned(16))) double lds[LCH_LDS_DOUBLES];
    const LcholChain& C = (blockIdx.x == 0) ? A : B;
    const int nx = *C.nx_dev, ns = *C.ns_dev;
    if(nx < LCH_NB) return;
    lchol_diag_block(nx + ns, C.M, 0, C.Linv, status, NULL, 0, lds);
}
__global__ __launch_bounds__(LCH_THREADS)
void lchol_nd_pair_kernel(LcholChain A, LcholChain B, const int* __restrict__ ndh, const int* __restrict__ skip, int l, int* __restrict__ status)
{
    if(skip != NULL && *skip) return;
    if(!ndh[NDH_ACTIVE]) return;
    const int nA = *A.nx_dev, nB = *B.nx_dev, nS = *A.ns_dev;
    LcholPlan qA = lchol_plan(nA + nS, l, true, nA/LCH_NB), qB = lchol_plan(nB + nS, l, true, nB/LCH_NB);
    if(nA < LCH_NB) qA.nblocks = 0;
    if(nB < LCH_NB) qB.nblocks = 0;
    __shared__ __attribute__((aligned(16))) double lds[LCH_LDS_DOUBLES];
    // the two long workgroups (the chains' next diagonal blocks) first, then A's others, then B's
    int bk = blockIdx.x;
    if(bk == 0) { if(qA.nblocks > 0) lchol_panel_body(nA + nS, l, qA, 0, A.M, A.Linv, status, true, lds); return; }
    if(bk == 1) { if(qB.nblocks > 0) lchol_panel_body(nB + nS, l, qB, 0, B.M, B.Linv, status, true, lds); return; }
    bk -= 2;
    const int ra = (qA.nblocks > 1) ? qA.nblocks - 1 : 0, rb = (qB.nblocks > 1) ? qB.nblocks - 1 : 0;
    if(bk < ra)           lchol_panel_body(nA + nS, l, qA, bk + 1,      A.M, A.Linv, status, true, lds);
    else if(bk - ra < rb) lchol_panel_body(nB + nS, l, qB, bk - ra + 1, B.M, B.Linv, status, true, lds);
}
// workgroups of lchol_nd_junction_kernel: [0] the separator's first block | the merge of the other tiles of the two
// borders into M_S (tile q as lch_tile_of(q, nbt = blocks of nS, incl00) numbers them, (0,0) left out; the rhs row's tiles last)
// | launch l_close of both chains (their closing launch if they have l_close panels)
__host__ __device__ inline int lchol_nd_merge_tiles(int nS) { const int nb = (nS + LCH_NB - 1)/LCH_NB; return nb*(nb + 1)/2 - 1 + nb; }
__global__ __launch_bounds__(LCH_THREADS)
void lchol_nd_junction_kernel(LcholChain A, LcholChain B, const int* __restrict__ ndh, const int* __restrict__ skip,
                              double* __restrict__ MS, double* __restrict__ LinvS, int n_host, int l_close, int nmerge_host,
                              int* __restrict__ status, const int* __restrict__ n1_dev, const double* __restrict__ iso, int iso_Nc,
                              unsigned* __restrict__ tail_counter)
{
    if(tail_counter != NULL && blockIdx.x == 0 && threadIdx.x == 0) *tail_counter = 0u;
    if(skip != NULL && *skip) return;
    __shared__ __attribute__((aligned(16))) double lds[LCH_LDS_DOUBLES];
    const int active = ndh[NDH_ACTIVE];
    const int nS = lchol_n(ndh + NDH_NSEFF, n_host);
    const int nA = active ? *A.nx_dev : 0, nB = active ? *B.nx_dev : 0;
    const int NA = nA + nS, NB = nB + nS;
    if(blockIdx.x == 0)
    {
        // (the isolated pairs: as in lchol_diag_kernel)
        if(iso != NULL)
        {
            const int n1 = *n1_dev;
            for(int p0 = n1 + 2*(int)threadIdx.x; p0 < iso_Nc; p0 += 2*LCH_THREADS)
            {
                const double* __restrict__ b = iso + (size_t)4*((p0 - n1) >> 1);
                const bool two = p0 + 1 < iso_Nc;
                const double s00 = b[0], s10 = two ? b[1] : 0.0, s11 = two ? b[2] : 1.0;
                if(!(s00 > 0.0) || !(s11 - s10*(s10/s00) > 0.0)) atomicExch(status, 1);
            }
        }
        if(active) lchol_diag_block(nS, MS, 0, LinvS, status, NULL, 0, lds,
                                    A.M + (size_t)nA*NA + nA, NA, B.M + (size_t)nB*NB + nB, NB);
        else       lchol_diag_block(nS, MS, 0, LinvS, status, NULL, 0, lds);
        return;
    }
    if(!active) return;
    // (the first workgroup is the long one and starts with a cold read: the others wait, as in lchol_panel_body)
#ifndef LCH_NO_SLEEP
    __builtin_amdgcn_s_sleep(127);
#endif
    int bk = (int)blockIdx.x - 1;
    if(bk < nmerge_host)
    {
        const int nbt = (nS + LCH_NB - 1)/LCH_NB;
        if(bk >= lchol_nd_merge_tiles(nS)) return;
        int bi, bj;
        lch_tile_of(bk, nbt, &bi, &bj);
        const bool rhs = (bi == nbt), diag = (bi == bj);
        const int i0 = rhs ? nS : LCH_NB*bi, c0 = LCH_NB*bj;
        const int ni = rhs ? 1 : min(LCH_NB, nS - i0), nc = min(LCH_NB, nS - c0);
        for(int idx = threadIdx.x; idx < ni*LCH_NB; idx += LCH_THREADS)
        {
            const int i = idx / LCH_NB, c = idx - i*LCH_NB;
            if(c >= nc || (diag && c > i)) continue;
            double* __restrict__ dst = &MS[(size_t)(i0 + i)*nS + c0 + c];
            *dst = (*dst + A.M[(size_t)(nA + i0 + i)*NA + nA + c0 + c]) + B.M[(size_t)(nB + i0 + i)*NB + nB + c0 + c];
        }
        return;
    }
    bk -= nmerge_host;
    LcholPlan qA = lchol_plan(NA, l_close, true, nA/LCH_NB), qB = lchol_plan(NB, l_close, true, nB/LCH_NB);
    if(nA < LCH_NB) qA.nblocks = 0;
    if(nB < LCH_NB) qB.nblocks = 0;
    const int ra = (qA.nblocks > 1) ? qA.nblocks - 1 : 0, rb = (qB.nblocks > 1) ? qB.nblocks - 1 : 0;
    if(bk < ra)           lchol_panel_body(NA, l_close, qA, bk + 1,      A.M, A.Linv, status, true, lds);
    else if(bk - ra < rb) lchol_panel_body(NB, l_close, qB, bk - ra + 1, B.M, B.Linv, status, true, lds);
}
// d_X = -Y_X^T (z_X + L_SX^T d_S) for X = A (workgroups [0, ncb_host)) and B (the others): 16 columns a workgroup as in
// lchol_apply_inverse_kernel, which has left L_SX^T d_S behind in shares of 16 entries of d_S (LcholCompact::ndpart): w = z +
// the shares in block order, made by every workgroup for itself in LDS. (w from the border rows themselves, here: every
// workgroup half a megabyte from the L2 - 41 us a thread a row, 17 us by 1024 threads with sixteen loads in flight)
__global__ __launch_bounds__(256)
void lchol_nd_apply_kernel(LcholChain A, LcholChain B, const int* __restrict__ ndh, const int* __restrict__ skip, int ncb_host,
                           const int* __restrict__ nperm, double* __restrict__ dout, const double* __restrict__ ndpart)
{
    if(skip != NULL && *skip) return;
    if(!ndh[NDH_ACTIVE]) return;
    const bool isA = (int)blockIdx.x < ncb_host;
    const LcholChain& C = isA ? A : B;
    const int cb = isA ? (int)blockIdx.x : (int)blockIdx.x - ncb_host;
    const int nx = *C.nx_dev, nS = *C.ns_dev;
    if(nx < LCH_NB || nx > LCH_ND_WMAX || cb*LCH_AI_COLS >= nx) return;
    const int nA = *A.nx_dev;
    const int pos0 = isA ? 0 : nA;
    constexpr int NB = LCH_NB;
    const int own = nx/NB, npad = nx;
    const double* __restrict__ Yb = C.Linv + (size_t)own*NB*NB;
    const double* __restrict__ zc = Yb + (size_t)npad*npad;
    __shared__ double w[LCH_ND_WMAX];
    __shared__ double part[16][LCH_AI_COLS];
    const int t = threadIdx.x;
    // rows >= this workgroup's first column alone are read below
    const int c_first = cb*LCH_AI_COLS;
    const int ncbS = (nS + LCH_AI_COLS - 1)/LCH_AI_COLS;
    for(int i = c_first + t; i < nx; i += 256)
    {
        const double* __restrict__ pp = ndpart + pos0 + i;
        double acc = 0.0;
        int b = 0;
        constexpr int U = 8;
        for(; b + U <= ncbS; b += U)
        {
            double m[U];
#pragma unroll
            for(int u = 0; u < U; u++) m[u] = pp[(size_t)(b + u)*(2*LCH_ND_WMAX)];
#pragma unroll
            for(int u = 0; u < U; u++) acc += m[u];
        }
        for(; b < ncbS; b++) acc += pp[(size_t)b*(2*LCH_ND_WMAX)];
        w[i] = zc[i] + acc;
    }
    __syncthreads();
    const int j16 = t & (LCH_AI_COLS - 1), slice = t >> 4;
    const int c = c_first + j16;
    const int q = c / NB, j = c - q*NB;
    double a0 = 0.0, a1 = 0.0;
    if(c < nx)
    {
        const double* __restrict__ X = C.Linv + (size_t)q*NB*NB;
        for(int i = j + slice; i < NB; i += 16) a0 = fma(X[(size_t)i*NB + j], w[q*NB + i], a0);
        const double* __restrict__ col = Yb + c;
        for(int i = (q + 1)*NB + slice; i < nx; i += 16) a1 = fma(col[(size_t)i*npad], w[i], a1);
    }
    part[slice][j16] = a0 + a1;
    __syncthreads();
    if(t < LCH_AI_COLS && c < nx)
    {
        double sacc = 0.0;
        for(int k = 0; k < 16; k++) sacc += part[k][t];
        const int v = nperm[pos0 + c];
        if(v >= 0) dout[v] = -sacc;
    }
}

// L^T d = z, panel by panel from the last: one workgroup. z is row n of M; on
// return r (= that row) holds -d.
// One workgroup pulls ~50 GB/s, and the factor of a 1206-variable block is 5.8 MB: 115-155 us. So the panels are
// taken in a few GROUPS, last group first: this kernel solves a group's panels [p_lo, p_hi) against the group's
// own rows (rows below the group are solved already and have been applied), then lchol_backward_apply_kernel - as
// many workgroups as there are 64-column blocks left of the group - subtracts the group's rows times its d from
// every earlier entry of z. The one-workgroup kernels stream the groups' diagonal triangles only (1/16 of the
// factor each with four groups). The last one (p_lo == 0) negates
__global__ __launch_bounds__(1024)
void lchol_backward_kernel(int n, const int* __restrict__ skip, double* __restrict__ M,
                           const double* __restrict__ Linv_all, int p_lo, int p_hi)
{
    if(skip != NULL && *skip) return;
    extern __shared__ double zs[];                      // z, n doubles: read by every thread in every panel
    double* __restrict__ z = M + (size_t)n*n;
    const int row_hi = min(n, p_hi*LCH_NB);             // rows of this group: [p_lo*64, row_hi)
    __shared__ double part[16][LCH_NB];
    __shared__ double w[LCH_NB];
    __shared__ double Xs[LCH_NB*LCH_NB];
    const int t = threadIdx.x;
    const int c = t & (LCH_NB-1), slice = t >> 6;       // 16 slices of rows for each of the 64 columns
    for(int i = t; i < n; i += blockDim.x) zs[i] = z[i];
    __syncthreads();
    for(int p = p_hi-1; p >= p_lo; p--)
    {
        const int j0 = p*LCH_NB;
        const int nb = min(LCH_NB, n - j0);
        const int m0 = j0 + nb;
        // w[c] = z[j0+c] - sum_{i >= m0} L[i][j0+c] d[i].  One workgroup streams the
        // block column: 24 loads in flight per thread (one at a time, a panel's
        // column of 1100 rows is 70 dependent memory round trips: this kernel took 318 us)
        // (this panel's inverse diagonal block into LDS on the way: read from memory
        //  inside the 64-step product below it was 64 dependent round trips per panel)
        {
            const double* __restrict__ X = Linv_all + (size_t)p*LCH_NB*LCH_NB;
#pragma unroll
            for(int u = 0; u < LCH_NB*LCH_NB/1024; u++) Xs[t + 1024*u] = X[t + 1024*u];
        }
        double acc = 0.0;
        if(c < nb)
        {
            const double* __restrict__ col = M + j0 + c;
            int i = m0 + slice;
            constexpr int UB = 24;
            for(; i + 16*(UB-1) < row_hi; i += 16*UB)
            {
                double v[UB];
#pragma unroll
                for(int u = 0; u < UB; u++) v[u] = col[(size_t)(i + 16*u)*n];
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for(int u = 0; u < UB; u += 2) { a0 += v[u]*zs[i + 16*u]; a1 += v[u+1]*zs[i + 16*(u+1)]; }
                acc += a0 + a1;
            }
            for(; i + 16*3 < row_hi; i += 16*4)
            {
                double v[4];
#pragma unroll
                for(int u = 0; u < 4; u++) v[u] = col[(size_t)(i + 16*u)*n];
#pragma unroll
                for(int u = 0; u < 4; u++) acc += v[u]*zs[i + 16*u];
            }
            for(; i < row_hi; i += 16) acc += col[(size_t)i*n]*zs[i];
        }
        part[slice][c] = acc;
        __syncthreads();
        if(t < LCH_NB)
        {
            double sacc = 0.0;
            for(int k = 0; k < 16; k++) sacc += part[k][t];
            w[t] = (t < nb) ? zs[j0 + t] - sacc : 0.0;
        }
        __syncthreads();
        // d_p = L11^-T w:  d[c] = sum_{k >= c} Linv[k][c] w[k]   (Xs: staged below, under the streaming)
        if(t < nb)
        {
            double sacc = 0.0;
            for(int k = t; k < nb; k++) sacc += Xs[k*LCH_NB + t]*w[k];
            zs[j0 + t] = sacc;
        }
        __syncthreads();
    }
    if(p_lo == 0) { for(int i = t; i < n; i += blockDim.x) z[i] = -zs[i]; }
    else          { for(int i = p_lo*LCH_NB + t; i < row_hi; i += blockDim.x) z[i] = zs[i]; }
}
// z[c] -= sum over the rows i in [row_lo, row_hi) of L[i][c] d[i] (d = z there), c < row_lo: 64 columns per workgroup,
// four slices of the rows, added in order
__global__ __launch_bounds__(256)
void lchol_backward_apply_kernel(int n, const int* __restrict__ skip, double* __restrict__ M, int row_lo, int row_hi)
{
    if(skip != NULL && *skip) return;
    double* __restrict__ z = M + (size_t)n*n;
    __shared__ double part[4][LCH_NB];
    const int t = threadIdx.x, c = blockIdx.x*LCH_NB + (t & (LCH_NB-1)), slice = t >> 6;
    double a0 = 0.0, a1 = 0.0;
    if(c < row_lo)
    {
        const double* __restrict__ col = M + c;
        int i = row_lo + slice;
        constexpr int UB = 16;
        for(; i + 4*(UB-1) < row_hi; i += 4*UB)
        {
            double v[UB], d[UB];
#pragma unroll
            for(int u = 0; u < UB; u++) { v[u] = col[(size_t)(i + 4*u)*n]; d[u] = z[i + 4*u]; }
#pragma unroll
            for(int u = 0; u < UB; u += 2) { a0 = fma(v[u], d[u], a0); a1 = fma(v[u+1], d[u+1], a1); }
        }
        for(; i < row_hi; i += 4) a0 = fma(col[(size_t)i*n], z[i], a0);
    }
    part[slice][t & (LCH_NB-1)] = a0 + a1;
    __syncthreads();
    if(t < LCH_NB && c < row_lo) z[c] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

} // namespace mrcal_amd
// dev / tests (no declaration in include/: not part of the interface): what launch l of the launch-per-panel factorization
// of an n x n matrix consists of (lchol_plan(), the function the host sizes its grids with and the kernels find their
// role by). out[10]: npanels, has_next, incl00, ntiles, ntrsm, nchain, ntile, nblocks, pprev, npad. Needs no GPU
extern "C" void mrcal_amd_debug_lchol_plan(int n, int l, int with_inverse, int own, int* out)
{
    const mrcal_amd::LcholPlan q = mrcal_amd::lchol_plan(n, l, with_inverse != 0, own);
    out[0] = q.npanels; out[1] = q.has_next; out[2] = q.incl00; out[3] = q.ntiles; out[4] = q.ntrsm;
    out[5] = q.nchain;  out[6] = q.ntile;    out[7] = q.nblocks; out[8] = q.pprev; out[9] = q.npad;
}
namespace mrcal_amd {
// the workspace behind FactorBuffers::Linv: [npanels][64][64] inverse diagonal blocks | Yb [npad][npad] | zc [npad]
static inline size_t lchol_npad(int n) { return (size_t)((n + LCH_NB - 1)/LCH_NB)*LCH_NB; }
// sd (optional): the trial step this factorization belongs to - its end-of-trial logic rides in the first launch and
// the verdict in the last (no step2_finish_kernel / step2_post_kernel around the call). *fused says whether that happened
// (not with the backward sweep of MRCAL_AMD_LCHOL_SWEEP, whose last launch is another).
// n_dev (optional, round 5): the size of the matrix as the DEVICE knows it, <= n (LcholCompact: the camera block without
// its isolated variables, whose number follows the boards). The launches and their grids are those of n; a launch past
// the device's last panel finds nothing to do
// nds (round 5): the dissection's chains in front (lchol_nd_*): M is then the separator's matrix, the first launches are
// lchol_nd_first_kernel, R of lchol_nd_pair_kernel and lchol_nd_junction_kernel (in lchol_diag_kernel's place), the last
// lchol_nd_apply_kernel; the end-of-trial logic has run already (it rides in the reduction: step2_reduce_kernel), the
// verdict rides in lchol_apply_inverse_kernel as ever
struct LcholNdLaunch { LcholChain A, B; const int* ndh; NdLimits lim; };
// lchol_tail_kernel's workgroups wait for each other (lchol_grid_barrier): every one of them must be RESIDENT at once.
// LCH_TAIL_WGS of them (any number gives the same bits: they share the blocks of a panel round-robin), but never more than
// the device holds of this kernel - a partitioned or CU-masked GPU with fewer than 96 free CUs would otherwise leave the
// resident ones spinning for the ones queued behind them. Asked once per process
static int lchol_tail_grid()
{
    static const int grid = []
    {
        int dev = 0, ncu = 0, per_cu = 0;
        if(hipGetDevice(&dev) != hipSuccess) return 1;
        if(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return 1;
        if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lchol_tail_kernel, LCH_THREADS, 0) != hipSuccess || per_cu <= 0) return 1;
        // (the occupancy query is an upper bound where the hardware admits one block fewer - MI355X_MICROARCH.md,
        //  "Residency and cooperative launch": one block per CU at most is always safe, and 100 KB of LDS allow no more)
        const long long resident = (long long)ncu*(per_cu > 1 ? 1 : per_cu);
        return (int)std::max(1LL, std::min<long long>(LCH_TAIL_WGS, resident));
    }();
    return grid;
}
hipError_t launch_cholesky_large(int n, const int* skip, double* M, double* Linv, int* status, hipStream_t stream,
                                 const Step2Dev* sd = NULL, bool* fused = NULL, const int* n_dev = NULL, const LcholCompact* compact = NULL,
                                 int likely_panels = 0 /* with n_dev: launches 0 .. likely_panels one by one, the rest in lchol_tail_kernel; 0: all one by one */,
                                 unsigned* tail_counter = NULL, const LcholNdLaunch* nds = NULL,
                                 bool finish_done = false /* with sd: the end-of-trial logic has run already (the first launch goes by `skip`); the verdict still rides in the last */,
                                 bool sweep = false /* the solve by the backward sweep in groups of panels (rounds 2-3; backward stable) instead of through
                                                       L^-1 built on the side (lchol_inverse_block): FactorBuffers::use_sweep */)
{
    const int npanels = (n + LCH_NB - 1)/LCH_NB;
    if(sweep && (n_dev != NULL || compact != NULL)) return hipErrorInvalidValue;
    Step2Dev sd0; memset(&sd0, 0, sizeof(sd0));
    LcholCompact cp0; memset(&cp0, 0, sizeof(cp0));
    const bool fuse = (sd != NULL && !sweep);
    if(fused != NULL) *fused = fuse;
    if(nds != NULL)
    {
        if(sweep || !fuse || !finish_done || n_dev == NULL || compact == NULL) return hipErrorInvalidValue;
        const int R = nds->lim.rounds, Nprov = LCH_NB*R + nds->lim.ns_max;
        hipLaunchKernelGGL(lchol_nd_first_kernel, dim3(2), dim3(LCH_THREADS), 0, stream, nds->A, nds->B, nds->ndh, skip, status);
        for(int l = 0; l < R; l++)
        {
            // (a chain with fewer panels, or a smaller border, has no more workgroups in launch l than this one: lchol_plan()
            //  grows with n and own term by term, and a closing launch is no larger than the panel's launch in its place)
            const LcholPlan q = lchol_plan(Nprov, l, true, R);
            hipLaunchKernelGGL(lchol_nd_pair_kernel, dim3(2 + 2*std::max(q.nblocks - 1, 0)), dim3(LCH_THREADS), 0, stream,
                               nds->A, nds->B, nds->ndh, skip, l, status);
        }
        const LcholPlan qc = lchol_plan(Nprov, R, true, R);
        const int nmerge = lchol_nd_merge_tiles(nds->lim.ns_max);
        hipLaunchKernelGGL(lchol_nd_junction_kernel, dim3(1 + nmerge + 2*std::max(qc.nblocks - 1, 0)), dim3(LCH_THREADS), 0, stream,
                           nds->A, nds->B, nds->ndh, skip, M, Linv, n, R, nmerge, status,
                           compact->cperm + 2*compact->Nc, compact->iso, compact->Nc, tail_counter);
    }
    else
    hipLaunchKernelGGL(lchol_diag_kernel, dim3(1), dim3(LCH_THREADS), 0, stream, n_dev, n, skip, M, 0, Linv, status,
                       (fuse && !finish_done) ? 1 : 0, fuse ? *sd : sd0, compact ? compact->iso : (const double*)NULL, compact ? compact->Nc : 0,
                       (n_dev != NULL) ? tail_counter : (unsigned*)NULL);
    const bool with_tail = n_dev != NULL && tail_counter != NULL && likely_panels > 0 && likely_panels < npanels && !sweep;
    const int  l_last = with_tail ? likely_panels : npanels;
    for(int l = 0; l <= l_last; l++)
    {
        const LcholPlan q = lchol_plan(n, l, !sweep);
        // (with a size the device decides, a launch that is a panel's at n may be the closing one there, or nothing: the
        //  plan of n has the workgroups for either - lchol_plan() grows with n term by term)
        const int nblocks = q.nblocks;
        if(nblocks <= 0) continue;
        hipLaunchKernelGGL(lchol_panel_kernel, dim3(nblocks), dim3(LCH_THREADS), 0, stream,
                           n_dev, n, skip, M, l, Linv, status, sweep ? 0 : 1);
    }
    if(with_tail)
        hipLaunchKernelGGL(lchol_tail_kernel, dim3(lchol_tail_grid()), dim3(LCH_THREADS), 0, stream,
                           n_dev, n, skip, M, l_last + 1, Linv, status, 1, tail_counter);
    if(!sweep)
    {
        const int niso_blocks = (compact != NULL) ? (n/2 + 1 + 255)/256 : 0;
        hipLaunchKernelGGL(lchol_apply_inverse_kernel, dim3((n + LCH_AI_COLS - 1)/LCH_AI_COLS + niso_blocks), dim3(256), 0, stream,
                           n_dev, n, skip, M, (const double*)Linv, fuse ? 1 : 0, fuse ? *sd : sd0, status, compact ? *compact : cp0);
        if(nds != NULL)
        {
            const int ncb = LCH_NB*nds->lim.rounds/LCH_AI_COLS;
            hipLaunchKernelGGL(lchol_nd_apply_kernel, dim3(2*ncb), dim3(256), 0, stream, nds->A, nds->B, nds->ndh, skip, ncb,
                               nds->ndh + NDH_WORDS + compact->Nc, compact->dout, (const double*)compact->ndpart);
        }
        return hipGetLastError();
    }
    // the backward sweep, in groups of panels (see lchol_backward_kernel)
    const int ngroups = (npanels >= 12) ? 4 : (npanels >= 6) ? 2 : 1;
    for(int g = ngroups - 1; g >= 0; g--)
    {
        const int p_lo = (int)((long long)npanels*g/ngroups), p_hi = (int)((long long)npanels*(g + 1)/ngroups);
        hipLaunchKernelGGL(lchol_backward_kernel, dim3(1), dim3(1024), (size_t)n*sizeof(double), stream, n, skip, M, Linv, p_lo, p_hi);
        if(p_lo > 0)
            hipLaunchKernelGGL(lchol_backward_apply_kernel, dim3(p_lo), dim3(256), 0, stream,
                               n, skip, M, p_lo*LCH_NB, std::min(n, p_hi*LCH_NB));
    }
    return hipGetLastError();
}
size_t cholesky_large_workspace_doubles(int n)
{
    if(chol_fits_lds(n)) return 1;       // the LDS kernel serves
    const size_t npad = lchol_npad(n);
    return (size_t)((n + LCH_NB - 1)/LCH_NB)*LCH_NB*LCH_NB + npad*npad + npad;
}

// d_e = -L^-T (y_e + Wt_e d_s);  also scatters d_s into the state-ordered step
__global__ __launch_bounds__(64)
void backsub_kernel(NormalDims nd, BlockRanges br, OpRef R, const int* __restrict__ skip_also,
                    const double* __restrict__ Wt, const double* __restrict__ LD,
                    const double* __restrict__ y, const double* __restrict__ ds)
{
    if(opref_skip(R)) return;
    if(skip_also != NULL && *skip_also) return;
    double* __restrict__ step = opref_get(R).step_gn;
    const int t   = threadIdx.x;
    if((int)blockIdx.x == br.count())
    {
        // the extra block copies d_s
        for(int i=t;i<nd.Nc;i+=blockDim.x)
            step[S_to_state(nd, i)] = ds[i];
        return;
    }
    const int blk = br.block(blockIdx.x);
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    // all the loads first: L, y, and this lane's columns of Wt_e against d_s
    const double Lv = (t < 36) ? LD[(size_t)blk*36 + t] : 0.0;
    const double yv = (t < de) ? y[e0 + t] : 0.0;
    double part[6] = {0,0,0,0,0,0};
    for(int c=t;c<nd.Nc;c+=blockDim.x)
    {
        const double d = ds[c];
#pragma unroll
        for(int i=0;i<6;i++) if(i < de) part[i] += Wt[(size_t)(e0+i)*nd.Nc + c]*d;
    }
#pragma unroll
    for(int i=0;i<6;i++)
        for(int off=32; off>0; off>>=1) part[i] += __shfl_down(part[i], off);
    __shared__ double Ls[36];
    if(t < 36) Ls[t] = Lv;
    __syncthreads();
    // lane 0 holds the sums; y comes from the lanes that loaded it
    double v[6];
#pragma unroll
    for(int i=0;i<6;i++) v[i] = __shfl(yv, i) + part[i];
    if(t == 0)
    {
        for(int i=de-1;i>=0;i--)
        {
            double s = v[i];
            for(int k=i+1;k<de;k++) s -= Ls[k*6+i]*v[k];
            v[i] = s/Ls[i*6+i];
        }
        for(int i=0;i<de;i++) step[nd.E_state0 + e0 + i] = -v[i];
    }
}

////////////////////////////////////////////////////////////////////////////////
// v^T N v = |J v|^2 from the blocks;  dot products
////////////////////////////////////////////////////////////////////////////////
// out[0] += v^T N v, and if nout == 3: out[1] += g . v,  out[2] += v . v   with N = [A B; Bt D]
// of the operating point: v^T N v = v_S^T A v_S + 2 v_E^T (Bt v_S) + v_E^T D v_E.
// One wave per group of rows of [A ; Bt]; one atomic triple per workgroup
// (QF_ROWS_PER_WAVE: solver_kernels.hpp. With 8 rows a wave, a 1206-variable camera block had 188 workgroups
//  walking 46 MB of Bt: 1.4 TB/s)
// this workgroup's (256 threads) part of (v^T N v, g.v, v.v): returned in threads 0, 1, 2
__device__ __forceinline__
double quadform_body(const NormalDims& nd, const OpDev& O, const double* __restrict__ v, int block, bool vv_E_only = false,
                     const unsigned* __restrict__ occ = NULL /* eblock_factor_kernel's bit per (block, 16-column tile) of Wt - and of Bt: the same columns */,
                     int nocc = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nrows = nd.Nc + nd.NE;
    const int row0  = (block*4 + wave)*QF_ROWS_PER_WAVE;

    // the 8 rows of this wave against v_S, all loads in flight together
    const double* __restrict__ M[QF_ROWS_PER_WAVE];
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
    {
        int row = row0 + rr;
        if(row >= Nrows) row = Nrows - 1;   // duplicate work, discarded below
        M[rr] = (row < nd.Nc) ? O.A + (size_t)row*nd.Nc : O.Bt + (size_t)(row - nd.Nc)*nd.Nc;
    }
    double acc[QF_ROWS_PER_WAVE];
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++) acc[rr] = 0.0;
    // (of A only the lower triangle: the splined assembly writes no other. An entry below the diagonal counts twice)
    int rowc[QF_ROWS_PER_WAVE];
    // (the splined models: a row of Bt holds something under the frame's board only - a sixth of its 76 tiles, 46 MB of
    //  zeros a step otherwise: a lane whose tile is empty asks for nothing)
    const unsigned* __restrict__ ob[QF_ROWS_PER_WAVE];
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
    {
        rowc[rr] = min(row0 + rr, Nrows - 1);
        ob[rr] = NULL;
        if(occ != NULL && rowc[rr] >= nd.Nc)
        {
            int blk, a, de, e0;
            E_to_block(nd, rowc[rr] - nd.Nc, &blk, &a, &de, &e0);
            ob[rr] = occ + (size_t)blk*nocc;
        }
    }
#pragma unroll 4
    for(int c = lane; c < nd.Nc; c += 64)
    {
        const double vs = v[S_to_state(nd, c)];
        const int tile = c >> 4;
#pragma unroll
        for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
        {
            const double wgt = (rowc[rr] >= nd.Nc) ? 1.0 : (c < rowc[rr]) ? 2.0 : (c == rowc[rr]) ? 1.0 : 0.0;
            // (a branch around the load: the lanes without one reading the row's first entry instead - no branch, the loads
            //  of four steps in flight - measured slower, 24 us against 20)
            const bool there = (ob[rr] == NULL) || ((ob[rr][tile >> 5] >> (tile & 31)) & 1u);
            if(wgt != 0.0 && there) acc[rr] += wgt*(M[rr][c]*vs);
        }
    }
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
        for(int off=32; off>0; off>>=1) acc[rr] += __shfl_down(acc[rr], off);
    // lane rr finishes row rr
    double mine = 0.0;
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
    {
        const double a0 = __shfl(acc[rr], 0);
        if(lane == rr) mine = a0;
    }
    double t_vNv = 0.0, t_gv = 0.0, t_vv = 0.0;
    const int row = row0 + lane;
    if(lane < QF_ROWS_PER_WAVE && row < Nrows)
    {
        int    is;      // state index of this row's variable
        double wgt;
        if(row < nd.Nc) { is = S_to_state(nd, row); wgt = 1.0; }
        else            { is = E_to_state(nd, row - nd.Nc);                              wgt = 2.0; }
        const double vr = v[is];
        double total = wgt*vr*mine;
        if(row >= nd.Nc)
        {
            int blk, a, de, e0;
            E_to_block(nd, row - nd.Nc, &blk, &a, &de, &e0);
            double s = 0.0;
            for(int c=0;c<de;c++) s += O.D[(size_t)blk*36 + a*6 + c]*v[nd.E_state0 + e0 + c];
            total += vr*s;
        }
        t_vNv = total;
        t_gv  = O.g[is]*vr;
        t_vv  = (vv_E_only && row < nd.Nc) ? 0.0 : vr*vr;
    }
    for(int off=4; off>0; off>>=1)
    {
        t_vNv += __shfl_down(t_vNv, off);
        t_gv  += __shfl_down(t_gv,  off);
        t_vv  += __shfl_down(t_vv,  off);
    }
    __shared__ double part[4][3];
    if(lane == 0) { part[wave][0] = t_vNv; part[wave][1] = t_gv; part[wave][2] = t_vv; }
    __syncthreads();
    if(threadIdx.x < 3)
        return (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    return 0.0;
}
__global__ __launch_bounds__(256)
void quadform_kernel(NormalDims nd, OpRef R, const double* __restrict__ v_in, int v_is_g,
                     double* __restrict__ out_in, int out_in_scalars_at, int nout)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const double* __restrict__ v   = v_is_g ? O.g : v_in;
    double*       __restrict__ out = (out_in != NULL) ? out_in : (O.scalars + out_in_scalars_at);
    const double mine = quadform_body(nd, O, v, blockIdx.x);
    if(threadIdx.x < nout) atomicAdd(&out[threadIdx.x], mine);
}

__global__ __launch_bounds__(256)
void dot_kernel(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out)
{
    double acc = 0.0;
    for(int i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += gridDim.x*blockDim.x)
        acc += a[i]*b[i];
    for(int off=32; off>0; off>>=1) acc += __shfl_down(acc, off);
    __shared__ double part[4];
    if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if(threadIdx.x == 0) atomicAdd(out, part[0]+part[1]+part[2]+part[3]);
}

// y = alpha a + beta b
__global__ __launch_bounds__(256)
void axpby_kernel(int n, double alpha, const double* __restrict__ a, double beta, const double* __restrict__ b,
                  double* __restrict__ y)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i < n) y[i] = alpha*a[i] + ((b != NULL) ? beta*b[i] : 0.0);
}

////////////////////////////////////////////////////////////////////////////////
// outlier rejection
////////////////////////////////////////////////////////////////////////////////
// marks board-corner outliers: weight *= -1 for inliers with |x| > k sigma in
// either coordinate (mrcal.c:4320-4345). counts[0] += number newly marked
__global__ __launch_bounds__(256)
void mark_outliers_kernel(int Npoints_board, double thresh_sq,
                          const double* __restrict__ x, double* __restrict__ pool,
                          int* __restrict__ counts)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= Npoints_board) return;
    const double w = pool[3*(size_t)i + 2];
    if(w <= 0.0) return;
    const double dx = x[2*(size_t)i], dy = x[2*(size_t)i+1];
    if(dx*dx > thresh_sq || dy*dy > thresh_sq)
    {
        pool[3*(size_t)i + 2] = -w;
        atomicAdd(&counts[0], 1);
    }
}

// outlier statistics (mrcal.c:4107-4124, 4282-4306): counts[0] = current
// outliers (weight <= 0), counts[1] = inliers beyond k1 sigma given var,
// sums[0] = sum of inlier x^2
// Fixed grid, grid-stride loop, per-workgroup partial sums in part[], summed in
// order by outlier_stats_sum_kernel: the variance (and with it the outlier
// threshold) does not depend on scheduling
#define OUTLIER_BLOCKS 512
__global__ __launch_bounds__(256)
void outlier_stats_kernel(int Npoints_board, double thresh_sq,
                          const double* __restrict__ x, const double* __restrict__ pool,
                          int* __restrict__ counts, double* __restrict__ part)
{
    double s = 0.0;
    int nout = 0, nbig = 0;
    for(int i = blockIdx.x*blockDim.x + threadIdx.x; i < Npoints_board; i += gridDim.x*blockDim.x)
    {
        const double w = pool[3*(size_t)i + 2];
        if(w <= 0.0) nout++;
        else
        {
            const double dx = x[2*(size_t)i], dy = x[2*(size_t)i+1];
            s += dx*dx + dy*dy;
            if(thresh_sq >= 0.0 && (dx*dx > thresh_sq || dy*dy > thresh_sq)) nbig++;
        }
    }
    for(int off=32; off>0; off>>=1)
    {
        s    += __shfl_down(s, off);
        nout += __shfl_down(nout, off);
        nbig += __shfl_down(nbig, off);
    }
    __shared__ double ps[4];
    if((threadIdx.x & 63) == 0)
    {
        ps[threadIdx.x >> 6] = s;
        if(nout) atomicAdd(&counts[0], nout);       // (integers: any order gives the same sum)
        if(nbig) atomicAdd(&counts[1], nbig);
    }
    __syncthreads();
    if(threadIdx.x == 0) part[blockIdx.x] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
}
__global__ __launch_bounds__(64)
void outlier_stats_sum_kernel(int n, const double* __restrict__ part, double* __restrict__ sums)
{
    if(threadIdx.x != 0) return;
    double s = 0.0;
    for(int i = 0; i < n; i++) s += part[i];
    sums[0] += s;
}

////////////////////////////////////////////////////////////////////////////////
// dog-leg control (libdogleg's trust-region logic, on the device)
////////////////////////////////////////////////////////////////////////////////
// rho test, trust-region update, accept/reject (one thread)
__device__ __forceinline__ void ctl_accept(const OpDev* __restrict__ ops, SolverCtl* ctl)
{
    const int ib = ctl->ib, ia = ctl->ia;
    const OpDev& from = ops[ib];
    // expected improvement: |x|^2 - |x + J s|^2 = -2 g.s - s^T N s
    const double expected = -2.0*from.scalars[SC_STEP_GS] - from.scalars[SC_STEP_SNS];
    ctl->expected_improvement = expected;
    const double observed = ctl->norm2_x[ib] - ctl->norm2_x[ia];
    double rho = observed/expected;
    // a trial point where the cost function is not finite (or a 0/0) is a
    // rejected step with a shrinking trust region, not a comparison with NaN
    // that neither accepts nor shrinks
    if(!(rho == rho) || !(ctl->norm2_x[ia] == ctl->norm2_x[ia])) rho = -1.0;
    double tr = ctl->trustregion;
    if(rho < ctl->trustregion_decrease_threshold)
        tr *= ctl->trustregion_decrease_factor;
    else if(rho > ctl->trustregion_increase_threshold && ctl->did_step_to_edge[ib])
        tr *= ctl->trustregion_increase_factor;
    ctl->trustregion = tr;
    if(rho > 0.0)
    {
        ctl->ib = ia; ctl->ia = ib;
        ctl->Nsteps_accepted++;
    }
    else if(ctl->check_termination &&
            (tr < ctl->trustregion_threshold || tr == 0.0 || !(tr == tr)))
        ctl->done = 1;
}
////////////////////////////////////////////////////////////////////////////////
// the fused step: choose | evaluate | assemble+eliminate | SYRK+finalize | reduce
// | finish+Cholesky | backsub+quadform.  See solver_kernels.hpp
////////////////////////////////////////////////////////////////////////////////
// The Gauss-Newton step is computed EAGERLY: a point that is accepted is
// factored in the launch that accepts it, from the elimination that rode along
// in its assembly. libdogleg computes it lazily at the start of the next trial;
// the step taken is the same. ctl->refactor: the current point must be
// eliminated (again) before a step can be chosen from it - lambda was raised
// after a failed factorization (libdogleg: "singular JtJ: adding lambda I from now
// on"), or its Gauss-Newton step was never computed.
//
// The SAME step runs sharded over several GPUs (frames partitioned over the
// ranks; mrcal_amd/parallel.py, solver.cpp). What a rank computes from its own
// frames only is summed over the ranks in TWO collectives per trial:
//   comm1 (after the reduce):    [ S | r | g_S | |x|^2 | status ]        Nc^2 + 2 Nc + 2 doubles
//   comm2 (after the backsub):   [ g^T N g | |g_E|^2 | |gn_E|^2 | gn_E . g_E ]      4 doubles
// Everything else is either rank-local (the frame part of the state, of g, of the
// steps) or REPLICATED: the camera block of the state, and the control block,
// which every rank advances with the same kernels on the same sums, so that all
// ranks take the same decisions without talking about them. For that the sums a
// rank forms by itself from replicated data must be bit-identical on all ranks:
// fixed-order reductions everywhere, no atomics.
//   comm1 == F.S (S, r and the tail are contiguous); comm2 given to the kernels:
//   sharded. NULL: single GPU, the partial sums are read where they were left.
// The dog-leg step from the current point: dogleg_choose.hpp. As a launch of its own where the evaluation that
// follows has no prologue launch to carry it (problems without boards, the splined models, the protocol driver)
__global__ __launch_bounds__(PRO_T)
void step2_choose_kernel(ChooseArgs a)
{
    __shared__ double scratch[17*7];
    const ChooseOut c = dogleg_choose_scalars(a, scratch);
    dogleg_choose_elementwise(a, c, blockIdx.x*blockDim.x + threadIdx.x);
    if(blockIdx.x == 0 && threadIdx.x == 0) dogleg_choose_record(a, c);
}

// S, r of the point being eliminated (reduction of the SYRK's slots: this rank's
// summand) and, behind them, what else the end of the trial needs from all
// ranks: g_S, |x|^2 and whether a frame block failed to factor. S | r | tail are
// contiguous: ONE all-reduce when sharded
__global__ __launch_bounds__(256)
void step2_reduce_kernel(NormalDims nd, const OpDev* __restrict__ ops, const SolverCtl* __restrict__ ctl,
                         const SolverCtlFlags* __restrict__ fl, int is_leader, int nred,
                         int nslots, const double* __restrict__ Spart,
                         double* __restrict__ S, double* __restrict__ r, const int* __restrict__ status,
                         const unsigned char* __restrict__ live, int* __restrict__ cperm_cur, double* __restrict__ iso,
                         int* __restrict__ err /* SolverCtl::error */,
                         double* __restrict__ ndMA, double* __restrict__ ndMB, int* __restrict__ ndp_cur, int nfill /* workgroups behind the last */,
                         int ride_finish, Step2Dev sd)
{
    if(fl->skip_elim)
    {
        // (nothing is reduced; the end of the trial step still has to be decided where it rides here)
        if(ride_finish && (int)blockIdx.x == nred) (void)step2_finish(sd, (int*)status, true);
        return;
    }
    const OpDev& O = ops[fl->elim_sel];
    if((int)blockIdx.x > nred)
    {
        // The workgroups behind the reduction's own (the dissection): a thread an entry of the copies of the permutation and
        // of the plan that the factorization goes by (one workgroup's ten trips to memory for them were half of this
        // launch's time), then of what no entry of the camera block goes to in the sides' matrices - the borders (zero:
        // the chains' updates add up there) and the pads' rows and columns (identity)
        const long long e_first = (long long)((int)blockIdx.x - nred - 1)*blockDim.x + threadIdx.x, e_step = (long long)nfill*blockDim.x;
        if(cperm_cur != NULL && O.cperm != NULL)
            for(long long e = e_first; e < 2*nd.Nc + 1; e += e_step) cperm_cur[e] = O.cperm[e];
        if(ndp_cur != NULL && O.ndp != NULL)
        {
            const long long nints = O.ndp[NDH_ACTIVE] ? (long long)nd_plan_ints(nd.Nc) : (long long)NDH_WORDS;
            for(long long e = e_first; e < nints; e += e_step) ndp_cur[e] = O.ndp[e];
        }
        if(ndMA == NULL || O.ndp == NULL || !O.ndp[NDH_ACTIVE]) return;
        // a thread an entry, no loop: per side  the border with the rhs row's part of it ((nS+1) x nS) | the pads' rows
        // (pads x nx) | the pads' columns under them ((nS+1) x pads). Sized by the host for the largest plan it provided for
        const int nS = O.ndp[NDH_NS];
        long long e = e_first;
        for(int side = 0; side < 2; side++)
        {
            const int nx = O.ndp[side ? NDH_NB : NDH_NA], nxr = O.ndp[side ? NDH_BRAW : NDH_ARAW], npd = nx - nxr, N = nx + nS;
            double* __restrict__ Mx = side ? ndMB : ndMA;
            const long long e1 = (long long)(nS + 1)*nS, e2 = (long long)npd*nx, e3 = (long long)(nS + 1)*npd;
            if(e < e1)      { const int i = nx + (int)(e/nS), j = nx + (int)(e % nS); if(j <= i) Mx[(size_t)i*N + j] = 0.0; return; }
            e -= e1;
            if(e < e2)      { const int i = nxr + (int)(e/nx), j = (int)(e % nx);     if(j <= i) Mx[(size_t)i*N + j] = (i == j) ? 1.0 : 0.0; return; }
            e -= e2;
            if(e < e3)      { const int i = nx + (int)(e/npd), j = nxr + (int)(e % npd); Mx[(size_t)i*N + j] = 0.0; return; }
            e -= e3;
        }
        return;
    }
    if((int)blockIdx.x < nred)
    {
        // r starts from this rank's g_S: of a point just assembled that is its own summand
        // (every rank adds its own); of a point re-eliminated later it is already the
        // sum over the ranks (step2_finish unpacked it): the leader alone adds it
        const int add_g = (fl->elim_mode == 1) ? 1 : is_leader;
        schur_reduce_body(nd, O, is_leader ? ctl->lambda : 0.0, add_g, nslots, Spart, S, r, blockIdx.x, live,
                          (cperm_cur != NULL && O.cperm != NULL) ? iso : (double*)NULL, err, ndMA, ndMB);
        return;
    }
    // (the permutation this reduction went by, for the factorization and the solve behind it: which of the two
    //  operating points was reduced is the device's to know)
    // (with workgroups behind this one - nfill - the copies are theirs)
    if(nfill == 0 && cperm_cur != NULL && O.cperm != NULL)
        for(int i = threadIdx.x; i < 2*nd.Nc + 1; i += blockDim.x) cperm_cur[i] = O.cperm[i];
    if(nfill == 0 && ndp_cur != NULL && O.ndp != NULL)
    {
        // (not active: the header alone - NDH_NSEFF is what the factorization's launches size themselves by)
        const int nints = O.ndp[NDH_ACTIVE] ? (int)nd_plan_ints(nd.Nc) : (int)NDH_WORDS;
        for(int i = threadIdx.x; i < nints; i += blockDim.x) ndp_cur[i] = O.ndp[i];
    }
    double* __restrict__ tail = r + nd.Nc;
    for(int i = threadIdx.x; i < nd.Nc + 2; i += blockDim.x)
    {
        double v;
        if(i < nd.Nc)       v = O.g[S_to_state(nd, i)];
        else if(i == nd.Nc) v = O.scalars[SC_NORM2_X];
        else                v = (*status != 0) ? 1.0 : 0.0;
        tail[i] = v;
    }
    // (round 5, where the dissection's launches follow: the end of the trial step, which the factorization's first launch
    //  carries otherwise - there the first launch is two workgroups that both need its verdict)
    if(ride_finish) { __syncthreads(); (void)step2_finish(sd, (int*)status, true); }
}

// End of a trial, in ONE workgroup, in front of the factorization: the rho test
// with accept/reject (ctl_accept), the termination tests; then: does the
// (possibly new) current point get its Gauss-Newton step now? Returns that, to
// every thread. comm1 = [S | r | g_S | |x|^2 | status] is complete (summed over
// the ranks when sharded): the camera-block part of the new point's gradient is
// taken from it
// no_unpack: the caller is the reduction that made the tail from this very gradient (single GPU): nothing to copy back
__device__ bool step2_finish(const Step2Dev& sd, int* chol_status, bool no_unpack)
{
    const NormalDims& nd = sd.nd;
    SolverCtl* ctl = sd.ctl;
    SolverCtlFlags* fl = sd.fl;
    const int t = threadIdx.x, nt = blockDim.x;
    const int mode = fl->elim_mode;
    __shared__ int s_go, s_unpack;
    const int ip = sd.initial ? ctl->ib : ctl->ia;        // the point that was evaluated (mode 1)
    const double* __restrict__ tail = sd.comm1_tail;
    const double norm2_x = tail[nd.Nc];
    const bool   eblock_failed = tail[nd.Nc + 1] != 0.0;
    __syncthreads();                      // everyone has read the control state
    if(t == 0)
    {
        if(mode == 1)
        {
            sd.ops[ip].scalars[SC_NORM2_X] = norm2_x;
            ctl->norm2_x[ip]  = norm2_x;
            ctl->gn_valid[ip] = 0;
            ctl->did_step_to_edge[ip] = 0;
            ctl->Nevaluations++;
            if(!sd.initial) ctl_accept(sd.ops, ctl);
        }
        int go = 0, unpack = 0;
        ctl->gn_fresh = 0;
        ctl->derive   = 0;
        if(!ctl->done && ctl->check_termination && ctl->Nsteps_accepted >= ctl->max_iterations)
            ctl->done = 1;
        const int ib = ctl->ib;
        if(mode == 1 && ib == ip) { unpack = 1; ctl->derive = 1; }       // a new current point
        if(!ctl->done)
        {
            if(mode == 2)      go = 1;
            else if(mode == 1) go = (ib == ip);
        }
        if(go && eblock_failed)
        {
            // a 6x6 (3x3) block was not positive definite: regularize, like libdogleg does
            ctl_raise_lambda(ctl);
            ctl->refactor = 1;
            go = 0;
        }
        if(go) ctl->Nfactorizations++;
        fl->skip_backsub = 1;             // until the factorization has succeeded
        fl->skip_chol    = go ? 0 : 1;
        s_go = go; s_unpack = unpack;
        (void)chol_status;
    }
    __syncthreads();
    if(s_unpack && !no_unpack)
    {
        const OpDev& O = sd.ops[ip];
        for(int i = t; i < nd.Nc; i += nt) O.g[S_to_state(nd, i)] = tail[i];
    }
    return s_go != 0;
}
__device__ void step2_chol_done(const Step2Dev& sd, bool not_positive_definite)
{
    SolverCtl* ctl = sd.ctl;
    if(not_positive_definite)
    {
        ctl_raise_lambda(ctl);
        ctl->refactor = 1;
        sd.fl->skip_backsub = 1;
    }
    else
    {
        const int ib = ctl->ib;
        ctl->refactor      = 0;
        ctl->gn_valid[ib]  = 1;
        ctl->gn_lambda[ib] = ctl->lambda;
        ctl->gn_fresh      = 1;
        sd.fl->skip_backsub = 0;
    }
}
// the same around the multi-launch Cholesky of big camera blocks
__global__ __launch_bounds__(1024)
void step2_finish_kernel(Step2Dev sd, int* chol_status)
{
    (void)step2_finish(sd, chol_status);
}
__global__ __launch_bounds__(64)
void step2_post_kernel(Step2Dev sd, const int* __restrict__ chol_status)
{
    if(threadIdx.x == 0 && !sd.fl->skip_chol) step2_chol_done(sd, *chol_status != 0);
}

// After the factorization, side by side in one launch (256 threads):
//   workgroups [0, nbs)   back-substitution d_e = -L^-T (y_e + Wt_e d_s), one WAVE per E block; each block
//                         leaves (|d_e|^2, d_e . g_e) in dots_part[block]
//   workgroup  nbs        d_s into the state-ordered step
//   the rest              the quadratic form g^T N g of a new current point (ctl->derive): per-workgroup
//                         partials into qf_part[.][0], and the frame/point part of |g|^2 into qf_part[.][2]
__global__ __launch_bounds__(256)
void step2_backsub_quadform_kernel(NormalDims nd, BlockRanges br, const OpDev* __restrict__ ops,
                                   const SolverCtl* __restrict__ ctl, const SolverCtlFlags* __restrict__ fl,
                                   const double* __restrict__ Wt, const double* __restrict__ LD,
                                   const double* __restrict__ y, const double* __restrict__ ds,
                                   double* __restrict__ dots_part, double* __restrict__ qf_part, int nbs,
                                   SolverCtl* __restrict__ snap, const unsigned* __restrict__ occ, int nocc)
{
    const OpDev& O = ops[ctl->ib];
    const int b = blockIdx.x;
    // The control block is final for this step (its last writer is the launch before this one): the
    // host's snapshot of it is written straight into pinned memory. As a hipMemcpyAsync it was a copy
    // kernel of its own behind every step, 4-6 us on the stream
    if(snap != NULL && b == 0 && threadIdx.x < (int)(sizeof(SolverCtl)/sizeof(int)))
        ((int*)snap)[threadIdx.x] = ((const int*)ctl)[threadIdx.x];
    if(b > nbs)
    {
        if(!ctl->derive) return;
        const int qb = b - nbs - 1;
        const double mine = quadform_body(nd, O, O.g, qb, true, occ, nocc);
        if(threadIdx.x < 3) qf_part[4*qb + threadIdx.x] = mine;
        return;
    }
    if(fl->skip_backsub) return;
    double* __restrict__ step = O.step_gn;
    if(b == nbs)
    {
        for(int i = threadIdx.x; i < nd.Nc; i += blockDim.x)
            step[S_to_state(nd, i)] = ds[i];
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ibk = 4*b + wave;
    if(ibk >= br.count()) return;
    const int blk = br.block(ibk);
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    // all the loads first: L, y, g_e and this lane's columns of Wt_e against d_s
    // (unconditional, clamped: three loads under conditions were three branches with a wait each)
    const double Lv = LD[(size_t)blk*36 + min(lane, 35)];
    const double yv = y[e0 + min(lane, de - 1)];
    const double gv = O.g[nd.E_state0 + e0 + min(lane, de - 1)];
    double part[6] = {0,0,0,0,0,0};
    // (four column groups asked for together: with a 1206-variable camera block the loop is 19 round trips otherwise)
    // (the tiles of Wt that hold nothing - five in six under the splined models - are not asked for)
    const unsigned* __restrict__ ob = (occ != NULL) ? occ + (size_t)blk*nocc : (const unsigned*)NULL;
#pragma unroll 4
    for(int c = lane; c < nd.Nc; c += 64)
    {
        const double d = ds[c];
        const int tile = c >> 4;
        if(ob != NULL && !((ob[tile >> 5] >> (tile & 31)) & 1u)) continue;
#pragma unroll
        for(int i=0;i<6;i++) if(i < de) part[i] += Wt[(size_t)(e0+i)*nd.Nc + c]*d;
    }
#pragma unroll
    for(int i=0;i<6;i++)
        for(int off=32; off>0; off>>=1) part[i] += __shfl_down(part[i], off);
    // lane 0 holds the sums; L, y, g come from the lanes that loaded them (no LDS, no barrier)
    double v[6], Lr[6][6], ge[6];
#pragma unroll
    for(int i=0;i<6;i++)
    {
        v[i]  = __shfl(yv, i) + __shfl(part[i], 0);
        ge[i] = __shfl(gv, i);
#pragma unroll
        for(int k=0;k<6;k++) Lr[i][k] = __shfl(Lv, i*6 + k);
    }
    if(lane == 0)
    {
        double d2 = 0.0, dg = 0.0;
#pragma unroll
        for(int i=5;i>=0;i--)
        {
            if(i >= de) continue;
            double sacc = v[i];
#pragma unroll
            for(int k=i+1;k<6;k++) if(k < de) sacc -= Lr[k][i]*v[k];
            v[i] = sacc/Lr[i][i];
        }
#pragma unroll
        for(int i=0;i<6;i++)
            if(i < de)
            {
                const double d = -v[i];
                step[nd.E_state0 + e0 + i] = d;
                d2 += d*d; dg += d*ge[i];
            }
        dots_part[2*ibk] = d2; dots_part[2*ibk + 1] = dg;
    }
}

// sharded: this rank's summands of comm2, each summed in a fixed order. One workgroup
__global__ __launch_bounds__(256)
void step2_pack2_kernel(const SolverCtl* __restrict__ ctl, const SolverCtlFlags* __restrict__ fl,
                        const double* __restrict__ qf_part, int qf_n,
                        const double* __restrict__ dots_part, int dots_n, double* __restrict__ comm2)
{
    __shared__ double scratch[17*7];
    double o[2] = {0.0, 0.0}, o2[2] = {0.0, 0.0};
    if(ctl->derive)
        block_sum_fixed<2>(qf_n, [&](int i, double (&t)[2]) { t[0] = qf_part[4*i]; t[1] = qf_part[4*i + 2]; }, o, scratch);
    if(!fl->skip_backsub)
        block_sum_fixed<2>(dots_n, [&](int i, double (&t)[2]) { t[0] = dots_part[2*i]; t[1] = dots_part[2*i + 1]; }, o2, scratch);
    if(threadIdx.x == 0)
    {
        comm2[COMM2_GNG] = o[0]; comm2[COMM2_GGE] = o[1]; comm2[COMM2_GNE2] = o2[0]; comm2[COMM2_GNE_GE] = o2[1];
    }
}

////////////////////////////////////////////////////////////////////////////////
// launchers
////////////////////////////////////////////////////////////////////////////////
hipError_t launch_zero_normal(const NormalDims& nd, const OpRef& R, hipStream_t stream)
{
    const size_t total = (size_t)nd.Nc*nd.Nc + (size_t)nd.NE*nd.Nc + (size_t)nd.NEb*36 + nd.Nstate + NSCALARS;
    int nb = (int)((total + 255)/256); if(nb > 2048) nb = 2048;
    hipLaunchKernelGGL(zero_normal_kernel, dim3(nb), dim3(256), 0, stream, nd, R);
    return hipGetLastError();
}
static size_t assemble_lds_bytes(const NormalDims& nd) { return (size_t)(6*nd.Nc + 42 + 48)*sizeof(double); }
// the first of the rows the assembly takes one lane each: with a plan for the rows that share destinations
// (GenPlan) only the regularization rows are left, whose destinations are their own
static int    assemble_row0(const DeviceProblem& P, const AssemblyPlan& plan)
{
    return (plan.gen.Nrows > 0) ? P.i_meas_regularization : 2*P.W*P.H*P.Nobs_board;
}
static int    assemble_row_blocks(const DeviceProblem& P, const AssemblyPlan& plan)
{
    const int row0 = assemble_row0(P, plan);
    return (P.Nmeas > row0) ? (P.Nmeas - row0 + 255)/256 : 0;
}
// the planned rows: chunks and eliminated blocks in one launch; then, AFTER whatever else finalizes into A, g and
// |x|^2 (launches on a stream are ordered: every destination is added to by one thread at a time), their sums
static hipError_t launch_gen_rows(const NormalDims& nd, const AssemblyPlan& plan, const OpRef& R, const int32_t* Jp, hipStream_t stream)
{
    const GenPlan& G = plan.gen;
    if(G.Nrows <= 0 || G.Nchunks + G.Neblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(gen_rows_kernel, dim3(G.Nchunks + G.Neblocks), dim3(256), gen_lds_bytes(G, nd), stream, nd, R, G, Jp);
    return hipGetLastError();
}
static hipError_t launch_gen_finalize(const NormalDims& nd, const AssemblyPlan& plan, const OpRef& R, hipStream_t stream)
{
    const GenPlan& G = plan.gen;
    if(G.Nrows <= 0 || G.Ndest <= 0) return hipSuccess;
    AssemblyPlan gp = plan;
    gp.Ndest = G.Ndest; gp.dest_id = G.dest_id; gp.dest_begin = G.dest_begin; gp.dest_src = G.dest_src;
    gp.pair_chunk_begin = G.group_chunk_begin; gp.chunk_part = G.part; gp.row_part_n = 0;
    hipLaunchKernelGGL(assemble_finalize_kernel, dim3((G.Ndest*FIN_LANES + 63)/64), dim3(64), 0, stream,
                       G.stride, nd, R.ops, R.sel, R.skip, gp);
    return hipGetLastError();
}
// The block normal equations of a point that was just evaluated, from the Grams
// (or row by row where there are none). The point's normal equations must have
// been cleared (launch_zero_normal / the prologue's side workgroups).
//   board problems with Grams: assemble_factor_kernel (frames | pair chunks | generic rows), then
//   assemble_finalize (here: its own launch; in the fused step it rides along in the SYRK launch)
hipError_t launch_assemble(const DeviceProblem& P, const NormalDims& nd, const BlockRanges& br, const AssemblyPlan& plan,
                           const EvalBuffers& B, hipStream_t stream,
                           hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, bool* forked)
{
    if(forked) *forked = false;
    // splined models: no per-observation Gram; every row goes through the generic path
    const bool by_rows = (P.lens_type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC);
    const int row0 = by_rows ? 0 : 2*P.W*P.H*P.Nobs_board;
    if(P.Nobs_board > 0 && !by_rows)
    {
        const int nframe_blocks = br.frame_hi - br.frame_lo;        // (the 6x6 eliminated blocks: frames, or cameras)
        FactorBuffers none; memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(assemble_factor_kernel, dim3(nframe_blocks + plan.Nchunks*assemble_chunk_slices(P) + assemble_row_blocks(P, plan)), dim3(256),
                           assemble_lds_bytes(nd), stream, P, nd, br, B.R.ops, B.R.sel, B.R.sel, (const SolverCtl*)NULL, B.R.skip,
                           (const int*)NULL, 1, 0, 0.0, plan, B.gram, none, nframe_blocks, assemble_row0(P, plan), P.Nmeas, B.Jp, B.Ji);
        hipError_t e = launch_gen_rows(nd, plan, B.R, B.Jp, stream);
        if(e != hipSuccess) return e;
        if(plan.Ndest > 0)
            hipLaunchKernelGGL(assemble_finalize_kernel, dim3((plan.Ndest*FIN_LANES + 63)/64), dim3(64), 0, stream,
                               gram_stride(P.Ndist), nd, B.R.ops, B.R.sel, B.R.skip, plan);
        e = launch_gen_finalize(nd, plan, B.R, stream);
        if(e != hipSuccess) return e;
    }
    else
    {
        // splined models: the board rows observation by observation (local Grams in
        // LDS), everything else row by row
        int rows_from = row0, rows_to = P.Nmeas;
        hipStream_t gstream = stream;
        const bool splined_boards = by_rows && P.Nobs_board > 0 && P.Nframes > 0;
        bool pairs_early = false;
        // (round 5) rows no fixed-order plan covers go through sums in which no addition rounds (ReproStep): everything
        // then stays on the one stream, the rows' sums are added to the blocks last
        const bool repro = plan.repro.lvl[0] != NULL;
        if(splined_boards)
        {
            rows_from = 2*P.W*P.H*P.Nobs_board;
            if(P.i_meas_regularization >= rows_from && P.i_meas_regularization < P.Nmeas) rows_to = P.i_meas_regularization;
            // What follows assemble_splined_kernel writes the camera block's A and g, and |x|^2: nothing the block
            // elimination or the SYRK read. With a side stream it runs beside them (unless there are other rows
            // - discrete points - that add to A with atomics at the same time)
            const bool use_side = side != NULL && forked != NULL && rows_to == rows_from && !repro;
            // Round 5: the regularization rows' pairs ride in assemble_splined_kernel's launch, behind the frames'
            // workgroups (which write the frames' blocks and Bt, never A or the camera block's g): they were 10 us at the
            // end of the side stream's chain, the longer of the two the reduction waits for. A's entries then take the
            // pairs' products before the gathered sums instead of after (other bits than round 4's, the same every time:
            // the launch boundary orders the two).
            // (Not where assemble_splined_kernel can fall back to row-by-row atomics on A and g - a grid that one
            //  board can cover with more than SPL_MAXSUB sub-boxes, a board of more than 1024 corners -: the pairs'
            //  plain read-modify-writes must not run beside those. Nor on the side stream behind a fork of their own -
            //  the first form of this round -: the fork cost the main stream 8 us; profiles/r05_config2_step_in_time_order.txt
            //  has the gaps the other fork and the join still cost)
            const int  nrp_early  = (P.Nmeas > rows_to) ? (P.Nmeas - rows_to + 511)/512 : 0;
            const bool pairs_ride = nrp_early > 0 && !spl_fallback_possible(P) && rows_to == rows_from;
            // (round 5) which control points a board covers at this point: spl_compact_body, for the reduction of this
            // trial step - one more workgroup of the same launch if its marks fit the launch's LDS, else a launch in front
            const int  nknots_all      = P.Ncameras_intrinsics*P.cfg.spline_Nx*P.cfg.spline_Ny;
            const bool compact_pending = plan.spl_compact != 0;
            const size_t compact_lds   = spl_compact_lds_ints(nknots_all, P.cfg.spline_Nx)*sizeof(int);
            const bool compact_ride    = compact_pending && compact_lds <= SPL_LDS_DOUBLES*sizeof(double);
            if(compact_pending && !compact_ride)
                hipLaunchKernelGGL(spl_compact_kernel, dim3(1), dim3(SPLC_T), compact_lds, stream, P, nd, B.R, plan.nd_lim);
            pairs_early = pairs_ride;
            // (a workgroup per frame and surface; 64 KB of LDS for the tile: two workgroups per CU)
            hipLaunchKernelGGL(assemble_splined_kernel, dim3(2*P.Nframes + (pairs_ride ? nrp_early : 0) + (compact_ride ? 1 : 0)), dim3(256), 0, stream,
                               P, nd, B.R, plan, B.Jp, B.Ji, pairs_ride ? nrp_early : 0, rows_to, compact_ride ? 1 : 0);
            // a copy of the row per wave, as many waves as the LDS holds copies
            const size_t row_bytes = (size_t)(nd.Nc + 1)*sizeof(double);
            const int nwaves = (int)std::min<size_t>(SPLG_WAVES, (size_t)(150*1024)/row_bytes);
            if(nwaves < 1) return hipErrorInvalidValue;
            const int ndense = splg_ndense(P, nd);
            if(use_side)
            {
                hipError_t e = hipEventRecord(ev_fork, stream);           if(e != hipSuccess) return e;
                e = hipStreamWaitEvent(side, ev_fork, 0);                 if(e != hipSuccess) return e;
                *forked = true;
                gstream = side;
            }
            // the knots' rows (a window of columns each), then the rows every pass holds (whole)
            const int nknotrows = splg_nknotrows(P);
            const int window = 2*(P.cfg.spline_order*P.cfg.spline_Nx + P.cfg.spline_order);
            // (orders whose row of A has more than 32 places: by the row-per-workgroup kernel, as until the end of round 4)
            const int order = P.cfg.spline_order;
            const bool pull = order*(2*order + 1) + order + 1 + P.Ncore_state <= 32;
            if(nknotrows > 0 && pull)
                hipLaunchKernelGGL(assemble_splined_gather_knots_kernel, dim3(nknotrows), dim3(64*SPLK_WAVES), 0, gstream, P, nd, B.R, plan);
            else if(nknotrows > 0)
                hipLaunchKernelGGL(assemble_splined_gather_kernel, dim3(nknotrows), dim3(64*SPLG_WAVES),
                                   (size_t)SPLG_WAVES*(window + 1 + 4)*sizeof(double), gstream, P, nd, B.R, plan, SPLG_WAVES, 0, window);
            hipLaunchKernelGGL(assemble_splined_gather_kernel, dim3(ndense*SPLG_E), dim3(64*SPLG_WAVES),
                               nwaves*row_bytes, gstream, P, nd, B.R, plan, nwaves, nknotrows, 0);
            // (the regularization rows: in pairs, rows_pairs_kernel, after whatever other rows there are)
        }
        bool planned = false;
        if(plan.gen.Nrows > 0 && rows_from <= plan.gen.row_first && plan.gen.row_end <= rows_to)
        {
            // the rows that share destinations in a fixed order (GenPlan); what is left of [rows_from, rows_to) is
            // the regularization rows, with destinations of their own: one workgroup, |x|^2 summed in order
            hipError_t e = launch_gen_rows(nd, plan, B.R, B.Jp, stream);
            if(e != hipSuccess) return e;
            e = launch_gen_finalize(nd, plan, B.R, stream);
            if(e != hipSuccess) return e;
            if(rows_to > plan.gen.row_end)
                hipLaunchKernelGGL(rows_single_kernel, dim3(1), dim3(256), 0, stream, nd, B.R, plan.gen.row_end, rows_to, B.Jp, B.Ji);
            planned = true;
        }
        if(repro)
        {
            // the marked board observations (where the grid is big enough for any) and the rows without a plan
            const int nboard = (splined_boards && spl_fallback_possible(P)) ? 2*P.W*P.H*P.Nobs_board : 0;
            const int r0 = planned ? rows_to : rows_from;
            const int nthreads = nboard + std::max(0, rows_to - r0);
            int N = 3; while(((long long)1 << (N - 3)) < (long long)P.Nmeas) N++;      // (as launch_assemble_rows)
            hipError_t e = hipMemsetAsync(plan.repro.any, 0, sizeof(int), stream);
            if(e != hipSuccess) return e;
            if(nthreads > 0)
            {
                hipLaunchKernelGGL(repro_step_colmax_kernel, dim3((nthreads + 255)/256), dim3(256), 0, stream,
                                   P, nd, B.R, plan, nboard, r0, rows_to, B.Jp, B.Ji);
                hipLaunchKernelGGL(repro_step_rows_kernel, dim3((nthreads + 63)/64), dim3(64), 0, stream,
                                   P, nd, B.R, plan, nboard, r0, rows_to, B.Jp, B.Ji, N);
            }
        }
        else if(rows_to > rows_from && !planned)
        {
            // many rows on a small camera block: sum in LDS first (rows_generic_lds_kernel)
            if(nd.Nc <= ROWS_LDS_NC && rows_to - rows_from >= 4096)
                hipLaunchKernelGGL(rows_generic_lds_kernel, dim3((rows_to - rows_from + 255)/256), dim3(256),
                                   4*(size_t)(nd.Nc*nd.Nc + nd.Nc)*sizeof(double), stream,
                                   nd, B.R, rows_from, rows_to, B.Jp, B.Ji);
            else
                hipLaunchKernelGGL(rows_generic_kernel, dim3((rows_to - rows_from + 63)/64), dim3(64), 0, stream,
                                   nd, B.R, rows_from, rows_to, B.Jp, B.Ji);
        }
        if(splined_boards)
        {
            const int nrp = (P.Nmeas > rows_to) ? (P.Nmeas - rows_to + 511)/512 : 0;
            if(nrp > 0 && !pairs_early)
                hipLaunchKernelGGL(rows_pairs_kernel, dim3(nrp), dim3(256), 0, gstream, nd, B.R, rows_to, P.Nmeas, B.Jp, B.Ji, plan.row_part);
            hipLaunchKernelGGL(assemble_splined_combine_kernel, dim3(splg_ndense(P, nd)), dim3(256), 0, gstream, P, nd, B.R, plan, nrp);
            if(gstream != stream)
            {
                const hipError_t e = hipEventRecord(ev_join, gstream);
                if(e != hipSuccess) return e;
            }
        }
        // (the pre-rounded sums, onto whatever the blocks hold by now: the last adder of every entry)
        if(repro)
            hipLaunchKernelGGL(repro_step_combine_kernel, dim3((int)std::min<size_t>(2048, (plan.repro.one + 255)/256)), dim3(256), 0, stream,
                               nd, B.R, plan.repro);
    }
    return hipGetLastError();
}

// Phase 1 of the Gauss-Newton solve, local to a shard: factor the local E
// blocks and form this shard's contribution to the Schur complement and to
// the reduced right-hand side: S_loc = A_loc (+ lambda I) - sum_local Wt^T Wt,
// r_loc = (g_S) - sum_local Wt^T y. The "(...)" terms are added by the shard
// leader only, so that the sum over shards has them once. lambda comes from
// the control block if one is given
// SYRK slicing: ~SYRK_TARGET_WAVES one-wave workgroups per part, slices a
// multiple of the unrolled k-loop. Both parts (frame blocks, point blocks) get
// the same number of slots whether or not they are populated
#ifndef SYRK_TARGET_WAVES
#define SYRK_TARGET_WAVES 2048
#endif
#ifndef SYRK_STRIP_FROM
#define SYRK_STRIP_FROM 256      // camera blocks wider than this: the strip kernel (one A operand for four B operands)
#endif
// workgroups along x of the SYRK launch: tile pairs, or strips of up to SYRK_STRIP of them (big camera blocks)
static int syrk_grid_x(const NormalDims& nd)
{
    const int nb = (nd.Nc + 15)/16;
    if(nd.Nc <= SYRK_STRIP_FROM) return nb*(nb+1)/2;
    int nstrips = 0;
    for(int bi = 0; bi < nb; bi++) nstrips += (nb - bi + SYRK_STRIP - 1)/SYRK_STRIP;
    return nstrips;
}
// The sparse SYRK (the splined models) takes more slices than the dense ones: a strip's blocks that count are few but
// unevenly spread - the strips over the middle of the imager have a third of a slice's blocks to go through, one trip
// to memory each, while nine strips in ten have none -, and a slot nobody wrote costs the reduction a flag
#ifndef SYRK_SPARSE_SLICES
#define SYRK_SPARSE_SLICES 8
#endif
static bool syrk_sparse_range(const NormalDims& nd) { return nd.Nc > SYRK_STRIP_FROM && nd.Nc <= 4096; }
// slices per part at most: a multiple of 8
static int syrk_max_slices(const NormalDims& nd, bool sparse)
{
    if(sparse) return SYRK_SPARSE_SLICES;
    int ns = SYRK_TARGET_WAVES / syrk_grid_x(nd);
    if(ns < 1) ns = 1;
    return (ns + 7) & ~7;
}
static void syrk_slicing(const NormalDims& nd, int nrows, bool sparse, int* nslices, int* e_per_slice)
{
    // no camera variables at all (a solve for the frames alone): no Schur complement, no slices
    if(nd.Nc == 0) { *nslices = 0; *e_per_slice = 4*SYRK_UNROLL; return; }
    int ns = syrk_max_slices(nd, sparse);
    int per = (nrows + ns - 1)/ns;
    per = ((per + 4*SYRK_UNROLL - 1)/(4*SYRK_UNROLL))*(4*SYRK_UNROLL);
    if(per < 4*SYRK_UNROLL) per = 4*SYRK_UNROLL;
    ns = (nrows + per - 1)/per;
    ns = (ns + 7) & ~7;         // a multiple of 8, for the slice -> XCD dealing (syrk_xcd_map); the last ones may be empty
    *nslices = ns; *e_per_slice = per;
}
size_t schur_partial_doubles(const NormalDims& nd)
{
    const int nb = (nd.Nc + 15)/16, npairs = nb*(nb+1)/2;
    if(nd.Nc == 0) return 64;
    // (the slots of the sparse SYRK if the camera block is of its size: which kernel runs is the caller's FactorBuffers::occ)
    const size_t nslots = 2*(size_t)std::max(syrk_max_slices(nd, false), syrk_sparse_range(nd) ? syrk_max_slices(nd, true) : 0);
    return nslots*npairs*256 + nslots*nb*16 + 64 + (nslots*npairs + 7)/8;
}
// the sparse SYRK's flags [nslots][npairs], behind the partial products
static unsigned char* syrk_live_flags(const NormalDims& nd, const FactorBuffers& F, int nslots)
{
    const int nb = (nd.Nc + 15)/16, npairs = nb*(nb+1)/2;
    return (unsigned char*)(F.Spart + (size_t)nslots*npairs*256 + (size_t)nslots*nb*16 + 64);
}

// the SYRK of the local E rows (two contiguous ranges: frames, points) into
// Spart; ride (optional): assemble_finalize() in an extra row of the first launch
static int launch_syrk(const NormalDims& nd, const BlockRanges& br, const int* skip, const FactorBuffers& F,
                       const FinalizeRide* ride, hipStream_t stream, const unsigned char** live /* out: the slots' flags, or NULL */)
{
    const int nb = (nd.Nc + 15)/16, npairs = nb*(nb+1)/2;
    const bool sparse = nd.Nc > SYRK_STRIP_FROM && F.occ != NULL && F.Wtile != NULL;
    int e_lo[2], e_hi[2], ns[2] = {0,0}, per[2] = {0,0};
    for(int part = 0; part < 2; part++)
    {
        br.e_range(nd, part, &e_lo[part], &e_hi[part]);
        if(e_hi[part] > e_lo[part]) syrk_slicing(nd, e_hi[part] - e_lo[part], sparse, &ns[part], &per[part]);
    }
    const int nslots = ns[0] + ns[1];
    unsigned char* flags = sparse ? syrk_live_flags(nd, F, nslots) : NULL;
    *live = flags;
    FinalizeRide none; memset(&none, 0, sizeof(none));
    bool rode = false;
    for(int part = 0, slot0 = 0; part < 2; slot0 += ns[part], part++)
        if(ns[part] > 0)
        {
            const bool with_ride = (ride != NULL && !rode);
            FinalizeRide fr = with_ride ? *ride : none;
            const int gx = (nd.Nc > SYRK_STRIP_FROM) ? syrk_grid_x(nd) : npairs;
            const int extra = with_ride ? (ride->plan.Ndest*FIN_LANES + gx*64 - 1)/(gx*64) : 0;
            fr.row0 = ns[part];
            rode = rode || with_ride;
            if(sparse)
                hipLaunchKernelGGL(schur_syrk_sparse_kernel, dim3(syrk_grid_x(nd), ns[part] + extra), dim3(64*SYRK_SPARSE_WAVES), 0, stream,
                                   nd, skip, e_lo[part], e_hi[part], slot0, nslots, F.Wtile, F.y, F.Spart, ns[part], fr,
                                   F.occ, occ_words(nd), flags);
            else if(nd.Nc > SYRK_STRIP_FROM)
                hipLaunchKernelGGL(schur_syrk_strip_kernel, dim3(syrk_grid_x(nd), ns[part] + extra), dim3(64), 0, stream,
                                   nd, skip, e_lo[part], e_hi[part], per[part], slot0, nslots, F.Wt, F.y, F.Spart, ns[part], fr);
            else
                hipLaunchKernelGGL(schur_syrk_mfma_kernel, dim3(npairs, ns[part] + extra), dim3(64), 0, stream,
                                   nd, skip, e_lo[part], e_hi[part], per[part], slot0, nslots, F.Wt, F.y, F.Spart, ns[part], fr);
        }
    if(ride != NULL && !rode && ride->plan.Ndest > 0)
        hipLaunchKernelGGL(assemble_finalize_kernel, dim3((ride->plan.Ndest*FIN_LANES + 63)/64), dim3(64), 0, stream,
                           ride->npos, nd, ride->ops, ride->sel, ride->skip, ride->plan);
    return nslots;
}

hipError_t launch_factor_local(const NormalDims& nd, const BlockRanges& br,
                               const OpRef& R, const FactorBuffers& F,
                               double lambda, const SolverCtl* ctl, bool is_leader, hipStream_t stream)
{
    if(br.count() > 0)
        hipLaunchKernelGGL(eblock_factor_kernel, dim3(br.count()), dim3(nd.Nc > 255 ? 256 : 64), 0, stream,
                           nd, br, 0, R, lambda, ctl, F.Wt, F.LD, F.y, F.status, F.occ, occ_words(nd), F.Wtile);
    const int nb = (nd.Nc + 15)/16, npairs = nb*(nb+1)/2;
    const unsigned char* live = NULL;
    const int nslots = launch_syrk(nd, br, R.skip, F, NULL, stream, &live);
    {
        const int n = (npairs*256 + nb*16)*(live ? 1 : SRED_SPLIT);
        hipLaunchKernelGGL(schur_reduce_kernel, dim3((n + 255)/256), dim3(256), 0, stream,
                           nd, R, lambda, ctl, is_leader ? 1 : 0, nslots, F.Spart, F.S, F.r, live);
    }
    return hipGetLastError();
}

// Phase 2: dense Cholesky of the (summed) Schur complement, d_S, and the
// back-substitution of the local E blocks into step_gn of the point
hipError_t launch_solve_backsub(const NormalDims& nd, const BlockRanges& br,
                                const OpRef& R, const FactorBuffers& F, const int* skip_also, bool keep_factor,
                                hipStream_t stream)
{
    {
        const int n = nd.Nc;
        if(chol_fits_lds(n))
        {
            Step2Dev none; memset(&none, 0, sizeof(none));
            hipLaunchKernelGGL(schur_cholesky_solve_kernel<0>, dim3(1), dim3(1024), chol_lds_bytes(n), stream,
                               n, R.skip, keep_factor ? 1 : 0, F.S, F.r, F.status, none);
        }
        else if(F.Linv != NULL)
            launch_cholesky_large(n, R.skip, F.S, F.Linv, F.status, stream, NULL, NULL, NULL, NULL, 0, NULL, NULL, false, F.use_sweep != 0);
        else
            hipLaunchKernelGGL(schur_cholesky_solve_global_kernel, dim3(1), dim3(1024), 0, stream,
                               n, R.skip, F.S, F.r, F.status);
    }
    hipLaunchKernelGGL(backsub_kernel, dim3(br.count()+1), dim3(64), 0, stream,
                       nd, br, R, skip_also, F.Wt, F.LD, F.y, F.r);
    return hipGetLastError();
}

static int quadform_blocks(const NormalDims& nd)
{
    return (nd.Nc + nd.NE + 4*QF_ROWS_PER_WAVE - 1)/(4*QF_ROWS_PER_WAVE);
}
hipError_t launch_quadform(const NormalDims& nd, const OpRef& R, const double* v, double* out,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(quadform_kernel, dim3(quadform_blocks(nd)), dim3(256), 0, stream,
                       nd, R, v, 0, out, 0, 1);
    return hipGetLastError();
}
hipError_t launch_dot(int n, const double* a, const double* b, double* out, hipStream_t stream)
{
    int nb = (n + 255)/256; if(nb > 1024) nb = 1024; if(nb < 1) nb = 1;
    hipLaunchKernelGGL(dot_kernel, dim3(nb), dim3(256), 0, stream, n, a, b, out);
    return hipGetLastError();
}
hipError_t launch_axpby(int n, double alpha, const double* a, double beta, const double* b, double* y,
                        hipStream_t stream)
{
    if(n <= 0) return hipSuccess;
    hipLaunchKernelGGL(axpby_kernel, dim3((n+255)/256), dim3(256), 0, stream, n, alpha, a, beta, b, y);
    return hipGetLastError();
}
size_t outlier_partial_doubles() { return OUTLIER_BLOCKS; }
hipError_t launch_outlier_stats(int Npoints_board, double thresh_sq, const double* x, const double* pool,
                                int* counts, double* sums, double* part, hipStream_t stream)
{
    if(Npoints_board <= 0) return hipSuccess;
    int nb = (Npoints_board + 255)/256; if(nb > OUTLIER_BLOCKS) nb = OUTLIER_BLOCKS;
    hipLaunchKernelGGL(outlier_stats_kernel, dim3(nb), dim3(256), 0, stream,
                       Npoints_board, thresh_sq, x, pool, counts, part);
    hipLaunchKernelGGL(outlier_stats_sum_kernel, dim3(1), dim3(64), 0, stream, nb, part, sums);
    return hipGetLastError();
}
hipError_t launch_mark_outliers(int Npoints_board, double thresh_sq, const double* x, double* pool,
                                int* counts, hipStream_t stream)
{
    if(Npoints_board <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_outliers_kernel, dim3((Npoints_board+255)/256), dim3(256), 0, stream,
                       Npoints_board, thresh_sq, x, pool, counts);
    return hipGetLastError();
}

// ---- the device-controlled step. ctl is followed in memory by its SolverCtlFlags
static SolverCtlFlags* ctl_flags(SolverCtl* ctl) { return (SolverCtlFlags*)(ctl + 1); }
const int* solver_ctl_skip_factor(const SolverCtl* ctl) { return &((const SolverCtlFlags*)(ctl + 1))->skip_factor; }
const int* solver_ctl_skip_eval  (const SolverCtl* ctl) { return &((const SolverCtlFlags*)(ctl + 1))->skip_eval; }
size_t     solver_ctl_bytes() { return sizeof(SolverCtl) + sizeof(SolverCtlFlags); }
// host image of [ctl | flags] before a run: the first thing the fused step does is
// the assembly + elimination of the starting point icur
void       solver_ctl_init_flags(void* ctl_image, int icur)
{
    SolverCtlFlags* fl = (SolverCtlFlags*)((SolverCtl*)ctl_image + 1);
    memset(fl, 0, sizeof(*fl));
    fl->elim_mode = 1; fl->elim_sel = icur; fl->skip_chol = 1; fl->skip_backsub = 1;
}

// what this shard does not own of a state vector, zeroed: the sum over the shards is then the state
__global__ __launch_bounds__(256)
void mask_state_kernel(NormalDims nd, BlockRanges br, int is_leader, double* __restrict__ b)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= nd.Nstate) return;
    bool mine;
    const int se = state_to_SE(nd, i);
    if(se >= 0) mine = is_leader != 0;        // the camera block
    else
    {
        const int e = -se - 1;
        int lo, hi, lo2, hi2;
        br.e_range(nd, 0, &lo, &hi);
        br.e_range(nd, 1, &lo2, &hi2);
        mine = (e >= lo && e < hi) || (e >= lo2 && e < hi2);
    }
    if(!mine) b[i] = 0.0;
}
hipError_t launch_mask_state(const NormalDims& nd, const BlockRanges& br, bool is_leader, double* b, hipStream_t stream)
{
    hipLaunchKernelGGL(mask_state_kernel, dim3((nd.Nstate + 255)/256), dim3(256), 0, stream, nd, br, is_leader ? 1 : 0, b);
    return hipGetLastError();
}

// ---- the fused step
static int quadform_blocks(const NormalDims& nd);
const int* solver_ctl_skip_eval2(const SolverCtl* ctl) { return &((const SolverCtlFlags*)(ctl + 1))->skip_eval; }

ChooseArgs step2_choose_args(const Step2Args& a)
{
    ChooseArgs c;
    c.nd = *a.nd; c.ops = a.ops; c.ctl = a.ctl; c.fl = ctl_flags(a.ctl); c.chol_status = a.F->status; c.step = a.step;
    c.qf_part = a.plan->qf_part; c.qf_n = quadform_blocks(*a.nd); c.dots_part = a.plan->dots_part; c.dots_n = a.br->count();
    c.comm2 = a.comm2;
    return c;
}
hipError_t launch_step2_choose(const Step2Args& a, hipStream_t stream)
{
    const NormalDims& nd = *a.nd;
    // (workgroups of 64, like the prologue's when the choice rides there: the fixed-order sums depend on the
    //  workgroup size, and the ranks of a sharded solve - with or without boards in their shard - must get the same bits)
    hipLaunchKernelGGL(step2_choose_kernel, dim3((nd.Nstate + PRO_T - 1)/PRO_T), dim3(PRO_T), 0, stream, step2_choose_args(a));
    return hipGetLastError();
}

// (launch_step2_assemble left work on the side stream: launch_step2_reduce, which always follows it, joins)
static thread_local bool step2_side_pending = false;
// the block normal equations of the point the flags name, and the elimination of its frame/point blocks
hipError_t launch_step2_assemble(const Step2Args& a, bool initial, hipStream_t stream)
{
    step2_side_pending = false;
    const DeviceProblem& P = *a.P;
    const NormalDims& nd = *a.nd;
    const BlockRanges& br = *a.br;
    SolverCtlFlags* fl = ctl_flags(a.ctl);
    const int* sel_eval = initial ? &a.ctl->ib : &a.ctl->ia;
    const bool by_rows = (P.lens_type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC);
    int nframes_fused = 0;
    if(P.Nobs_board > 0 && !by_rows)
    {
        const int row0 = 2*P.W*P.H*P.Nobs_board;
        nframes_fused = br.frame_hi - br.frame_lo;
        hipLaunchKernelGGL(assemble_factor_kernel, dim3(nframes_fused + a.plan->Nchunks*assemble_chunk_slices(P) + assemble_row_blocks(P, *a.plan)), dim3(256),
                           assemble_lds_bytes(nd), stream, P, nd, br, a.ops, sel_eval, &a.ctl->ib, a.ctl, (const int*)NULL,
                           &fl->elim_mode, 0, 1, 0.0, *a.plan, a.gram, *a.F, nframes_fused, assemble_row0(P, *a.plan), P.Nmeas, a.Jp, a.Ji);
        // the planned rows of the evaluated point (their own launch: they ride on nothing; a trial without an
        // evaluation skips them). Their sums are added after the Grams' (launch_step2_reduce)
        (void)row0;
        const hipError_t e = launch_gen_rows(nd, *a.plan, OpRef{ a.ops, sel_eval, &fl->skip_asm }, a.Jp, stream);
        if(e != hipSuccess) return e;
    }
    else
    {
        // no Grams (splined models, problems without boards): the atomic row-by-row assembly of the evaluated point
        EvalBuffers B; memset(&B, 0, sizeof(B));
        B.R = OpRef{ a.ops, sel_eval, &fl->skip_asm }; B.Jp = (int32_t*)a.Jp; B.Ji = (int32_t*)a.Ji;
        bool forked = false;
        const hipError_t e = launch_assemble(P, nd, br, *a.plan, B, stream, a.side, a.ev_fork, a.ev_join, &forked);
        if(e != hipSuccess) return e;
        step2_side_pending = forked;
    }
    // the blocks the fused kernel did not eliminate: all of them on the row-by-row
    // path; the point blocks otherwise (their rows are accumulated in the same launch)
    const int nrest = br.count() - nframes_fused;
    if(nrest > 0)
    {
        const OpRef R = { a.ops, &fl->elim_sel, &fl->skip_elim };
        // (the occupancy of Wt's tiles is tracked only when EVERY block comes through here: F.occ is only
        //  allocated for the splined models, whose blocks all do)
        hipLaunchKernelGGL(eblock_factor_kernel, dim3(nrest), dim3(nd.Nc > 255 ? 256 : 64), 0, stream,
                           nd, br, nframes_fused, R, 0.0, a.ctl, a.F->Wt, a.F->LD, a.F->y, a.F->status,
                           (nframes_fused == 0) ? a.F->occ : (unsigned*)NULL, occ_words(nd),
                           (nframes_fused == 0) ? a.F->Wtile : (double*)NULL);
    }
    return hipGetLastError();
}

// SYRK (+ finalize of A, g, |x|^2) | S, r and the tail of comm1. (Sharded: comm1 is all-reduced after this)
// Does the end-of-trial logic (step2_finish) ride in the reduction's launch (round 5)? On a single GPU the tail it reads -
// g_S, |x|^2, the block elimination's status - is complete when the reduction's last workgroup has written it, and that
// workgroup can decide the trial there and then, beside the others: the factorization's first launch starts on its matrix
// at once (and may be several workgroups: the dissection's). Sharded, the tail is summed over the ranks behind this launch.
// (With the backward sweep - FactorBuffers::use_sweep - the end-of-trial logic and the verdict are launches of their own.)
static bool step2_finish_rides(const Step2Args& a)
{
    return a.comm2 == NULL && !a.F->use_sweep;
}
hipError_t launch_step2_reduce(const Step2Args& a, hipStream_t stream, int initial)
{
    const DeviceProblem& P = *a.P;
    const NormalDims& nd = *a.nd;
    const BlockRanges& br = *a.br;
    const FactorBuffers& F = *a.F;
    SolverCtlFlags* fl = ctl_flags(a.ctl);
    const bool with_grams = P.Nobs_board > 0 && P.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC;
    FinalizeRide ride; memset(&ride, 0, sizeof(ride));
    if(with_grams && a.plan->Ndest > 0)
    {
        ride.npos = gram_stride(P.Ndist); ride.ops = a.ops; ride.sel = &fl->elim_sel; ride.skip = &fl->skip_asm; ride.plan = *a.plan;
    }
    const unsigned char* live = NULL;
    const int nslots = launch_syrk(nd, br, &fl->skip_elim, F, ride.npos ? &ride : NULL, stream, &live);
    if(with_grams)
    {
        // (after the ride: one adder per destination at a time)
        const hipError_t e = launch_gen_finalize(nd, *a.plan, OpRef{ a.ops, &fl->elim_sel, &fl->skip_asm }, stream);
        if(e != hipSuccess) return e;
    }
    // A, g of the camera block and |x|^2 may still be on their way on the side stream
    if(step2_side_pending)
    {
        const hipError_t e = hipStreamWaitEvent(stream, a.ev_join, 0);
        if(e != hipSuccess) return e;
        step2_side_pending = false;
    }
    const int nb = (nd.Nc + 15)/16, npairs = nb*(nb+1)/2;
    const int nred = ((npairs*256 + nb*16)*(live ? 1 : SRED_SPLIT) + 255)/256;
    // (the dissection: its matrices' borders and pads by nfill more workgroups; where its launches follow, the end of the
    //  trial step rides in this launch's last ordinary workgroup - launch_step2_factor() then leaves it out)
    const bool rides = initial >= 0 && step2_finish_rides(a);
    const bool nd_on = F.ndMA != NULL && F.cperm_cur != NULL;
    const bool nd_launches = nd_on && F.nd_lim.rounds > 0 && rides;
    int nfill = 0;
    if(nd_launches)
    {
        // (a thread an entry of what the dissection's matrices hold beside the camera block's entries: step2_reduce_kernel)
        const long long NSp = F.nd_lim.ns_max, nxm = (long long)ND_PANEL*F.nd_lim.rounds, pads = ND_PANEL - 1;
        const long long per = (NSp + 1)*NSp + pads*nxm + (NSp + 1)*pads;
        nfill = (int)((2*per + 255)/256);
        const int ncopy = (int)((std::max<long long>(2*nd.Nc + 1, (long long)nd_plan_ints(nd.Nc)) + 255)/256);
        nfill = std::max(nfill, ncopy);
    }
    Step2Dev sd; memset(&sd, 0, sizeof(sd));
    if(rides) { sd.nd = nd; sd.ops = a.ops; sd.ctl = a.ctl; sd.fl = fl; sd.initial = initial ? 1 : 0; sd.comm1_tail = F.r + nd.Nc; }
    hipLaunchKernelGGL(step2_reduce_kernel, dim3(nred + 1 + nfill), dim3(256), 0, stream,
                       nd, a.ops, a.ctl, fl, a.is_leader ? 1 : 0, nred, nslots, F.Spart, F.S, F.r, F.status, live, F.cperm_cur, F.iso, &a.ctl->error,
                       nd_launches ? F.ndMA : (double*)NULL, nd_launches ? F.ndMB : (double*)NULL, nd_on ? F.ndp_cur : (int*)NULL, nfill,
                       rides ? 1 : 0, sd);
    return hipGetLastError();
}
int64_t step2_comm1_doubles(const NormalDims& nd) { return (int64_t)nd.Nc*nd.Nc + 2*nd.Nc + 2; }

// finish + Cholesky | back-substitution + quadratic form | (sharded) this rank's summands of comm2
hipError_t launch_step2_factor(const Step2Args& a, bool initial, hipStream_t stream)
{
    const NormalDims& nd = *a.nd;
    const BlockRanges& br = *a.br;
    const FactorBuffers& F = *a.F;
    SolverCtlFlags* fl = ctl_flags(a.ctl);
    Step2Dev sd;
    sd.nd = nd; sd.ops = a.ops; sd.ctl = a.ctl; sd.fl = fl; sd.initial = initial ? 1 : 0;
    sd.comm1_tail = F.r + nd.Nc;
    {
        const int n = nd.Nc;
        // (round 5, single GPU: the end-of-trial logic has run in the reduction's launch - step2_finish_rides())
        const bool finish_done = step2_finish_rides(a);
        if(chol_fits_lds(n))
        {
            if(finish_done)
                hipLaunchKernelGGL(schur_cholesky_solve_kernel<2>, dim3(1), dim3(1024), chol_lds_bytes(n), stream,
                                   n, (const int*)&fl->skip_chol, 0, F.S, F.r, F.status, sd);
            else
                hipLaunchKernelGGL(schur_cholesky_solve_kernel<1>, dim3(1), dim3(1024), chol_lds_bytes(n), stream,
                                   n, (const int*)NULL, 0, F.S, F.r, F.status, sd);
        }
        else
        {
            // (round 5: finish and post ride in the factorization's first and last launch; with the backward sweep the
            //  factorization's last launch is another: launches of their own then)
            const bool separate = F.use_sweep != 0;
            bool fused = false;
            if(separate) hipLaunchKernelGGL(step2_finish_kernel, dim3(1), dim3(1024), 0, stream, sd, F.status);
            // (the splined models: the camera block as the reduction left it - without the control points no board covers)
            LcholCompact cp; memset(&cp, 0, sizeof(cp));
            const bool compact = F.cperm_cur != NULL;       // (what the reduction went by; never with the backward sweep: solver.cpp)
            if(compact) { cp.cperm = F.cperm_cur; cp.iso = F.iso; cp.dout = F.r; cp.Nc = n; }
            // (the dissection's launches, where the host has provided for them: learn_likely_size())
            const bool nd_launches = compact && F.ndMA != NULL && F.nd_lim.rounds > 0 && finish_done;
            LcholNdLaunch nds; memset(&nds, 0, sizeof(nds));
            if(nd_launches)
            {
                const int* h = F.ndp_cur;
                nds.A = LcholChain{ F.ndMA, F.ndLinvA, h + NDH_NA, h + NDH_NS };
                nds.B = LcholChain{ F.ndMB, F.ndLinvB, h + NDH_NB, h + NDH_NS };
                nds.ndh = h; nds.lim = F.nd_lim;
                cp.ndh = h; cp.ndMA = F.ndMA; cp.ndMB = F.ndMB; cp.ndpart = F.ndPart;
            }
            launch_cholesky_large(n, &fl->skip_chol, F.S, F.Linv, F.status, stream, separate ? NULL : &sd, &fused,
                                  compact ? (nd_launches ? F.ndp_cur + NDH_NSEFF : F.cperm_cur + 2*n) : (const int*)NULL,
                                  compact ? &cp : (const LcholCompact*)NULL,
                                  compact ? (nd_launches ? F.nd_likely_panels : F.lchol_likely_panels) : 0,
                                  compact ? (unsigned*)(F.cperm_cur + 2*n + 1) : (unsigned*)NULL, nd_launches ? &nds : (const LcholNdLaunch*)NULL,
                                  finish_done, F.use_sweep != 0);
            if(separate) hipLaunchKernelGGL(step2_post_kernel, dim3(1), dim3(64), 0, stream, sd, F.status);
            else if(!fused) return hipErrorInvalidValue;
        }
    }
    const int nbs = (br.count() + 3)/4, nqf = quadform_blocks(nd);
    hipLaunchKernelGGL(step2_backsub_quadform_kernel, dim3(nbs + 1 + nqf), dim3(256), 0, stream,
                       nd, br, a.ops, a.ctl, fl, F.Wt, F.LD, F.y, F.r, a.plan->dots_part, a.plan->qf_part, nbs, a.snap,
                       (nd.Nc > SYRK_STRIP_FROM) ? F.occ : (const unsigned*)NULL, occ_words(nd));
    if(a.comm2 != NULL)
        hipLaunchKernelGGL(step2_pack2_kernel, dim3(1), dim3(256), 0, stream,
                           a.ctl, fl, a.plan->qf_part, nqf, a.plan->dots_part, br.count(), (double*)a.comm2);
    return hipGetLastError();
}

} // namespace mrcal_amd

////////////////////////////////////////////////////////////////////////////////
// solves against a kept factorization (the CHOLMOD_factorization equivalent):
// (JtJ) x = b with N = JtJ = [A B; Bt D] factored as in launch_factor_local()
// + schur_cholesky (keep_factor): the E blocks' L_e in LD, Wt = L_e^-1 Bt_e,
// and the Cholesky factor of the Schur complement in the lower triangle of S.
//   y_e = L_e^-1 b_e ;  r = b_S - Wt^T y ;  x_S = S^-1 r ;  x_e = L_e^-T (y_e - Wt_e x_S)
////////////////////////////////////////////////////////////////////////////////
namespace mrcal_amd {

// The factorization kept by launch_factor_local() + the Cholesky of S is an LL^T
// factorization of the PERMUTED matrix: with the eliminated blocks first,
//     P (JtJ) P^T = [ D  Bt ]  =  L L^T ,   L = [ L_E   0  ]      L_E = blockdiag(chol(D_e))   (F.LD)
//                   [ B  A  ]                   [ Wt^T  L_S ]     Wt  = L_E^-1 Bt              (F.Wt)
//                                                                 L_S = chol(S)               (F.S, lower)
// "factor order" = the order of P: [ E (frames, then points) | S (intrinsics, extrinsics, warp) ].
// order 0: vectors in state order; 1: in factor order.
__device__ __forceinline__ int fs_index_E(const NormalDims& nd, int order, int e) { return order ? e : E_to_state(nd, e); }
__device__ __forceinline__ int fs_index_S(const NormalDims& nd, int order, int c)
{
    return order ? nd.NE + c : (S_to_state(nd, c));
}

// y_e = L_e^-1 b_e, one thread per E block
__global__ __launch_bounds__(64)
void fsolve_forward_kernel(NormalDims nd, const double* __restrict__ LD,
                           const double* __restrict__ b, double* __restrict__ y, int order, size_t sb)
{
    // (blockIdx.z in every fsolve kernel: the right-hand side of a batch; sb its stride in b and x.
    //  y, r and the partial sums of a batch lie one right-hand side after the other)
    b += blockIdx.z*sb; y += (size_t)blockIdx.z*nd.NE;
    const int blk = blockIdx.x*blockDim.x + threadIdx.x;
    if(blk >= nd.NEb) return;
    const int de = (blk < nd.Nfb) ? 6 : 3;
    const int e0 = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    const double* __restrict__ L = LD + (size_t)blk*36;
    double w[6];
    for(int i=0;i<de;i++)
    {
        double v = b[fs_index_E(nd, order, e0 + i)];
        for(int k=0;k<i;k++) v -= L[i*6+k]*w[k];
        w[i] = v / L[i*6+i];
        y[e0+i] = w[i];
    }
}
// r[c] = b_S[c] - sum_e Wt[e][c] y[e]. (One thread per S column walking all of Wt: 1.36 ms at 6000 x 140.)
// In two launches: a workgroup sums a slab of rows for 256 columns (coalesced across the columns) into
// part[slab][c]; then one thread per column adds the slabs IN ORDER: no atomics, the same bits every time
__global__ __launch_bounds__(256)
void fsolve_reduce_partial_kernel(NormalDims nd, const double* __restrict__ Wt, const double* __restrict__ y,
                                  double* __restrict__ part, int rows_per_slab)
{
    y += (size_t)blockIdx.z*nd.NE; part += (size_t)blockIdx.z*gridDim.y*nd.Nc;
    const int c = blockIdx.x*blockDim.x + threadIdx.x;
    const int e0 = blockIdx.y*rows_per_slab, e1 = min(nd.NE, e0 + rows_per_slab);
    if(c >= nd.Nc) return;
    double acc0 = 0.0, acc1 = 0.0;
    int e = e0;
    for(; e + 1 < e1; e += 2)
    {
        acc0 += Wt[(size_t)e*nd.Nc + c]*y[e];
        acc1 += Wt[(size_t)(e+1)*nd.Nc + c]*y[e+1];
    }
    if(e < e1) acc0 += Wt[(size_t)e*nd.Nc + c]*y[e];
    part[(size_t)blockIdx.y*nd.Nc + c] = acc0 + acc1;
}
// (16 lanes per column: lane k adds the slabs k, k+16, ... in order, then the 16 sums are added in lane order)
__global__ __launch_bounds__(256)
void fsolve_reduce_kernel(NormalDims nd, const double* __restrict__ part, int nslabs,
                          const double* __restrict__ b, double* __restrict__ r, int order, size_t sb)
{
    b += blockIdx.z*sb; r += (size_t)blockIdx.z*nd.Nc; part += (size_t)blockIdx.z*nslabs*nd.Nc;
    const int gid = blockIdx.x*blockDim.x + threadIdx.x;
    const int c = gid >> 4, k = gid & 15;
    const bool ok = c < nd.Nc;
    double acc = 0.0;
    if(ok) for(int s = k; s < nslabs; s += 16) acc += part[(size_t)s*nd.Nc + c];
    // fixed order: ((0+1)+(2+3))+... over the 16 lanes of the column
    for(int off = 1; off < 16; off <<= 1) acc += __shfl_xor(acc, off);
    if(ok && k == 0) r[c] = b[fs_index_S(nd, order, c)] - acc;
}
// r <- L^-1 r (parts & 1), then r <- L^-T r (parts & 2), L the lower triangle of S (row-major n x n), one workgroup
__global__ __launch_bounds__(1024)
void fsolve_dense_kernel(int n, const double* __restrict__ S, double* __restrict__ r, int parts)
{
    r += (size_t)blockIdx.x*n;
    const int t = threadIdx.x, nt = blockDim.x;
    __shared__ double piv;
    if(parts & 1)
    for(int j=0;j<n;j++)
    {
        if(t == 0) { piv = r[j]/S[(size_t)j*n + j]; r[j] = piv; }
        __syncthreads();
        const double pj = piv;
        for(int i=j+1+t;i<n;i+=nt) r[i] -= S[(size_t)i*n + j]*pj;
        __syncthreads();
    }
    if(parts & 2)
    for(int j=n-1;j>=0;j--)
    {
        if(t == 0) { piv = r[j]/S[(size_t)j*n + j]; r[j] = piv; }
        __syncthreads();
        const double pj = piv;
        for(int i=t;i<j;i+=nt) r[i] -= S[(size_t)j*n + i]*pj;
        __syncthreads();
    }
}
// The same with the factor in LDS (n <= 178, the sizes schur_cholesky_solve_kernel keeps there): the whole
// workgroup loads the packed triangle, then ONE wave runs the two sweeps with r in registers (lane l holds
// entries l, l+64, l+128) - a step is a cross-lane read of the pivot entry and one multiply-add per slot, its
// multipliers (a column of L going forward, a row going back) requested a step ahead. The global-memory
// version above pays two workgroup barriers and a memory round trip per column: 98 us at n = 140, this 12
__global__ __launch_bounds__(1024)
void fsolve_dense_lds_kernel(int n, const double* __restrict__ S, double* __restrict__ r, int parts)
{
    extern __shared__ __attribute__((aligned(16))) double Lp[];      // packed lower triangle, then 1/diagonal
    r += (size_t)blockIdx.x*n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto rowptr = [&](int i) -> double* { return Lp + ((i*(i+1)) >> 1); };
    double* __restrict__ rds = Lp + (((n*(n+1)) >> 1) + 1);
    {
        // everything asked for before anything is stored (as in schur_cholesky_solve_kernel)
        double v[12][3];
#pragma unroll
        for(int a = 0; a < 12; a++)
#pragma unroll
            for(int b = 0; b < 3; b++)
                if(b <= a/4)
                {
                    const int  i = wave_u + 16*a, j = lane + 64*b;
                    const bool ok = (i < n && j <= i);
                    v[a][b] = S[ok ? (size_t)i*n + j : 0];
                }
#pragma unroll
        for(int a = 0; a < 12; a++)
#pragma unroll
            for(int b = 0; b < 3; b++)
                if(b <= a/4)
                {
                    const int i = wave_u + 16*a, j = lane + 64*b;
                    if(i < n && j <= i) { rowptr(i)[j] = v[a][b]; if(j == i) rds[i] = 1.0/v[a][b]; }
                }
    }
    __syncthreads();
    if(wave != 0) return;
    // slot k of lane l = entry l + 64 k. Every LDS read below is UNCONDITIONAL (index clamped into the
    // triangle, value selected afterwards): a load under a condition becomes a branch with a wait of its own,
    // and a step of the sweeps was 800 cycles of those
    double z[3];
    int    tri[3];                      // start of row i of the packed triangle (row n-1 for the lanes past the end)
#pragma unroll
    for(int k = 0; k < 3; k++)
    {
        const int i = lane + 64*k, ic = min(i, n - 1);
        z[k]   = r[ic];
        z[k]   = (i < n) ? z[k] : 0.0;
        tri[k] = (ic*(ic+1)) >> 1;
    }
    // (the slot that holds entry j is a compile-time constant inside each of the three j ranges below: indexing
    //  z[] with a run-time slot number would put the array into scratch memory)
    if(parts & 1)
    {
        // L w = r, right-looking: w_j = z_j / L_jj ; z_i -= L[i][j] w_j for i > j
        double col[3];
#pragma unroll
        for(int k = 0; k < 3; k++) { const int i = lane + 64*k; const double v = Lp[tri[k]]; col[k] = (i < n && i > 0) ? v : 0.0; }
        double rdj = rds[0];
        auto sweep = [&](auto KS)
        {
            constexpr int ks = decltype(KS)::value;
            for(int j = 64*ks; j < min(n, 64*ks + 64); j++)
            {
                const double wj = readlane_f64(z[ks], j & 63)*rdj;
                double nxt[3];
#pragma unroll
                for(int k = 0; k < 3; k++)
                {
                    // (column j+1 of row i; rows i <= j+1 read their own diagonal instead: inside the row)
                    const int i = lane + 64*k, ic = min(i, n - 1);
                    const double v = Lp[tri[k] + min(j + 1, ic)];
                    nxt[k] = (i < n && i > j + 1) ? v : 0.0;
                }
                const double rdn = rds[min(j + 1, n - 1)];
#pragma unroll
                for(int k = 0; k < 3; k++)
                {
                    const int i = lane + 64*k;
                    const double upd = fma(-col[k], wj, z[k]);
                    z[k] = (i == j) ? wj : (i > j) ? upd : z[k];
                    col[k] = nxt[k];
                }
                rdj = rdn;
            }
        };
        sweep(std::integral_constant<int,0>{}); sweep(std::integral_constant<int,1>{}); sweep(std::integral_constant<int,2>{});
    }
    if(parts & 2)
    {
        // L^T x = w, right-looking from the end: x_j = z_j / L_jj ; z_i -= L[j][i] x_j for i < j
        double row[3];
        const int last = ((n-1)*n) >> 1;
#pragma unroll
        for(int k = 0; k < 3; k++) { const int i = lane + 64*k; const double v = Lp[last + min(i, n - 1)]; row[k] = (i < n - 1) ? v : 0.0; }
        double rdj = rds[n-1];
        auto sweep = [&](auto KS)
        {
            constexpr int ks = decltype(KS)::value;
            for(int j = min(n, 64*ks + 64) - 1; j >= 64*ks; j--)
            {
                const double xj = readlane_f64(z[ks], j & 63)*rdj;
                double nxt[3];
                const int jm = max(j - 1, 0), rowm = (jm*(jm+1)) >> 1;
#pragma unroll
                for(int k = 0; k < 3; k++)
                {
                    const int i = lane + 64*k;
                    const double v = Lp[rowm + min(i, jm)];
                    nxt[k] = (j >= 1 && i < j - 1) ? v : 0.0;
                }
                const double rdn = rds[jm];
#pragma unroll
                for(int k = 0; k < 3; k++)
                {
                    const int i = lane + 64*k;
                    const double upd = fma(-row[k], xj, z[k]);
                    z[k] = (i == j) ? xj : (i < j) ? upd : z[k];
                    row[k] = nxt[k];
                }
                rdj = rdn;
            }
        };
        sweep(std::integral_constant<int,2>{}); sweep(std::integral_constant<int,1>{}); sweep(std::integral_constant<int,0>{});
    }
#pragma unroll
    for(int k = 0; k < 3; k++) { const int i = lane + 64*k; if(i < n) r[i] = z[k]; }
}
// The same for a big camera block (splined models: n = 1206), by blocks of 64 columns, r in LDS. A block is
// (a) its 64 x 64 diagonal triangle into LDS (row stride 65: a column read is conflict-free), (b) one wave
// solving it with r in registers as above, (c) the whole workgroup subtracting the block's part from the rows
// (forward) / columns (backward) that remain - reads of S that are row segments either way: going forward 16
// lanes share a row's 64 entries, going back a thread owns a column and half of the block's rows.
// fsolve_dense_kernel pays two barriers and a memory round trip per COLUMN: 1870 us at n = 1206
#define FSB_LD 65
__global__ __launch_bounds__(1024)
void fsolve_dense_blocked_kernel(int n, const double* __restrict__ S, double* __restrict__ r, int parts)
{
    extern __shared__ __attribute__((aligned(16))) double fsb_lds[];
    const int npad = (n + 63) & ~63, nblocks = npad >> 6;
    double* __restrict__ rs   = fsb_lds;                 // npad
    double* __restrict__ Ld   = rs + npad;               // 64 x 65
    double* __restrict__ zs   = Ld + 64*FSB_LD;          // 64
    double* __restrict__ part = zs + 64;                 // 1024
    r += (size_t)blockIdx.x*n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for(int i = t; i < npad; i += 1024) rs[i] = (i < n) ? r[i] : 0.0;
    // rows past n are rows of the identity: their entries of r are zero and stay zero
    auto load_diag = [&](int j0)
    {
#pragma unroll
        for(int e = t; e < 64*64; e += 1024)
        {
            const int i = e >> 6, k = e & 63;
            const bool ok = (j0 + i < n) && (k <= i);
            const double v = S[ok ? (size_t)(j0 + i)*n + j0 + k : 0];
            Ld[i*FSB_LD + k] = ok ? v : (i == k ? 1.0 : 0.0);
        }
    };
    if(parts & 1)
    for(int b = 0; b < nblocks; b++)
    {
        const int j0 = b << 6;
        load_diag(j0);
        __syncthreads();
        if(wave == 0)
        {
            double z = rs[j0 + lane];
            const double inv = 1.0/Ld[lane*FSB_LD + lane];
            double m = Ld[lane*FSB_LD];
            for(int k = 0; k < 64; k++)
            {
                const double mnext = Ld[lane*FSB_LD + ((k + 1) & 63)];
                const double zk = readlane_f64(z, k)*readlane_f64(inv, k);
                z = (lane == k) ? zk : ((lane > k) ? fma(-m, zk, z) : z);
                m = mnext;
            }
            rs[j0 + lane] = z; zs[lane] = z;
        }
        __syncthreads();
        {
            const int sub = t & 15;
            const double z0 = zs[4*sub], z1 = zs[4*sub+1], z2 = zs[4*sub+2], z3 = zs[4*sub+3];
#pragma unroll 4
            for(int i = j0 + 64 + (t >> 4); i < n; i += 64)
            {
                const double* __restrict__ p = S + (size_t)i*n + j0 + 4*sub;
                double a = (p[0]*z0 + p[1]*z1) + (p[2]*z2 + p[3]*z3);
                a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
                if(sub == 0) rs[i] -= a;
            }
        }
        __syncthreads();
    }
    if(parts & 2)
    for(int b = nblocks - 1; b >= 0; b--)
    {
        const int j0 = b << 6;
        load_diag(j0);
        __syncthreads();
        if(wave == 0)
        {
            double z = rs[j0 + lane];
            const double inv = 1.0/Ld[lane*FSB_LD + lane];
            double m = Ld[63*FSB_LD + lane];
            for(int k = 63; k >= 0; k--)
            {
                const double mnext = Ld[((k - 1) & 63)*FSB_LD + lane];
                const double zk = readlane_f64(z, k)*readlane_f64(inv, k);
                z = (lane == k) ? zk : ((lane < k) ? fma(-m, zk, z) : z);
                m = mnext;
            }
            rs[j0 + lane] = z; zs[lane] = z;
        }
        __syncthreads();
        // columns i < j0: two threads per column, 32 of the block's rows each
        const int kmax = min(64, n - j0);
        const int g = t >> 9, w = t & 511;
        for(int i0 = 0; i0 < j0; i0 += 512)
        {
            const int i = i0 + w;
            double a0 = 0.0, a1 = 0.0;
            if(i < j0)
            {
                const double* __restrict__ p = S + (size_t)(j0 + 32*g)*n + i;
#pragma unroll
                for(int k = 0; k < 32; k += 2)
                {
                    const int k0 = 32*g + k;
                    if(k0     < kmax) a0 = fma(p[(size_t)k*n],       zs[k0],     a0);
                    if(k0 + 1 < kmax) a1 = fma(p[(size_t)(k + 1)*n], zs[k0 + 1], a1);
                }
            }
            part[t] = a0 + a1;
            __syncthreads();
            if(g == 0 && i < j0) rs[i] -= part[w] + part[512 + w];
            __syncthreads();
        }
    }
    for(int i = t; i < n; i += 1024) r[i] = rs[i];
}
static inline size_t fsolve_blocked_lds_bytes(int n) { return (size_t)(((n + 63) & ~63) + 64*FSB_LD + 64 + 1024)*sizeof(double); }
// x_e = L_e^-T (y_e - Wt_e x_S), one workgroup per E block; the extra block copies x_S
__global__ __launch_bounds__(64)
void fsolve_backsub_kernel(NormalDims nd, const double* __restrict__ Wt, const double* __restrict__ LD,
                           const double* __restrict__ y, const double* __restrict__ xs,
                           double* __restrict__ x, int order, size_t sb)
{
    y += (size_t)blockIdx.z*nd.NE; xs += (size_t)blockIdx.z*nd.Nc; x += blockIdx.z*sb;
    const int t = threadIdx.x;
    if((int)blockIdx.x == nd.NEb)
    {
        for(int i=t;i<nd.Nc;i+=blockDim.x) x[fs_index_S(nd, order, i)] = xs[i];
        return;
    }
    const int blk = blockIdx.x;
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    __shared__ double red[6][64];
    double part[6] = {0,0,0,0,0,0};
    for(int c=t;c<nd.Nc;c+=blockDim.x)
    {
        const double d = xs[c];
        for(int i=0;i<de;i++) part[i] += Wt[(size_t)(e0+i)*nd.Nc + c]*d;
    }
    for(int i=0;i<6;i++) red[i][t] = part[i];
    __syncthreads();
    if(t == 0)
    {
        double v[6];
        const double* L = LD + (size_t)blk*36;
        for(int i=0;i<de;i++)
        {
            double s = y[e0+i];
            for(int k=0;k<64;k++) s -= red[i][k];
            v[i] = s;
        }
        for(int i=de-1;i>=0;i--)
        {
            double s = v[i];
            for(int k=i+1;k<de;k++) s -= L[k*6+i]*v[k];
            v[i] = s/L[i*6+i];
        }
        for(int i=0;i<de;i++) x[fs_index_E(nd, order, e0 + i)] = v[i];
    }
}
// min and max over the diagonal of the whole factor: out[0] = min, out[1] = max
__global__ __launch_bounds__(256)
void fsolve_diag_minmax_kernel(NormalDims nd, const double* __restrict__ S, const double* __restrict__ LD,
                               double* __restrict__ out)
{
    double mn = 1e300, mx = 0.0;
    const int total = nd.Nc + nd.NE;
    for(int i = blockIdx.x*blockDim.x + threadIdx.x; i < total; i += gridDim.x*blockDim.x)
    {
        double d;
        if(i < nd.Nc) d = S[(size_t)i*nd.Nc + i];
        else
        {
            int blk, a, de, e0;
            E_to_block(nd, i - nd.Nc, &blk, &a, &de, &e0);
            d = LD[(size_t)blk*36 + a*6 + a];
        }
        mn = fmin(mn, d); mx = fmax(mx, d);
    }
    for(int off=32; off>0; off>>=1) { mn = fmin(mn, __shfl_down(mn, off)); mx = fmax(mx, __shfl_down(mx, off)); }
    if((threadIdx.x & 63) == 0)
    {
        // positive doubles order like their bit patterns
        atomicMin((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(mn));
        atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(mx));
    }
}

// y = b_E, r = b_S  /  x = [y ; r]
__global__ __launch_bounds__(256)
void fsolve_split_kernel(NormalDims nd, const double* __restrict__ b, double* __restrict__ y, double* __restrict__ r, int order, size_t sb)
{
    b += blockIdx.z*sb; y += (size_t)blockIdx.z*nd.NE; r += (size_t)blockIdx.z*nd.Nc;
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i < nd.NE) y[i] = b[fs_index_E(nd, order, i)];
    else if(i < nd.NE + nd.Nc) r[i - nd.NE] = b[fs_index_S(nd, order, i - nd.NE)];
}
__global__ __launch_bounds__(256)
void fsolve_join_kernel(NormalDims nd, const double* __restrict__ y, const double* __restrict__ r, double* __restrict__ x, int order, size_t sb)
{
    x += blockIdx.z*sb; y += (size_t)blockIdx.z*nd.NE; r += (size_t)blockIdx.z*nd.Nc;
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i < nd.NE) x[fs_index_E(nd, order, i)] = y[i];
    else if(i < nd.NE + nd.Nc) x[fs_index_S(nd, order, i - nd.NE)] = r[i - nd.NE];
}
// to_factor: x (factor order) = P b (state order); else x (state order) = P^T b (factor order)
__global__ __launch_bounds__(256)
void fsolve_permute_kernel(NormalDims nd, const double* __restrict__ b, double* __restrict__ x, int to_factor, size_t sb)
{
    b += blockIdx.z*sb; x += blockIdx.z*sb;
    const int i = blockIdx.x*blockDim.x + threadIdx.x;      // index in factor order
    if(i >= nd.NE + nd.Nc) return;
    const int is = (i < nd.NE) ? fs_index_E(nd, 0, i) : fs_index_S(nd, 0, i - nd.NE);
    if(to_factor) x[i] = b[is]; else x[is] = b[i];
}

// ---- the consumers of J that mrcal's projection uncertainty uses (mrcal-genpywrap.py:477-731), on the
// device-resident CSR J of a factorization
// y = Jt x without atomics (round 4: the same bits every time, like the solve). The rows are cut into chunks of a
// fixed number of rows (a function of the matrix's shape alone); ONE wave walks a chunk's rows in order, a lane per
// entry of the row, adding into the chunk's own copy of y in LDS with LDS atomics - the columns of one row are distinct in every Jacobian
// the problems make, so a wave instruction adds to an address once (a caller's row that repeats a column is served too:
// the LDS takes an instruction's adds in lane order), and consecutive rows are consecutive instructions of the same
// wave: the order of every sum is the row order. The chunks' copies go to part[chunk][.] and csr_Jt_x_sum_kernel adds them per
// column in chunk order. A y longer than the LDS tile is done in column tiles (a pass over the chunk's rows each).
// (History: one lane per row with atomics, 41 ms at the metric's size; rows of a half-wave with the same columns summed
//  first, then atomics, 2.1 ms; this - see profiles/r04_*)
#define JTX_TILE 7680          // doubles of y per pass: 60 KB of LDS
__global__ __launch_bounds__(64)
void csr_Jt_x_chunk_kernel(int Nrows, int Ncols, int rows_per_chunk, const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                           const double* __restrict__ Jx, const double* __restrict__ x, double* __restrict__ part)
{
    __shared__ double acc[JTX_TILE];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x*rows_per_chunk, r1 = min(Nrows, r0 + rows_per_chunk);
    double* __restrict__ out = part + (size_t)blockIdx.x*Ncols;
    for(int c0 = 0; c0 < Ncols; c0 += JTX_TILE)
    {
        const int nc = min(JTX_TILE, Ncols - c0);
        for(int i = lane; i < nc; i += 64) acc[i] = 0.0;
        __builtin_amdgcn_wave_barrier();
        for(int r = r0; r < r1; r++)
        {
            const double xr = x[r];
            if(xr == 0.0) continue;                         // (wave-uniform; outlier rows are all zero)
            const int p0 = Jp[r], p1 = Jp[r+1];
            for(int p = p0 + lane; p < p1; p += 64)
            {
                const int c = Ji[p] - c0;
                // (ds_add_f64, not a read-modify-write: a CSR row may list a column twice - scipy allows it, the
                //  reference's loop adds both - and two lanes of one instruction then meet at one address: the LDS
                //  applies the adds of an instruction one after the other, lane by lane. With distinct columns, the
                //  only case the problems' own Jacobians have, it is the same a + v as before)
                if(c >= 0 && c < nc) atomicAdd(&acc[c], Jx[p]*xr);
            }
            // (a row longer than 64 entries: its later entries are later instructions; LDS serves a wave in order)
        }
        __builtin_amdgcn_wave_barrier();
        for(int i = lane; i < nc; i += 64) out[c0 + i] = acc[i];
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(256)
void csr_Jt_x_sum_kernel(int Ncols, int Nchunks, const double* __restrict__ part, double* __restrict__ y)
{
    const int c = blockIdx.x*blockDim.x + threadIdx.x;
    if(c >= Ncols) return;
    double a = 0.0;
    for(int k = 0; k < Nchunks; k++) a += part[(size_t)k*Ncols + c];
    y[c] = a;
}
// how many chunks / rows per chunk for a matrix of this shape (and nothing else: the summation order must not
// depend on the device or on the day)
int csr_Jt_x_chunks(int Nrows, int* rows_per_chunk)
{
    int rpc = (Nrows + 1023)/1024;
    if(rpc < 512) rpc = 512;
    *rows_per_chunk = rpc;
    return (Nrows + rpc - 1)/rpc;
}
size_t csr_Jt_x_scratch_doubles(int Nrows, int Ncols)
{
    int rpc;
    return (size_t)csr_Jt_x_chunks(Nrows, &rpc)*(size_t)(Ncols > 0 ? Ncols : 1);
}
// out (NX x NX) += sum over the leading rows of outer(A j, A j), A (NX x Nstate) row-major
template<int NX>
__global__ __launch_bounds__(256)
void csr_A_Jt_J_At_kernel(int Nrows, int Nstate, const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                          const double* __restrict__ Jx, const double* __restrict__ A, double* __restrict__ out)
{
    const int r = blockIdx.x*blockDim.x + threadIdx.x;
    double jta[NX];
#pragma unroll
    for(int i=0;i<NX;i++) jta[i] = 0.0;
    if(r < Nrows)
        for(int32_t p = Jp[r]; p < Jp[r+1]; p++)
        {
            const int32_t c = Ji[p];
            const double  v = Jx[p];
#pragma unroll
            for(int i=0;i<NX;i++) jta[i] += A[(size_t)i*Nstate + c]*v;
        }
    __shared__ double part[4][NX*NX];
#pragma unroll
    for(int i=0;i<NX;i++)
#pragma unroll
        for(int j=0;j<NX;j++)
        {
            double v = jta[i]*jta[j];
            for(int off=32; off>0; off>>=1) v += __shfl_down(v, off);
            if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][i*NX + j] = v;
        }
    __syncthreads();
    // (per-workgroup partials, added in workgroup order by csr_A_Jt_J_At_sum_kernel: no atomics, the same bits every time)
    if(threadIdx.x < NX*NX)
        out[(size_t)blockIdx.x*64 + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(64)
void csr_A_Jt_J_At_sum_kernel(int n, int Nblocks, const double* __restrict__ part, double* __restrict__ out)
{
    if((int)threadIdx.x >= n) return;
    double a = 0.0;
    for(int k = 0; k < Nblocks; k++) a += part[(size_t)k*64 + threadIdx.x];
    out[threadIdx.x] = a;
}
// scratch: csr_Jt_x_scratch_doubles(Nrows, Ncols) doubles. y need not be cleared
hipError_t launch_csr_Jt_x(int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji, const double* Jx, const double* x, double* y,
                           double* scratch, hipStream_t stream)
{
    if(Ncols <= 0) return hipSuccess;
    if(Nrows <= 0) return hipMemsetAsync(y, 0, (size_t)Ncols*sizeof(double), stream);
    int rpc;
    const int nchunks = csr_Jt_x_chunks(Nrows, &rpc);
    hipLaunchKernelGGL(csr_Jt_x_chunk_kernel, dim3(nchunks), dim3(64), 0, stream, Nrows, Ncols, rpc, Jp, Ji, Jx, x, scratch);
    hipLaunchKernelGGL(csr_Jt_x_sum_kernel, dim3((Ncols + 255)/256), dim3(256), 0, stream, Ncols, nchunks, scratch, y);
    return hipGetLastError();
}
// scratch: 64 doubles per 256 rows ((Nrows + 255)/256 * 64)
hipError_t launch_csr_A_Jt_J_At(int NX, int Nrows, int Nstate, const int32_t* Jp, const int32_t* Ji, const double* Jx,
                                const double* A, double* out, double* scratch, hipStream_t stream)
{
    if(Nrows <= 0) return hipMemsetAsync(out, 0, (size_t)NX*NX*sizeof(double), stream);
    const dim3 g((Nrows + 255)/256), b(256);
    switch(NX)
    {
#define CASE(n) case n: hipLaunchKernelGGL(csr_A_Jt_J_At_kernel<n>, g, b, 0, stream, Nrows, Nstate, Jp, Ji, Jx, A, scratch); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(csr_A_Jt_J_At_sum_kernel, dim3(1), dim3(64), 0, stream, NX*NX, (int)g.x, scratch, out);
    return hipGetLastError();
}

hipError_t launch_fsolve(const NormalDims& nd, const FactorBuffers& F,
                         const double* b, double* x, hipStream_t stream)
{
    return launch_fsolve_sys(nd, F, FSOLVE_A, b, x, stream);
}
// The systems of cholmod_solve2() (same codes) against the kept factorization:
// A x = b in state order; the others in factor order [E | S]: L x = b, L^T x = b,
// L L^T x = b (D is the identity: this is an LL^T factorization, so LD == L,
// DLt == Lt, and D copies); P, Pt permute between state and factor order
hipError_t launch_fsolve_sys(const NormalDims& nd, const FactorBuffers& F, int sys,
                             const double* b, double* x, hipStream_t stream)
{
    // slabs of rows of Wt are summed side by side into F.Spart (free between factorizations)
    return launch_fsolve_sys_batch(nd, F, sys, b, x, 1, F.y, F.r, F.Spart, schur_partial_doubles(nd), stream);
}
// rows of Wt a slab of the r = b_S - Wt^T y sum takes when nrhs right-hand sides are solved side by side
static void fsolve_slabs(const NormalDims& nd, int nrhs, size_t room_per_rhs, int* rows_per_slab, int* nslabs)
{
    // 64 rows whatever the batch: a right-hand side gets the same bits alone and in company
    (void)nrhs;
    int rps = 64;
    int ns  = (nd.NE + rps - 1)/rps;
    const int max_slabs = (nd.Nc > 0) ? (int)std::min<size_t>(room_per_rhs/(size_t)nd.Nc, 4096) : 1;
    if(ns > max_slabs) { ns = max_slabs > 0 ? max_slabs : 1; rps = (nd.NE + ns - 1)/ns; ns = (nd.NE + rps - 1)/rps; }
    *rows_per_slab = rps; *nslabs = ns;
}
size_t fsolve_batch_scratch_doubles(const NormalDims& nd, int nrhs)
{
    int rps, ns;
    fsolve_slabs(nd, nrhs, (size_t)1 << 40, &rps, &ns);
    return (size_t)nd.NE + (size_t)nd.Nc + (size_t)std::max(ns, 1)*(size_t)nd.Nc;
}
// nrhs systems at once: b, x are [nrhs][Nstate]; y [nrhs][NE], r [nrhs][Nc], part [nrhs][part_per_rhs] scratch.
// A right-hand side is its own set of workgroups of every kernel (blockIdx.z; blockIdx.x of the one-workgroup
// triangular solve), so a batch costs about what one costs until the chip is full
hipError_t launch_fsolve_sys_batch(const NormalDims& nd, const FactorBuffers& F, int sys,
                                   const double* b, double* x, int nrhs,
                                   double* y, double* r, double* part, size_t part_per_rhs, hipStream_t stream)
{
    const int n = nd.Nstate;
    if(nrhs <= 0) return hipSuccess;
    if(nrhs > 65535) return hipErrorInvalidValue;
    const unsigned Z = (unsigned)nrhs;
    const size_t sb = (size_t)n;
    if(sys == FSOLVE_D) return hipMemcpyAsync(x, b, (size_t)nrhs*n*sizeof(double), hipMemcpyDeviceToDevice, stream);
    if(sys == FSOLVE_P || sys == FSOLVE_Pt)
    {
        hipLaunchKernelGGL(fsolve_permute_kernel, dim3((n + 255)/256, 1, Z), dim3(256), 0, stream, nd, b, x, sys == FSOLVE_P ? 1 : 0, sb);
        return hipGetLastError();
    }
    const int  order   = (sys == FSOLVE_A) ? 0 : 1;
    const bool forward = (sys == FSOLVE_A || sys == FSOLVE_LDLt || sys == FSOLVE_L  || sys == FSOLVE_LD);
    const bool backwrd = (sys == FSOLVE_A || sys == FSOLVE_LDLt || sys == FSOLVE_Lt || sys == FSOLVE_DLt);
    if(forward)
    {
        if(nd.NEb > 0)
            hipLaunchKernelGGL(fsolve_forward_kernel, dim3((nd.NEb + 63)/64, 1, Z), dim3(64), 0, stream, nd, F.LD, b, y, order, sb);
        int rows_per_slab, nslabs;
        fsolve_slabs(nd, nrhs, part_per_rhs, &rows_per_slab, &nslabs);
        if(nd.NE > 0 && nd.Nc > 0)
            hipLaunchKernelGGL(fsolve_reduce_partial_kernel, dim3((nd.Nc + 255)/256, nslabs, Z), dim3(256), 0, stream,
                               nd, F.Wt, y, part, rows_per_slab);
        else nslabs = 0;
        hipLaunchKernelGGL(fsolve_reduce_kernel, dim3((16*nd.Nc + 255)/256, 1, Z), dim3(256), 0, stream, nd, part, nslabs, b, r, order, sb);
    }
    else
        // y = b_E, r = b_S as they are
        hipLaunchKernelGGL(fsolve_split_kernel, dim3((n + 255)/256, 1, Z), dim3(256), 0, stream, nd, b, y, r, order, sb);
    const int parts = (forward ? 1 : 0) | (backwrd ? 2 : 0);
    if(nd.Nc > 0 && nd.Nc <= 178)
        hipLaunchKernelGGL(fsolve_dense_lds_kernel, dim3(Z), dim3(1024), (size_t)(((nd.Nc*(nd.Nc+1)) >> 1) + 2 + nd.Nc)*sizeof(double), stream,
                           nd.Nc, F.S, r, parts);
    else if(nd.Nc > 0 && fsolve_blocked_lds_bytes(nd.Nc) <= 156*1024)
        hipLaunchKernelGGL(fsolve_dense_blocked_kernel, dim3(Z), dim3(1024), fsolve_blocked_lds_bytes(nd.Nc), stream,
                           nd.Nc, F.S, r, parts);
    else
        hipLaunchKernelGGL(fsolve_dense_kernel, dim3(Z), dim3(1024), 0, stream, nd.Nc, F.S, r, parts);
    if(backwrd)
        hipLaunchKernelGGL(fsolve_backsub_kernel, dim3(nd.NEb + 1, 1, Z), dim3(64), 0, stream, nd, F.Wt, F.LD, y, r, x, order, sb);
    else
        hipLaunchKernelGGL(fsolve_join_kernel, dim3((n + 255)/256, 1, Z), dim3(256), 0, stream, nd, y, r, x, order, sb);
    return hipGetLastError();
}
hipError_t launch_fsolve_diag_minmax(const NormalDims& nd, const FactorBuffers& F, double* out2, hipStream_t stream)
{
    hipLaunchKernelGGL(fsolve_diag_minmax_kernel, dim3(64), dim3(256), 0, stream, nd, F.S, F.LD, out2);
    return hipGetLastError();
}
// normal equations of a bare CSR matrix (every row through the generic path)
//
// Nothing is known about such a matrix but its partition, so every row adds the products of its entries to A, Bt
// and D with atomics - and the sum of doubles in the order the atomics happen to land is not the same twice. It IS
// the same twice when no addition rounds (round 4; the idea of Demmel & Nguyen's pre-rounded reproducible sums):
// with c_i = the binary exponent above the largest |entry| of column i (a pass of integer atomicMax: any order, the
// same result), a product t of columns i, j is below 2^(c_i + c_j), and of the n < 2^(N-1) products that can meet in
// one place
//     q1 = t rounded to a multiple of u1 = 2^(c_i + c_j + N - 52)       ((t + 1.5 2^52 u1) - 1.5 2^52 u1, exactly)
// sum to less than 2^52 u1: every partial sum is a multiple of u1 with 52 bits or fewer, no addition rounds, any
// order gives the same double. The remainder r1 = t - q1 is exact and at most u1/2; it is split the same way one
// level down, and that one's remainder once more: three accumulators per entry, what is dropped below
// 2^(c_i + c_j + 3N - 159) a product (N = 23: 2^-90 of the largest product that can occur there). The entry is
// (s1 + s2) + s3. Three times the atomics of the plain row-by-row assembly (rows_generic_kernel), and the same bits
// every time: what a CHOLMOD_factorization(J) made from a bare matrix is built from.
__global__ __launch_bounds__(256)
void csr_column_max_kernel(long long Nnz, int Nstate, const int32_t* __restrict__ Ji, const double* __restrict__ Jv,
                           unsigned long long* __restrict__ cmax /* [Nstate], zeroed: the bits of the largest |value| */)
{
    for(long long p = (long long)blockIdx.x*blockDim.x + threadIdx.x; p < Nnz; p += (long long)gridDim.x*blockDim.x)
    {
        const int c = Ji[p];
        if((unsigned)c >= (unsigned)Nstate) continue;
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(Jv[p]));
        if(b > cmax[c]) atomicMax(&cmax[c], b);      // (monotone in |value|; NaN ends up largest and poisons the sums, as it should)
    }
}
// 64 consecutive rows a wave; runs of rows with the same columns (a board observation's x rows, its y rows) are summed
// across a half-wave first and ONE lane adds the sum - rows_generic_wave<true>: a thirtieth of the atomics
__global__ __launch_bounds__(64)
void rows_repro_kernel(NormalDims nd, OpRef R, int row0, int row1, const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                       const unsigned long long* __restrict__ cmax, int N, ReproAcc a0, ReproAcc a1, ReproAcc a2)
{
    const ReproCtx rc = { { a0, a1, a2 }, cmax, N };
    rows_generic_wave<true>(nd, opref_get(R), row0 + blockIdx.x*blockDim.x, row1, Jp, Ji, &rc);
}
// entry = (level 1 + level 2) + level 3. sym > 0: the array is made of sym x sym blocks of which the lower triangles
// were summed; the upper ones are their mirror images
__global__ __launch_bounds__(256)
void repro_combine_kernel(size_t n, int sym, double* __restrict__ s1, const double* __restrict__ s2, const double* __restrict__ s3)
{
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x)
    {
        size_t j = i;
        if(sym > 0)
        {
            const size_t blk = i / ((size_t)sym*sym), e = i - blk*(size_t)sym*sym;
            const size_t r = e / sym, c = e - r*sym;
            if(c > r) continue;                 // (written by its mirror image's thread)
            j = blk*(size_t)sym*sym + c*sym + r;
        }
        const double v = (s1[i] + s2[i]) + s3[i];
        s1[i] = v;
        if(j != i) s1[j] = v;
    }
}
size_t assemble_rows_scratch_doubles(const NormalDims& nd)
{
    const size_t one = (size_t)nd.Nc*nd.Nc + (size_t)nd.NE*nd.Nc + (size_t)nd.NEb*36;
    return 2*one + (size_t)nd.Nstate + 64;
}
// scratch: assemble_rows_scratch_doubles(nd) doubles (the second and third levels of A, Bt, D; the columns' maxima), or
// NULL: the plain row-by-row assembly, whose sums depend on the order the atomics land in. x is not looked at with a
// scratch (a bare matrix has none: g and |x|^2 stay zero)
hipError_t launch_assemble_rows(const NormalDims& nd, const OpRef& R, int Nmeas,
                                const int32_t* Jp, const int32_t* Ji, hipStream_t stream, double* scratch, long long Nnz)
{
    {
        const size_t total = (size_t)nd.Nc*nd.Nc + (size_t)nd.NE*nd.Nc + (size_t)nd.NEb*36 + nd.Nstate + NSCALARS;
        int nb = (int)((total + 255)/256); if(nb > 2048) nb = 2048;
        hipLaunchKernelGGL(zero_normal_kernel, dim3(nb), dim3(256), 0, stream, nd, R);
    }
    if(Nmeas <= 0) return hipGetLastError();
    if(scratch == NULL || R.sel != NULL)
    {
        hipLaunchKernelGGL(rows_generic_kernel, dim3((Nmeas + 63)/64), dim3(64), 0, stream,
                           nd, R, 0, Nmeas, Jp, Ji);
        return hipGetLastError();
    }
    const size_t nA = (size_t)nd.Nc*nd.Nc, nB = (size_t)nd.NE*nd.Nc, nD = (size_t)nd.NEb*36, one = nA + nB + nD;
    hipError_t e = hipMemsetAsync(scratch, 0, assemble_rows_scratch_doubles(nd)*sizeof(double), stream);
    if(e != hipSuccess) return e;
    // (R.sel == NULL: the operating point's pointers are the host's to read)
    OpDev O;
    e = hipMemcpyAsync(&O, R.ops, sizeof(OpDev), hipMemcpyDeviceToHost, stream);   if(e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);                                               if(e != hipSuccess) return e;
    unsigned long long* cmax = (unsigned long long*)(scratch + 2*one);
    // Nmeas <= 2^(N-3): the addends that meet in one place - a run's sum, a row's product; two of them for a row that lists
    // a column twice - stay below 2^(N-1)
    int N = 3; while(((long long)1 << (N - 3)) < (long long)Nmeas) N++;
    {
        long long nb = (Nnz + 255)/256; if(nb > 4096) nb = 4096; if(nb < 1) nb = 1;
        hipLaunchKernelGGL(csr_column_max_kernel, dim3((int)nb), dim3(256), 0, stream, Nnz, nd.Nstate, Ji, O.Jv, cmax);
    }
    const ReproAcc a0 = { O.A, O.Bt, O.D, NULL, NULL };
    const ReproAcc a1 = { scratch, scratch + nA, scratch + nA + nB, NULL, NULL };
    const ReproAcc a2 = { scratch + one, scratch + one + nA, scratch + one + nA + nB, NULL, NULL };
    hipLaunchKernelGGL(rows_repro_kernel, dim3((Nmeas + 63)/64), dim3(64), 0, stream, nd, R, 0, Nmeas, Jp, Ji, cmax, N, a0, a1, a2);
    if(nA > 0) hipLaunchKernelGGL(repro_combine_kernel, dim3((int)std::min<size_t>(2048, (nA + 255)/256)), dim3(256), 0, stream, nA, nd.Nc, O.A,  a1.A,  a2.A);
    if(nB > 0) hipLaunchKernelGGL(repro_combine_kernel, dim3((int)std::min<size_t>(2048, (nB + 255)/256)), dim3(256), 0, stream, nB, 0,     O.Bt, a1.Bt, a2.Bt);
    if(nD > 0) hipLaunchKernelGGL(repro_combine_kernel, dim3((int)std::min<size_t>(2048, (nD + 255)/256)), dim3(256), 0, stream, nD, 6,     O.D,  a1.D,  a2.D);
    return hipGetLastError();
}

} // namespace mrcal_amd
