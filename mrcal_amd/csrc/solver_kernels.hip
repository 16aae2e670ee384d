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
#include "problem.hpp"
#include "solver_kernels.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// index helpers
////////////////////////////////////////////////////////////////////////////////
// state index -> S index (>=0) or -(1 + E index)
__device__ __forceinline__ int state_to_SE(const NormalDims& nd, int col)
{
    if(col < nd.Nie) return col;
    if(nd.Nwarp && col >= nd.i_state_warp) return nd.Nie + (col - nd.i_state_warp);
    return -(1 + (col - nd.Nie));
}
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

// The assembly kernels walk the "compact" columns of a board observation: the
// columns that exist in the state, in state order, then the residual. Compact
// column j -> state index (-1 for the residual column)
__device__ __forceinline__ int board_tile_col_to_state(const DeviceProblem& P, const BoardObsMeta& m, int j)
{
    if(j < P.Nintr_state) return m.i_state_intrinsics + j;
    j -= P.Nintr_state;
    if(P.do_optimize_extrinsics && m.icam_extrinsics >= 0)
    {
        if(j < 6) return m.i_state_extrinsics + j;
        j -= 6;
    }
    if(P.do_optimize_frames)
    {
        if(j < 6) return m.i_state_frame + j;
        j -= 6;
    }
    if(P.has_warp_state)
    {
        if(j < 2) return P.i_state_warp + j;
        j -= 2;
    }
    return -1;
}
// compact column j -> column of the board kernel's fixed tile layout (problem.hpp)
__device__ __forceinline__ int board_compact_to_tile_col(const DeviceProblem& P, const BoardObsMeta& m, int j)
{
    if(j < P.Ncore_state) return j;
    j -= P.Ncore_state;
    if(j < P.Ndist_state) return 4 + j;
    j -= P.Ndist_state;
    if(P.do_optimize_extrinsics && m.icam_extrinsics >= 0)
    {
        if(j < 6) return tile_ext0(P.Ndist) + j;
        j -= 6;
    }
    if(P.do_optimize_frames)
    {
        if(j < 6) return tile_frame0(P.Ndist) + j;
        j -= 6;
    }
    if(P.has_warp_state)
    {
        if(j < 2) return tile_warp0(P.Ndist) + j;
        j -= 2;
    }
    return tile_xcol(P.Ndist);
}
__device__ __forceinline__ int board_tile_ncols(const DeviceProblem& P, const BoardObsMeta& m)
{
    return P.Nintr_state +
        ((P.do_optimize_extrinsics && m.icam_extrinsics >= 0) ? 6 : 0) +
        (P.do_optimize_frames ? 6 : 0) +
        (P.has_warp_state ? 2 : 0) + 1;
}
// G[i][j] by compact columns
__device__ __forceinline__ double board_gram(const DeviceProblem& P, const BoardObsMeta& m,
                                             const double* __restrict__ G, int i, int j)
{
    return gram_get(G, tile_nblk(P.Ndist),
                    board_compact_to_tile_col(P, m, i), board_compact_to_tile_col(P, m, j));
}

////////////////////////////////////////////////////////////////////////////////
// assembly from the per-observation Grams
////////////////////////////////////////////////////////////////////////////////

// One workgroup per frame. Its observations are contiguous (the API requires
// frame-sorted observations, mrcal-pywrap.c:1063-1138). Accumulates, in
// observation order and without atomics:
//   D_f  += G[frame,frame]      g_f += G[frame,x]     Bt[frame rows][S cols] += G[S,frame]
__global__ __launch_bounds__(256)
void assemble_frames_kernel(DeviceProblem P, NormalDims nd,
                            const int* __restrict__ frame_obs_begin, // [Nframes+1], local obs indices
                            const double* __restrict__ gram,
                            double* __restrict__ Bt, double* __restrict__ D, double* __restrict__ g)
{
    const int f = blockIdx.x;
    const int t = threadIdx.x;
    const int o0 = frame_obs_begin[f], o1 = frame_obs_begin[f+1];
    if(o0 >= o1) return;
    const int e0 = 6*f;   // frame blocks come first in E

    for(int o = o0; o < o1; o++)
    {
        const BoardObsMeta m = P.board_meta[o];
        const double* __restrict__ G = gram + (size_t)o*gram_stride(P.Ndist);
        const int ncols = board_tile_ncols(P, m);
        // tile column of the frame block
        const int jf = P.Nintr_state + ((P.do_optimize_extrinsics && m.icam_extrinsics >= 0) ? 6 : 0);
        // D and g: 36 + 6 entries
        if(t < 36)
        {
            const int a = t/6, c = t - 6*a;
            D[(size_t)f*36 + t] += board_gram(P, m, G, jf+a, jf+c);
        }
        else if(t < 42)
        {
            const int a = t - 36;
            g[nd.Nie + e0 + a] += board_gram(P, m, G, jf+a, ncols-1);
        }
        // Bt: (ncols-1-6) S columns x 6
        const int nS = ncols - 1 - 6;
        for(int idx = t; idx < nS*6; idx += blockDim.x)
        {
            const int js = idx/6, a = idx - 6*js;
            const int j  = (js < jf) ? js : js + 6;   // skip the frame columns
            const int s  = state_to_SE(nd, board_tile_col_to_state(P, m, j));
            Bt[(size_t)(e0 + a)*nd.Nc + s] += board_gram(P, m, G, j, jf+a);
        }
        __syncthreads(); // two observations of a frame may share S columns (the warp always)
    }
}

// S-S part: observations that see the same (intrinsics, extrinsics) pair
// scatter to the same entries of A, so they are summed per pair first. One
// workgroup per chunk of one pair's observation list
__global__ __launch_bounds__(256)
void reduce_pairs_kernel(DeviceProblem P, NormalDims nd,
                         const int* __restrict__ chunk_begin,  // [Nchunks+1] into pair_obs
                         const int* __restrict__ pair_obs,     // observation indices grouped by pair
                         const double* __restrict__ gram,
                         double* __restrict__ A, double* __restrict__ g, double* __restrict__ norm2_x)
{
    const int c0 = chunk_begin[blockIdx.x], c1 = chunk_begin[blockIdx.x+1];
    if(c0 >= c1) return;
    const BoardObsMeta m0 = P.board_meta[pair_obs[c0]];
    const int ncols = board_tile_ncols(P, m0);
    const int jf    = P.Nintr_state + ((P.do_optimize_extrinsics && m0.icam_extrinsics >= 0) ? 6 : 0);
    const int nfr   = P.do_optimize_frames ? 6 : 0;
    const int nS1   = ncols - nfr;     // S columns + the residual column
    // upper triangle incl. the residual column: entries (i<=j)
    const int nent  = nS1*(nS1+1)/2;
    for(int idx = threadIdx.x; idx < nent; idx += blockDim.x)
    {
        // unrank idx -> (i<=j) over an nS1 x nS1 upper triangle, row-major
        int i = 0, rem = idx;
        while(rem >= nS1 - i) { rem -= nS1 - i; i++; }
        const int j  = i + rem;
        const int ti = (i < jf) ? i : i + nfr;
        const int tj = (j < jf) ? j : j + nfr;
        double acc = 0.0;
        for(int c = c0; c < c1; c++)
            acc += board_gram(P, m0, gram + (size_t)pair_obs[c]*gram_stride(P.Ndist), ti, tj);

        const int si = board_tile_col_to_state(P, m0, ti);
        const int sj = board_tile_col_to_state(P, m0, tj);
        if(sj < 0)
        {
            if(si < 0) atomicAdd(norm2_x, acc);
            else       atomicAdd(&g[si], acc);
        }
        else
        {
            const int a = state_to_SE(nd, si), bb = state_to_SE(nd, sj);
            atomicAdd(&A[(size_t)a*nd.Nc + bb], acc);
            if(a != bb) atomicAdd(&A[(size_t)bb*nd.Nc + a], acc);
        }
    }
}

// Rows that do not come from board observations (discrete points,
// regularization): one lane per CSR row, scattered with atomics. These are few
__global__ __launch_bounds__(64)
void rows_generic_kernel(NormalDims nd, int row0, int row1,
                         const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                         const double* __restrict__ Jv, const double* __restrict__ x,
                         double* __restrict__ A, double* __restrict__ Bt, double* __restrict__ D,
                         double* __restrict__ g, double* __restrict__ norm2_x)
{
    const int r = row0 + blockIdx.x*blockDim.x + threadIdx.x;
    if(r >= row1) return;
    const int p0 = Jp[r], p1 = Jp[r+1];
    const double xr = x[r];
    atomicAdd(norm2_x, xr*xr);
    for(int p = p0; p < p1; p++)
    {
        const int    ci = Ji[p];
        const double vi = Jv[p];
        atomicAdd(&g[ci], vi*xr);
        const int si = state_to_SE(nd, ci);
        for(int q = p0; q < p1; q++)
        {
            const int    cj = Ji[q];
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
                // bi != bj cannot happen: no row touches two E blocks
            }
        }
    }
}

////////////////////////////////////////////////////////////////////////////////
// Schur complement
////////////////////////////////////////////////////////////////////////////////

// One workgroup per E block: L L^T = D_e + lambda I;  Wt_e = L^-1 Bt_e;  y_e = L^-1 g_e.
// status[0] is set to 1 if any block is not positive definite
__global__ __launch_bounds__(64)
void eblock_factor_kernel(NormalDims nd, BlockRanges br, double lambda,
                          const double* __restrict__ Bt, const double* __restrict__ D,
                          const double* __restrict__ g,
                          double* __restrict__ Wt, double* __restrict__ LD, double* __restrict__ y,
                          int* __restrict__ status)
{
    __shared__ double L[36];
    const int blk = br.block(blockIdx.x);
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    const int t   = threadIdx.x;

    if(t < 36) L[t] = D[(size_t)blk*36 + t] + (((t/6) == (t%6)) ? lambda : 0.0);
    __syncthreads();
    if(t == 0)
    {
        bool ok = true;
        for(int j=0;j<de;j++)
        {
            double d = L[j*6+j];
            for(int k=0;k<j;k++) d -= L[j*6+k]*L[j*6+k];
            if(!(d > 0.0)) { ok = false; d = 1.0; }
            d = sqrt(d);
            L[j*6+j] = d;
            for(int i=j+1;i<de;i++)
            {
                double v = L[i*6+j];
                for(int k=0;k<j;k++) v -= L[i*6+k]*L[j*6+k];
                L[i*6+j] = v/d;
            }
        }
        if(!ok) atomicExch(status, 1);
    }
    __syncthreads();
    if(t < 36) LD[(size_t)blk*36 + t] = L[t];

    // forward substitution, one column of Bt_e per lane; column Nc is g_e
    for(int c = t; c <= nd.Nc; c += blockDim.x)
    {
        double w[6];
        for(int i=0;i<de;i++)
        {
            double v = (c < nd.Nc) ? Bt[(size_t)(e0+i)*nd.Nc + c] : g[nd.Nie + e0 + i];
            for(int k=0;k<i;k++) v -= L[i*6+k]*w[k];
            w[i] = v / L[i*6+i];
        }
        if(c < nd.Nc) for(int i=0;i<de;i++) Wt[(size_t)(e0+i)*nd.Nc + c] = w[i];
        else          for(int i=0;i<de;i++) y[e0+i] = w[i];
    }
}

// S = A + lambda I ;  r = g_S.  The SYRK then subtracts Wt^T Wt and Wt^T y
__global__ __launch_bounds__(256)
void schur_init_kernel(NormalDims nd, double lambda, int with_rhs,
                       const double* __restrict__ A, const double* __restrict__ g,
                       double* __restrict__ S, double* __restrict__ r)
{
    const size_t idx = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    const size_t n2  = (size_t)nd.Nc*nd.Nc;
    if(idx < n2)
    {
        const int i = (int)(idx / nd.Nc), j = (int)(idx - (size_t)i*nd.Nc);
        S[idx] = A[idx] + ((i==j) ? lambda : 0.0);
    }
    if(idx < (size_t)nd.Nc)
    {
        const int i = (int)idx;
        r[i] = with_rhs ? g[(i < nd.Nie) ? i : nd.i_state_warp + (i - nd.Nie)] : 0.0;
    }
}

// S -= Wt^T Wt ,  r -= Wt^T y.   Tile (32 x 32 of S) x (slice of E rows) per
// workgroup; partial products are added atomically. Tiles with bj < bi are
// skipped; the lower triangle is mirrored by the last step of the Cholesky
#define SYRK_TILE 32
__global__ __launch_bounds__(256)
void schur_syrk_kernel(NormalDims nd, int e_lo, int e_hi, int e_per_slice,
                       const double* __restrict__ Wt, const double* __restrict__ y,
                       double* __restrict__ S, double* __restrict__ r)
{
    const int bi = blockIdx.x, bj = blockIdx.y;
    if(bj < bi) return;
    const int e_begin = e_lo + blockIdx.z*e_per_slice;
    const int e_end   = min(e_hi, e_begin + e_per_slice);
    if(e_begin >= e_end) return;

    __shared__ double Wi[16][SYRK_TILE+1];
    __shared__ double Wj[16][SYRK_TILE+1];
    __shared__ double ys[16];

    const int t  = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;     // 16 x 16 threads, 2x2 outputs each
    const int i0 = bi*SYRK_TILE, j0 = bj*SYRK_TILE;
    double acc[2][2] = {{0,0},{0,0}};
    double accr[2]   = {0,0};

    for(int e = e_begin; e < e_end; e += 16)
    {
        // stage 16 E rows x 32 columns of each block
        for(int idx = t; idx < 16*SYRK_TILE; idx += 256)
        {
            const int ee = idx / SYRK_TILE, cc = idx - ee*SYRK_TILE;
            const bool ok = (e + ee < e_end);
            Wi[ee][cc] = (ok && i0+cc < nd.Nc) ? Wt[(size_t)(e+ee)*nd.Nc + i0 + cc] : 0.0;
            Wj[ee][cc] = (ok && j0+cc < nd.Nc) ? Wt[(size_t)(e+ee)*nd.Nc + j0 + cc] : 0.0;
        }
        if(t < 16) ys[t] = (e + t < e_end) ? y[e+t] : 0.0;
        __syncthreads();
#pragma unroll
        for(int ee=0; ee<16; ee++)
        {
            const double a0 = Wi[ee][ty], a1 = Wi[ee][ty+16];
            const double b0 = Wj[ee][tx], b1 = Wj[ee][tx+16];
            acc[0][0] += a0*b0; acc[0][1] += a0*b1;
            acc[1][0] += a1*b0; acc[1][1] += a1*b1;
            if(bj == bi && tx == 0) { accr[0] += a0*ys[ee]; accr[1] += a1*ys[ee]; }
        }
        __syncthreads();
    }
    for(int a=0;a<2;a++)
        for(int bb=0;bb<2;bb++)
        {
            const int i = i0 + ty + 16*a, j = j0 + tx + 16*bb;
            if(i < nd.Nc && j < nd.Nc && j >= i)
                atomicAdd(&S[(size_t)i*nd.Nc + j], -acc[a][bb]);
        }
    if(bj == bi && tx == 0)
        for(int a=0;a<2;a++)
        {
            const int i = i0 + ty + 16*a;
            if(i < nd.Nc) atomicAdd(&r[i], -accr[a]);
        }
}

// Dense Cholesky of S (upper triangle valid on input), one workgroup. On output
// the LOWER triangle of S holds L. Then solves S d = -r in place: r <- d.
// The matrix is staged in LDS when it fits (n <= CHOL_LDS_NMAX), else it is
// factored in place in global memory (correct, slow; large camera blocks are
// a later-round optimization).
__device__ __forceinline__ double& chol_at(double* M, int ld, int i, int j) { return M[(size_t)i*ld + j]; }

template<bool IN_LDS>
__global__ __launch_bounds__(1024)
void schur_cholesky_solve_kernel(int n, double* __restrict__ S, double* __restrict__ r,
                                 int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t  = threadIdx.x;
    const int nt = blockDim.x;
    const int ld = IN_LDS ? (n | 1) : n;
    double* M = IN_LDS ? lds : S;

    // symmetrize into the lower triangle (and into LDS)
    for(int idx = t; idx < n*n; idx += nt)
    {
        const int i = idx / n, j = idx - i*n;
        if(j <= i) chol_at(M, ld, i, j) = S[(size_t)j*n + i];
    }
    __syncthreads();

    __shared__ int notpd;
    if(t == 0) notpd = 0;
    __syncthreads();

    // right-looking, panels of PB columns
    constexpr int PB = 8;
    for(int j0 = 0; j0 < n; j0 += PB)
    {
        const int jb = min(PB, n - j0);
        // factor the panel's diagonal block and the panel below it, column by column.
        // rows are spread over the threads; within a column only the pivot is serial
        for(int jj = 0; jj < jb; jj++)
        {
            const int j = j0 + jj;
            if(t == 0)
            {
                double d = chol_at(M, ld, j, j);
                if(!(d > 0.0)) { notpd = 1; d = 1.0; }
                chol_at(M, ld, j, j) = sqrt(d);
            }
            __syncthreads();
            const double djj = chol_at(M, ld, j, j);
            for(int i = j + 1 + t; i < n; i += nt)
                chol_at(M, ld, i, j) /= djj;
            __syncthreads();
            // update the remaining columns of the panel only
            const int ncols = jb - jj - 1;
            for(int idx = t; idx < ncols*(n - j - 1); idx += nt)
            {
                const int cc = idx % ncols, ii = idx / ncols;
                const int c = j + 1 + cc, i = j + 1 + ii;
                if(i >= c)
                    chol_at(M, ld, i, c) -= chol_at(M, ld, i, j)*chol_at(M, ld, c, j);
            }
            __syncthreads();
        }
        // trailing update: M[i][c] -= sum_k L[i][k] L[c][k], k in the panel; i >= c >= j0+jb
        const int m0 = j0 + jb;
        const int nm = n - m0;
        for(int idx = t; idx < nm*nm; idx += nt)
        {
            const int ii = idx / nm, cc = idx - ii*nm;
            if(cc > ii) continue;
            const int i = m0 + ii, c = m0 + cc;
            double acc = 0.0;
#pragma unroll
            for(int kk = 0; kk < PB; kk++)
                if(kk < jb)
                    acc += chol_at(M, ld, i, j0+kk)*chol_at(M, ld, c, j0+kk);
            chol_at(M, ld, i, c) -= acc;
        }
        __syncthreads();
    }
    if(t == 0 && notpd) atomicExch(status, 1);

    // L z = r ; L^T d = z ; r <- -d.  Column-oriented, the vector lives in LDS
    __shared__ double piv;
    double* v = IN_LDS ? (lds + (size_t)n*ld) : r;
    if(IN_LDS) { for(int i=t;i<n;i+=nt) v[i] = r[i]; __syncthreads(); }
    for(int j=0;j<n;j++)
    {
        if(t == 0) { piv = v[j]/chol_at(M, ld, j, j); v[j] = piv; }
        __syncthreads();
        const double pj = piv;
        for(int i=j+1+t;i<n;i+=nt) v[i] -= chol_at(M, ld, i, j)*pj;
        __syncthreads();
    }
    for(int j=n-1;j>=0;j--)
    {
        if(t == 0) { piv = v[j]/chol_at(M, ld, j, j); v[j] = piv; }
        __syncthreads();
        const double pj = piv;
        for(int i=t;i<j;i+=nt) v[i] -= chol_at(M, ld, j, i)*pj;
        __syncthreads();
    }
    for(int i=t;i<n;i+=nt) r[i] = -v[i];

    // keep the factor for later solves (uncertainty, solve_xt_JtJ_bt)
    if(IN_LDS)
        for(int idx = t; idx < n*n; idx += nt)
        {
            const int i = idx / n, j = idx - i*n;
            if(j <= i) S[(size_t)i*n + j] = chol_at(M, ld, i, j);
        }
}

// d_e = -L^-T (y_e + Wt_e d_s);  also scatters d_s into the state-ordered step
__global__ __launch_bounds__(64)
void backsub_kernel(NormalDims nd, BlockRanges br,
                    const double* __restrict__ Wt, const double* __restrict__ LD,
                    const double* __restrict__ y, const double* __restrict__ ds,
                    double* __restrict__ step)
{
    const int t   = threadIdx.x;
    if((int)blockIdx.x == br.count())
    {
        // the extra block copies d_s
        for(int i=t;i<nd.Nc;i+=blockDim.x)
            step[(i < nd.Nie) ? i : nd.i_state_warp + (i - nd.Nie)] = ds[i];
        return;
    }
    const int blk = br.block(blockIdx.x);
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    __shared__ double red[6][64];
    double part[6] = {0,0,0,0,0,0};
    for(int c=t;c<nd.Nc;c+=blockDim.x)
    {
        const double d = ds[c];
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
            for(int k=0;k<64;k++) s += red[i][k];
            v[i] = s;
        }
        for(int i=de-1;i>=0;i--)
        {
            double s = v[i];
            for(int k=i+1;k<de;k++) s -= L[k*6+i]*v[k];
            v[i] = s/L[i*6+i];
        }
        for(int i=0;i<de;i++) step[nd.Nie + e0 + i] = -v[i];
    }
}

////////////////////////////////////////////////////////////////////////////////
// v^T N v = |J v|^2 from the blocks;  dot products
////////////////////////////////////////////////////////////////////////////////
// one wave per row of [A ; Bt]: out += v_row * (row . v_S) * (1 for A rows, 2 for Bt rows);
// D blocks by the first NEb lanes of the grid
__global__ __launch_bounds__(64)
void quadform_kernel(NormalDims nd,
                     const double* __restrict__ A, const double* __restrict__ Bt, const double* __restrict__ D,
                     const double* __restrict__ v, double* __restrict__ out)
{
    const int row  = blockIdx.x;
    const int lane = threadIdx.x;
    const double* __restrict__ M = (row < nd.Nc) ? A + (size_t)row*nd.Nc : Bt + (size_t)(row - nd.Nc)*nd.Nc;
    double acc = 0.0;
    for(int c=lane;c<nd.Nc;c+=64)
        acc += M[c]*v[(c < nd.Nie) ? c : nd.i_state_warp + (c - nd.Nie)];
    for(int off=32; off>0; off>>=1) acc += __shfl_down(acc, off);
    if(lane == 0)
    {
        double vr, w;
        if(row < nd.Nc) { vr = v[(row < nd.Nie) ? row : nd.i_state_warp + (row - nd.Nie)]; w = 1.0; }
        else            { vr = v[nd.Nie + (row - nd.Nc)];                                   w = 2.0; }
        double total = w*vr*acc;
        // the D block of this E row
        if(row >= nd.Nc)
        {
            int blk, a, de, e0;
            E_to_block(nd, row - nd.Nc, &blk, &a, &de, &e0);
            double s = 0.0;
            for(int c=0;c<de;c++) s += D[(size_t)blk*36 + a*6 + c]*v[nd.Nie + e0 + c];
            total += vr*s;
        }
        atomicAdd(out, total);
    }
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
__global__ __launch_bounds__(256)
void outlier_stats_kernel(int Npoints_board, double thresh_sq,
                          const double* __restrict__ x, const double* __restrict__ pool,
                          int* __restrict__ counts, double* __restrict__ sums)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    double s = 0.0;
    int nout = 0, nbig = 0;
    if(i < Npoints_board)
    {
        const double w = pool[3*(size_t)i + 2];
        if(w <= 0.0) nout = 1;
        else
        {
            const double dx = x[2*(size_t)i], dy = x[2*(size_t)i+1];
            s = dx*dx + dy*dy;
            if(thresh_sq >= 0.0 && (dx*dx > thresh_sq || dy*dy > thresh_sq)) nbig = 1;
        }
    }
    for(int off=32; off>0; off>>=1)
    {
        s    += __shfl_down(s, off);
        nout += __shfl_down(nout, off);
        nbig += __shfl_down(nbig, off);
    }
    if((threadIdx.x & 63) == 0)
    {
        if(s != 0.0) atomicAdd(&sums[0], s);
        if(nout)     atomicAdd(&counts[0], nout);
        if(nbig)     atomicAdd(&counts[1], nbig);
    }
}

////////////////////////////////////////////////////////////////////////////////
// launchers
////////////////////////////////////////////////////////////////////////////////
hipError_t launch_assemble(const DeviceProblem& P, const NormalDims& nd, const AssemblyPlan& plan,
                           const EvalBuffers& B, const NormalBuffers& N, hipStream_t stream)
{
    hipError_t e;
    if((e = hipMemsetAsync(N.A,  0, (size_t)nd.Nc*nd.Nc*sizeof(double), stream)) != hipSuccess) return e;
    if(nd.NE > 0)
    {
        if((e = hipMemsetAsync(N.Bt, 0, (size_t)nd.NE*nd.Nc*sizeof(double), stream)) != hipSuccess) return e;
        if((e = hipMemsetAsync(N.D,  0, (size_t)nd.NEb*36*sizeof(double),   stream)) != hipSuccess) return e;
    }
    if((e = hipMemsetAsync(N.g,  0, (size_t)nd.Nstate*sizeof(double), stream)) != hipSuccess) return e;
    if((e = hipMemsetAsync(N.scalars, 0, NSCALARS*sizeof(double), stream)) != hipSuccess) return e;

    if(P.Nobs_board > 0)
    {
        if(P.do_optimize_frames)
            hipLaunchKernelGGL(assemble_frames_kernel, dim3(P.Nframes), dim3(256), 0, stream,
                               P, nd, plan.frame_obs_begin, B.gram, N.Bt, N.D, N.g);
        hipLaunchKernelGGL(reduce_pairs_kernel, dim3(plan.Nchunks), dim3(256), 0, stream,
                           P, nd, plan.chunk_begin, plan.pair_obs, B.gram, N.A, N.g, &N.scalars[SC_NORM2_X]);
    }
    const int row0 = 2*P.W*P.H*P.Nobs_board;
    if(P.Nmeas > row0)
        hipLaunchKernelGGL(rows_generic_kernel, dim3((P.Nmeas - row0 + 63)/64), dim3(64), 0, stream,
                           nd, row0, P.Nmeas, B.Jp, B.Ji, B.Jv, B.x, N.A, N.Bt, N.D, N.g, &N.scalars[SC_NORM2_X]);
    return hipGetLastError();
}

// Phase 1 of the Gauss-Newton solve, local to a shard: factor the local E
// blocks and form this shard's contribution to the Schur complement and to
// the reduced right-hand side: S_loc = A_loc (+ lambda I) - sum_local Wt^T Wt,
// r_loc = (g_S) - sum_local Wt^T y. The "(...)" terms are added by the shard
// leader only, so that the sum over shards has them once
hipError_t launch_factor_local(const NormalDims& nd, const BlockRanges& br,
                               const NormalBuffers& N, const FactorBuffers& F,
                               double lambda, bool is_leader, hipStream_t stream)
{
    hipError_t e;
    if((e = hipMemsetAsync(F.status, 0, sizeof(int), stream)) != hipSuccess) return e;
    if(br.count() > 0)
        hipLaunchKernelGGL(eblock_factor_kernel, dim3(br.count()), dim3(64), 0, stream,
                           nd, br, lambda, N.Bt, N.D, N.g, F.Wt, F.LD, F.y, F.status);
    {
        const size_t n2 = (size_t)nd.Nc*nd.Nc;
        hipLaunchKernelGGL(schur_init_kernel, dim3((unsigned)((n2 + 255)/256)), dim3(256), 0, stream,
                           nd, is_leader ? lambda : 0.0, is_leader ? 1 : 0, N.A, N.g, F.S, F.r);
    }
    // the E rows of the local blocks: two contiguous ranges (frames, points)
    for(int part = 0; part < 2; part++)
    {
        int e_lo, e_hi;
        br.e_range(nd, part, &e_lo, &e_hi);
        if(e_hi <= e_lo) continue;
        const int ntile  = (nd.Nc + SYRK_TILE - 1)/SYRK_TILE;
        int nslices = 2048 / (ntile*(ntile+1)/2);
        if(nslices < 1) nslices = 1;
        int e_per_slice = (e_hi - e_lo + nslices - 1)/nslices;
        e_per_slice = ((e_per_slice + 15)/16)*16;
        nslices = (e_hi - e_lo + e_per_slice - 1)/e_per_slice;
        hipLaunchKernelGGL(schur_syrk_kernel, dim3(ntile, ntile, nslices), dim3(256), 0, stream,
                           nd, e_lo, e_hi, e_per_slice, F.Wt, F.y, F.S, F.r);
    }
    return hipGetLastError();
}

// Phase 2: dense Cholesky of the (summed) Schur complement, d_S, and the
// back-substitution of the local E blocks into step_gn
hipError_t launch_solve_backsub(const NormalDims& nd, const BlockRanges& br,
                                const FactorBuffers& F, double* step_gn, hipStream_t stream)
{
    {
        const int n = nd.Nc;
        const size_t lds = ((size_t)n*(n|1) + n)*sizeof(double);
        if(lds <= 160*1024 - 64)
            hipLaunchKernelGGL((schur_cholesky_solve_kernel<true>), dim3(1), dim3(1024), lds, stream,
                               n, F.S, F.r, F.status);
        else
            hipLaunchKernelGGL((schur_cholesky_solve_kernel<false>), dim3(1), dim3(1024), 0, stream,
                               n, F.S, F.r, F.status);
    }
    if(br.count() < nd.NEb && nd.NE > 0)
    {
        // a shard writes only its own blocks; the others' entries must be 0 for
        // the all-reduce that follows (they may hold the previous sum)
        hipError_t e = hipMemsetAsync(step_gn + nd.Nie, 0, (size_t)nd.NE*sizeof(double), stream);
        if(e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(backsub_kernel, dim3(br.count()+1), dim3(64), 0, stream,
                       nd, br, F.Wt, F.LD, F.y, F.r, step_gn);
    return hipGetLastError();
}

hipError_t launch_factor_and_solve(const NormalDims& nd, const BlockRanges& br,
                                   const NormalBuffers& N, const FactorBuffers& F,
                                   double lambda, double* step_gn, hipStream_t stream)
{
    hipError_t e = launch_factor_local(nd, br, N, F, lambda, true, stream);
    if(e != hipSuccess) return e;
    return launch_solve_backsub(nd, br, F, step_gn, stream);
}

hipError_t launch_quadform(const NormalDims& nd, const NormalBuffers& N, const double* v, double* out,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(quadform_kernel, dim3(nd.Nc + nd.NE), dim3(64), 0, stream,
                       nd, N.A, N.Bt, N.D, v, out);
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
hipError_t launch_outlier_stats(int Npoints_board, double thresh_sq, const double* x, const double* pool,
                                int* counts, double* sums, hipStream_t stream)
{
    if(Npoints_board <= 0) return hipSuccess;
    hipLaunchKernelGGL(outlier_stats_kernel, dim3((Npoints_board+255)/256), dim3(256), 0, stream,
                       Npoints_board, thresh_sq, x, pool, counts, sums);
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

} // namespace mrcal_amd
