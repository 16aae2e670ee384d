// The launch-per-panel Cholesky of big camera blocks (the splined models), its nested-dissection form, the back-substitution
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "chol_diag16.hpp"
#include "solver_kernel_decls.hpp"

namespace mrcal_amd {

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
__global__ __launch_bounds__(256)
void lchol_apply_inverse_kernel(const int* __restrict__ n_dev, int n_host, const int* __restrict__ skip, double* __restrict__ M,
                                const double* __restrict__ Linv, int with_post, Step2Dev sd, int* __restrict__ chol_status,
                                LcholCompact cp, LcholDiagSpread diag_minmax)
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
        // (round 6) how far apart the factor's diagonal entries lie: X's diagonal is 1 / L's. For the automatic fallback
        // to the backward sweep (solver.cpp): d = -Y^T z through the explicit inverse loses n eps / (min / max) digits
        if(diag_minmax != NULL)
        {
            const unsigned long long lb = (unsigned long long)__double_as_longlong(fabs(1.0/Linv[(size_t)q*NB*NB + (size_t)j*NB + j]));
            atomicMin(&diag_minmax[0], lb); atomicMax(&diag_minmax[1], lb);
        }
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
                           const int* __restrict__ nperm, double* __restrict__ dout, const double* __restrict__ ndpart,
                           LcholDiagSpread diag_minmax)
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
        if(v >= 0 && diag_minmax != NULL)         // (the positions that pad a side to whole panels are identity rows)
        {
            const unsigned long long lb = (unsigned long long)__double_as_longlong(fabs(1.0/C.Linv[(size_t)q*NB*NB + (size_t)j*NB + j]));
            atomicMin(&diag_minmax[0], lb); atomicMax(&diag_minmax[1], lb);
        }
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
                                 const Step2Dev* sd, bool* fused, const int* n_dev, const LcholCompact* compact,
                                 int likely_panels /* with n_dev: launches 0 .. likely_panels one by one, the rest in lchol_tail_kernel; 0: all one by one */,
                                 unsigned* tail_counter, const LcholNdLaunch* nds,
                                 bool finish_done /* with sd: the end-of-trial logic has run already (the first launch goes by `skip`); the verdict still rides in the last */,
                                 bool sweep /* the solve by the backward sweep in groups of panels (rounds 2-3; backward stable) instead of through
                                                       L^-1 built on the side (lchol_inverse_block): FactorBuffers::use_sweep */,
                                 LcholDiagSpread diag_minmax)
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
                           n_dev, n, skip, M, (const double*)Linv, fuse ? 1 : 0, fuse ? *sd : sd0, status, compact ? *compact : cp0, diag_minmax);
        if(nds != NULL)
        {
            const int ncb = LCH_NB*nds->lim.rounds/LCH_AI_COLS;
            hipLaunchKernelGGL(lchol_nd_apply_kernel, dim3(2*ncb), dim3(256), 0, stream, nds->A, nds->B, nds->ndh, skip, ncb,
                               nds->ndh + NDH_WORDS + compact->Nc, compact->dout, (const double*)compact->ndpart, diag_minmax);
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

} // namespace mrcal_amd
