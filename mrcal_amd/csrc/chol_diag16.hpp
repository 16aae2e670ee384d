// The 16 x 16 diagonal block of a Cholesky panel by one wave in registers (chol_factor_diag16) and the sizes of the
// one-workgroup LDS Cholesky: shared by cholesky_lds.hip (the camera blocks that fit the LDS) and cholesky_large.hip
// (the launch-per-panel Cholesky, whose 64 x 64 diagonal blocks are four of these)
#pragma once
#include "solver_device.hpp"

namespace mrcal_amd {

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

} // namespace mrcal_amd
