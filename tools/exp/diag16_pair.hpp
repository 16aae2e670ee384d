// EXPERIMENT, not part of the product (round 4; measured and dropped: profiles/r04_diag16_variants.txt).
// Included by tools/exp/diag16_pair_bench.hip behind csrc/solver_kernels.hip.
namespace mrcal_amd {
// The same block by TWO waves (round 4). One wave's issue rate is what chol_factor_diag16() is bound by (250 cycles a
// column for ~30 vector instructions, of which the pivot chain - v_rsq_f64, two Newton steps, the scaling - is 12 that
// wait for each other and leave no free slot: profiles/r04_diag16_variants.txt), so the instructions that are NOT on
// the chain go to a second wave:
//   role 0, the chain:  column j = what the helper made of it (every update up to column j-FAST-1) + the last FAST
//                       updates, whose multipliers come by v_readlane from this wave's own registers; pivot, 1/sqrt,
//                       scaling; the scaled column published in LDS
//   role 1, the helper: keeps the columns right of FAST in ITS registers; for every published column k: row[c] -=
//                       L[i][k] L[c][k] for c > k+FAST, then column k+FAST+1 - complete up to k - published for the chain
// The two meet in LDS only (data, then a flag holding this call's epoch: LDS serves a wave's requests in order, so a
// reader that saw the flag sees the data); FAST columns of slack cover the round trip chain -> helper -> chain.
// Every update reaches an entry in ascending column order, by whichever wave: the result does not depend on timing.
//   ex: CHOL_PAIR_LDS_DOUBLES of LDS, its flags zero before the first call of a kernel; epoch: nonzero, different from
//   call to call. Both waves pass the same arguments; role 0 returns the not-positive flag and stores L and X
#ifndef CHOL_PAIR_FAST
#define CHOL_PAIR_FAST 4
#endif
#define CHOL_PAIR_LDS_DOUBLES (2*CHOL_PB*32 + 24)
#define CHOL_PAIR_SPIN_LIMIT (1 << 20)
// The LDS traffic between the two waves, spelled out: the compiler has no notion of another wave writing the LDS it is
// about to read (volatile made every access a FLAT load with its own wait: 9-15k cycles a block instead of 4k), and a
// flag and the data behind it are ONE round trip only if both reads are issued back to back, flag first
typedef double pair_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pair_lds(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ void pair_put(unsigned data_addr, double v, unsigned flag_addr, int epoch)
{
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3" :: "v"(data_addr), "v"(v), "v"(flag_addr), "v"(epoch) : "memory");
}
__device__ __forceinline__ void pair_put_flag(unsigned flag_addr, int epoch)
{
    asm volatile("ds_write_b32 %0, %1" :: "v"(flag_addr), "v"(epoch) : "memory");
}
__device__ __forceinline__ int pair_get_flag(unsigned flag_addr)
{
    int f;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(flag_addr) : "memory");
    return __builtin_amdgcn_readfirstlane(f);
}
// the flag, and one double behind it
__device__ __forceinline__ int pair_get(unsigned flag_addr, unsigned data_addr, double& v)
{
    int f;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(v) : "v"(flag_addr), "v"(data_addr) : "memory");
    return __builtin_amdgcn_readfirstlane(f);
}
// the flag, this lane's entry of a published column, and the column's entries 4..15 (one address for the whole wave:
// broadcasts). Always all twelve: a fixed instruction sequence, and what is not needed costs a read of 16 bytes
__device__ __forceinline__ int pair_get_column(unsigned flag_addr, unsigned mine_addr, unsigned col_addr /* entry 0 */,
                                               double& lk, pair_d2 (&m)[6])
{
    int f;
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b64 %1, %9\n\t"
                 "ds_read2_b64 %2, %10 offset0:4 offset1:5\n\tds_read2_b64 %3, %10 offset0:6 offset1:7\n\t"
                 "ds_read2_b64 %4, %10 offset0:8 offset1:9\n\tds_read2_b64 %5, %10 offset0:10 offset1:11\n\t"
                 "ds_read2_b64 %6, %10 offset0:12 offset1:13\n\tds_read2_b64 %7, %10 offset0:14 offset1:15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(lk), "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5])
                 : "v"(flag_addr), "v"(mine_addr), "v"(col_addr) : "memory");
    return __builtin_amdgcn_readfirstlane(f);
}
// entries 4..15 of a row of 16 doubles in LDS, as another wave left them (NOT through a __restrict__ pointer: the
// compiler may move such a load in front of the wait for the other wave's flag)
__device__ __forceinline__ void pair_get_entries_4_15(unsigned row_addr, pair_d2 (&m)[6])
{
    asm volatile("ds_read2_b64 %0, %6 offset0:4 offset1:5\n\tds_read2_b64 %1, %6 offset0:6 offset1:7\n\t"
                 "ds_read2_b64 %2, %6 offset0:8 offset1:9\n\tds_read2_b64 %3, %6 offset0:10 offset1:11\n\t"
                 "ds_read2_b64 %4, %6 offset0:12 offset1:13\n\tds_read2_b64 %5, %6 offset0:14 offset1:15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5]) : "v"(row_addr) : "memory");
}
template<class DstL>
__device__ __forceinline__
bool chol_factor_diag16_pair(const int role, const int lane, const int jb, const double* __restrict__ rowL,
                             double* __restrict__ X, double* __restrict__ ex, const int epoch, DstL dstL)
{
    constexpr int FAST = CHOL_PAIR_FAST;
    static_assert(FAST >= 3, "the helper reads a published column from entry 4 on");
    const int  r16  = lane & 15;
    const bool mine = lane < 16 && r16 < jb;
    const int  l32  = lane & 31;
    // ex: colL [16][32] scaled columns, by the chain | colU [16][32] columns brought up to date, by the helper |
    //     flagL [16], flagU [16], go: ints
    const unsigned colL = pair_lds(ex), colU = colL + CHOL_PB*32*8, flagL = colU + CHOL_PB*32*8, flagU = flagL + 16*4, go = flagL + 32*4;
    bool bad = false;
#define IC(v) std::integral_constant<int,(v)>{}
    if(role == 0)
    {
        double row[CHOL_PB];
        {
            double tmp[CHOL_PB];
#pragma unroll
            for(int c = 0; c < CHOL_PB; c++) tmp[c] = rowL[c];
            pair_put_flag(go, epoch);                     // (behind this wave's earlier writes of the block: in order)
#pragma unroll
            for(int c = 0; c < CHOL_PB; c++)
                row[c] = mine ? ((c <= r16) ? tmp[c] : 0.0) : ((lane < 32 && c == r16) ? 1.0 : 0.0);
        }
        auto column = [&](auto J)
        {
            constexpr int j = decltype(J)::value;
            double base = row[j];
            if constexpr(j > FAST)
            {
                int spin = 0;
                while(pair_get(flagU + 4*j, colU + 8*(j*32 + l32), base) != epoch)
                    if(++spin > CHOL_PAIR_SPIN_LIMIT) { bad = true; break; }
            }
#pragma unroll
            for(int k = (j > FAST ? j - FAST : 0); k < j; k++) base = fma(-row[k], readlane_f64(row[k], j), base);
            const double piv = readlane_f64(base, j);
            bad = bad || !(piv > 0.0);
#ifdef CHOL_PAIR_DEBUG
            if(lane == 0) { CHOL_PAIR_DEBUG[j] = piv; CHOL_PAIR_DEBUG[16 + j] = bad ? 1.0 : 0.0; }
#endif
            const double rd0 = __builtin_amdgcn_rsq(piv);
            const double hp  = -0.5*piv;
            const double sq  = rd0*rd0;
            const double lr  = base*rd0;
            const double u   = fma(hp, sq, 1.5);
            const double rd1 = rd0*u;
            const double u2  = fma(hp, rd1*rd1, 1.5);
            const double l   = (lr*u)*u2;
            row[j] = l;
            // (lanes 32..63 repeat lanes 0..31: same address, same value)
            if constexpr(j + FAST + 1 < CHOL_PB) pair_put(colL + 8*(j*32 + l32), l, flagL + 4*j, epoch);
        };
#define CHOL_COL(j) __builtin_amdgcn_sched_barrier(0); column(IC(j));
        CHOL_COL(0)  CHOL_COL(1)  CHOL_COL(2)  CHOL_COL(3)  CHOL_COL(4)  CHOL_COL(5)  CHOL_COL(6)  CHOL_COL(7)
        CHOL_COL(8)  CHOL_COL(9)  CHOL_COL(10) CHOL_COL(11) CHOL_COL(12) CHOL_COL(13) CHOL_COL(14) CHOL_COL(15)
#undef CHOL_COL
        const bool isid = (lane >= 16 && lane < 32);
        double* __restrict__ dstX = X + r16*CHOL_XLD;
#pragma unroll
        for(int c = 0; c < CHOL_PB; c++)
        {
            double* dst = isid ? dstX + c : dstL(c);
            *dst = row[c];
        }
        return bad;
    }
    // the helper
    {
        int spin = 0;
        while(pair_get_flag(go) != epoch) if(++spin > CHOL_PAIR_SPIN_LIMIT) return false;
    }
    double col[CHOL_PB];
    {
        pair_d2 t2[6];
        pair_get_entries_4_15(pair_lds(rowL), t2);
#pragma unroll
        for(int c = FAST + 1; c < CHOL_PB; c++)
        {
            const double tc = ((c - 4) & 1) ? t2[(c - 4) >> 1].y : t2[(c - 4) >> 1].x;
            col[c] = mine ? ((c <= r16) ? tc : 0.0) : ((lane < 32 && c == r16) ? 1.0 : 0.0);
        }
    }
    auto help = [&](auto K)
    {
        constexpr int k = decltype(K)::value;
        double lk; pair_d2 m2[6];
        int spin = 0;
        while(pair_get_column(flagL + 4*k, colL + 8*(k*32 + l32), colL + 8*(k*32), lk, m2) != epoch)
            if(++spin > CHOL_PAIR_SPIN_LIMIT) return;
#pragma unroll
        for(int c = k + FAST + 1; c < CHOL_PB; c++)
        {
            const double mc = ((c - 4) & 1) ? m2[(c - 4) >> 1].y : m2[(c - 4) >> 1].x;
            col[c] = fma(-lk, mc, col[c]);
        }
        pair_put(colU + 8*((k + FAST + 1)*32 + l32), col[k + FAST + 1], flagU + 4*(k + FAST + 1), epoch);
    };
#define CHOL_HELP(k) if constexpr((k) + FAST + 1 < CHOL_PB) { __builtin_amdgcn_sched_barrier(0); help(IC(k)); }
    CHOL_HELP(0) CHOL_HELP(1) CHOL_HELP(2) CHOL_HELP(3) CHOL_HELP(4) CHOL_HELP(5) CHOL_HELP(6) CHOL_HELP(7)
    CHOL_HELP(8) CHOL_HELP(9) CHOL_HELP(10) CHOL_HELP(11) CHOL_HELP(12) CHOL_HELP(13) CHOL_HELP(14)
#undef CHOL_HELP
#undef IC
    return false;
}

}
