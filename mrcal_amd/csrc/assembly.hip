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
// The assembly of the block normal equations (from the per-observation Grams; rows outside the Grams; a caller's bare matrix)
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "assembly_splined.hpp"
#include "solver_kernel_decls.hpp"

namespace mrcal_amd {


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
// the planned rows as workgroups of assemble_factor_kernel's launch: how many, and the launch's LDS with them
int gen_ride_blocks(const AssemblyPlan& plan)
{
    const GenPlan& G = plan.gen;
    return (G.Nrows > 0 && G.Nchunks + G.Neblocks > 0) ? G.Nchunks + G.Neblocks : 0;
}
size_t assemble_lds_bytes_with_gen(const NormalDims& nd, const AssemblyPlan& plan)
{
    const size_t a = assemble_lds_bytes(nd), b = gen_ride_blocks(plan) ? gen_lds_bytes(plan.gen, nd) : 0;
    return a > b ? a : b;
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

// The Gram assembly (+ elimination of the frame blocks), the generic rows and the planned rows in
// ONE launch (all four kinds of work are independent): workgroups
// [0, nframe_blocks) take a frame each, the next Nchunks*slices 256 positions of a pair chunk each, the
// next 256 generic rows each, the last ngen a chunk of planned rows or an eliminated block of theirs each.
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
                            const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, int ngen)
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
    else if(b >= (int)gridDim.x - ngen)
    {
        // (round 6) the planned rows - discrete points, triangulated pairs - as the launch's last workgroups: they were
        // gen_rows_kernel, a launch of its own behind this one that needed nothing of it (11.5 + 9.5 us at BASELINE
        // configuration 5). The same functions on the same point (skip_asm is elim_mode != 1)
        const int gb = b - ((int)gridDim.x - ngen);
        if(gb < plan.gen.Nchunks) gen_chunk(plan.gen, O, Jp, gb, lds_f);
        else                      gen_eblock(plan.gen, nd, O, Jp, gb - plan.gen.Nchunks, lds_f);
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
// the planned rows' sums: a wave a destination
__global__ __launch_bounds__(64)
void assemble_finalize_gen_kernel(int npos, NormalDims nd, const OpDev* __restrict__ ops, const int* __restrict__ sel,
                                  const int* __restrict__ skip, AssemblyPlan plan)
{
    if(skip != NULL && *skip) return;
    assemble_finalize<FIN_LANES_GEN>(npos, nd, ops[sel ? *sel : 0], plan, blockIdx.x*blockDim.x + threadIdx.x);
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
// launchers
////////////////////////////////////////////////////////////////////////////////
hipError_t launch_zero_normal(const NormalDims& nd, const OpRef& R, hipStream_t stream)
{
    const size_t total = (size_t)nd.Nc*nd.Nc + (size_t)nd.NE*nd.Nc + (size_t)nd.NEb*36 + nd.Nstate + NSCALARS;
    int nb = (int)((total + 255)/256); if(nb > 2048) nb = 2048;
    hipLaunchKernelGGL(zero_normal_kernel, dim3(nb), dim3(256), 0, stream, nd, R);
    return hipGetLastError();
}
// the planned rows: chunks and eliminated blocks in one launch; then, AFTER whatever else finalizes into A, g and
// |x|^2 (launches on a stream are ordered: every destination is added to by one thread at a time), their sums
hipError_t launch_gen_rows(const NormalDims& nd, const AssemblyPlan& plan, const OpRef& R, const int32_t* Jp, hipStream_t stream)
{
    const GenPlan& G = plan.gen;
    if(G.Nrows <= 0 || G.Nchunks + G.Neblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(gen_rows_kernel, dim3(G.Nchunks + G.Neblocks), dim3(256), gen_lds_bytes(G, nd), stream, nd, R, G, Jp);
    return hipGetLastError();
}
hipError_t launch_gen_finalize(const NormalDims& nd, const AssemblyPlan& plan, const OpRef& R, hipStream_t stream)
{
    const GenPlan& G = plan.gen;
    if(G.Nrows <= 0 || G.Ndest <= 0) return hipSuccess;
    AssemblyPlan gp = plan;
    gp.Ndest = G.Ndest; gp.dest_id = G.dest_id; gp.dest_begin = G.dest_begin; gp.dest_src = G.dest_src;
    gp.pair_chunk_begin = G.group_chunk_begin; gp.chunk_part = G.part; gp.row_part_n = 0;
    hipLaunchKernelGGL(assemble_finalize_gen_kernel, dim3((G.Ndest*FIN_LANES_GEN + 63)/64), dim3(64), 0, stream,
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
        const int ngen = gen_ride_blocks(plan);
        hipLaunchKernelGGL(assemble_factor_kernel, dim3(nframe_blocks + plan.Nchunks*assemble_chunk_slices(P) + assemble_row_blocks(P, plan) + ngen), dim3(256),
                           assemble_lds_bytes_with_gen(nd, plan), stream, P, nd, br, B.R.ops, B.R.sel, B.R.sel, (const SolverCtl*)NULL, B.R.skip,
                           (const int*)NULL, 1, 0, 0.0, plan, B.gram, none, nframe_blocks, assemble_row0(P, plan), P.Nmeas, B.Jp, B.Ji, ngen);
        hipError_t e = hipGetLastError();
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
