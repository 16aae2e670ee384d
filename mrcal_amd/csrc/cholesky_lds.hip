// The camera block's dense Cholesky by one workgroup in LDS (camera blocks to 178 variables)
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "chol_diag16.hpp"
#include "solver_kernel_decls.hpp"
#include "step_device.hpp"

namespace mrcal_amd {

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

// FINISH: 0 a factorization and solve and nothing else | 1 the end of the trial step in front, the verdict behind (sharded:
// the end-of-trial logic needs the tail summed over the ranks, which is complete only now) | 2 the verdict behind alone:
// the end-of-trial logic has run in the reduction's launch (single GPU, round 5: step2_reduce_kernel) and `skip` is its word
// (the kernel's body. Round 6 ran it as the first workgroup of a launch that also held the back-substitution's workgroups,
//  waiting for a word it left: measured 5 us SLOWER than the two launches - LEDGER R6.16 - and taken out again)
template<int FINISH>
__device__ __forceinline__
void schur_cholesky_solve_body(int n, const int* __restrict__ skip, int keep_factor,
                               double* __restrict__ S, double* __restrict__ r,
                               int* __restrict__ status, const Step2Dev& sd, const double* __restrict__ Spk = NULL)
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
    // (round 6: where the reduction left a PACKED copy - this kernel's own layout - the triangle is one run of memory:
    //  a tenth of a load per entry and thread, all of them asked for at once, instead of 28 loads a thread of which
    //  half fetch what lies above the diagonal: CHOL_TS, 11.5k cycles of the kernel's 112k)
    if(Spk != NULL)
    {
        const int ntri = ((n*(n + 1)) >> 1) + n;        // (with the right-hand side, row n)
        // Wave 0 takes the first diagonal block's rows (136 entries) and nothing else, and goes on to factor that block
        // - it reads what it wrote itself - while the other fifteen waves bring the rest in: no barrier before the loop's
        // own at the end of its first pass (p = -1). The first block's chain started 8k cycles into the kernel, behind
        // everybody's loads; it needs a fiftieth of them
        const int n16 = min(n, CHOL_PB), nfirst = (n16*(n16 + 1)) >> 1;
        if(wave_u == 0)
        {
            double v[3];
#pragma unroll
            for(int u = 0; u < 3; u++) { const int idx = lane + 64*u; v[u] = Spk[idx < nfirst ? idx : 0]; }
#pragma unroll
            for(int u = 0; u < 3; u++) { const int idx = lane + 64*u; if(idx < nfirst) Mp[idx] = v[u]; }
        }
        else
        {
            constexpr int UP = 16;
            const int nt1 = nt - 64;
            for(int i0 = nfirst + (t - 64); i0 < ntri; i0 += UP*nt1)
            {
                double v[UP];
#pragma unroll
                for(int u = 0; u < UP; u++) { const int idx = i0 + u*nt1; v[u] = Spk[idx < ntri ? idx : 0]; }
#pragma unroll
                for(int u = 0; u < UP; u++) { const int idx = i0 + u*nt1; if(idx < ntri) Mp[idx] = v[u]; }
            }
        }
    }
    else
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
    if(Spk == NULL)
    {
        for(int j = t; j < n; j += nt) rowptr(n)[j] = r[j];
        __syncthreads();
    }

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
            // (the last, partial panel. Its reads asked for together - from rows that exist, the terms past the panel's end
            //  left out of the same chain of multiply-adds - measured: this panel 4.6k -> 5.1k cycles and the full panels'
            //  chain behind it 1.6k -> 2.2k each; taken out again)
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
    // (round 6) Wave 0's chain through the full panels stays in REGISTERS: d_p, which lanes 0..15 have just made, reaches
    // the sixteen dot products that update panel p-1's entries by v_readlane instead of through an LDS write and read,
    // and so do the updated entries on their way into X_{p-1}'s product. The same products and sums in the same order
    // as back_update() / back_diag() - the same bits -; a panel of the sweep was 2.2k cycles, four LDS round trips of
    // which two were the wave talking to itself (CHOL_TS: 23.7k cycles for the sweep of 140 variables)
    auto back_chain = [&](int p, double& dcur) __attribute__((always_inline))
    {
        const int j0 = p*CHOL_PB, jm = j0 - CHOL_PB, l16 = (lane < CHOL_PB) ? lane : 0;
        const int i = jm + l16;
        const double* __restrict__ X = Xs + (p - 1)*CHOL_PB*CHOL_XLD + l16*CHOL_XLD;
        double lv[CHOL_PB], xv[CHOL_PB];
#pragma unroll
        for(int c = 0; c < CHOL_PB; c++) { lv[c] = rowptr(j0+c)[i]; xv[c] = X[c]; }
        const double zi = z[i];
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for(int c = 0; c < CHOL_PB; c += 2) { acc0 += lv[c]*readlane_f64(dcur, c); acc1 += lv[c+1]*readlane_f64(dcur, c+1); }
        const double w = zi - (acc0 + acc1);
        double b0 = 0.0, b1 = 0.0;
#pragma unroll
        for(int k = 0; k < CHOL_PB; k += 2)
        {
            const double w0 = readlane_f64(w, k), w1 = readlane_f64(w, k+1);
            b0 += (k   >= lane) ? xv[k]  *w0 : 0.0;
            b1 += (k+1 >= lane) ? xv[k+1]*w1 : 0.0;
        }
        dcur = b0 + b1;
        if(lane < CHOL_PB) z[jm + lane] = dcur;
    };
    double dcur = 0.0;
    bool have_dcur = false;
    for(int p = npanels-1; p >= 1; p--)
    {
        const int j0 = p*CHOL_PB;
        const int jb = min(CHOL_PB, n - j0);
        if(wave == 0)
        {
            if(have_dcur && jb == CHOL_PB) back_chain(p, dcur);
            else
            {
                if(lane < CHOL_PB) back_update(j0 - CHOL_PB + lane, j0, jb);     // panel p-1 is full
                back_diag(p - 1);
                dcur = z[j0 - CHOL_PB + ((lane < CHOL_PB) ? lane : 0)];           // (the wave's own writes: in order)
                have_dcur = true;
            }
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

template<int FINISH>
__global__ __launch_bounds__(1024)
void schur_cholesky_solve_kernel(int n, const int* __restrict__ skip, int keep_factor,
                                 double* __restrict__ S, double* __restrict__ r,
                                 int* __restrict__ status, Step2Dev sd)
{
    schur_cholesky_solve_body<FINISH>(n, skip, keep_factor, S, r, status, sd);
}

// (round 6) The quadratic form of a new current point - g^T N g and |g_E|^2, per-workgroup partials for the next choice -
// never needed the factorization: its workgroups rode in the launch BEHIND it, beside the back-substitution's, since
// round 3. On a single GPU (the end of the trial - which point is current, is it new - decided in the reduction's launch)
// they are workgroups of the factorization's launch instead: independent of its one workgroup, nothing to wait for, done
// long before it; the launch behind is the back-substitution alone. Four of step2_backsub_quadform_kernel's workgroups
// of 256 to one of 1024 here (quadform_body counts the quarters): the same partials, the same bits. Every workgroup
// reserves the factorization's LDS (the launch's one size): a workgroup a CU at 140 variables - 192 of them
__global__ __launch_bounds__(1024)
void step2_chol_quadform_kernel(int n, int* __restrict__ status, Step2Dev sd, NormalDims nd,
                                double* __restrict__ S, double* __restrict__ r, double* __restrict__ qf_part, int nqf,
                                const double* __restrict__ Spk)
{
    if(blockIdx.x == 0)
    {
        schur_cholesky_solve_body<2>(n, (const int*)&sd.fl->skip_chol, 0, S, r, status, sd, Spk);
        return;
    }
    const SolverCtl* __restrict__ ctl = sd.ctl;
    if(!ctl->derive) return;
    const OpDev& O = sd.ops[ctl->ib];
    const int qb = 4*((int)blockIdx.x - 1) + (int)(threadIdx.x >> 8);
    // (a quarter past the last row: its waves clamp their rows and add nothing - but every thread is at the barrier)
    const double mine = quadform_body(nd, O, O.g, qb, true, (const unsigned*)NULL, 0);
    if((threadIdx.x & 255) < 3 && qb < nqf) qf_part[4*qb + (threadIdx.x & 255)] = mine;
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

// the launches, for the units that do not see the kernels (solver_kernel_decls.hpp)
hipError_t launch_cholesky_lds(int finish, int n, const int* skip, int keep_factor, double* S, double* r, int* status,
                               const Step2Dev& sd, hipStream_t stream)
{
    const dim3 g(1), b(1024);
    const size_t lds = chol_lds_bytes(n);
    switch(finish)
    {
    case 0:  hipLaunchKernelGGL(schur_cholesky_solve_kernel<0>, g, b, lds, stream, n, skip, keep_factor, S, r, status, sd); break;
    case 1:  hipLaunchKernelGGL(schur_cholesky_solve_kernel<1>, g, b, lds, stream, n, skip, keep_factor, S, r, status, sd); break;
    case 2:  hipLaunchKernelGGL(schur_cholesky_solve_kernel<2>, g, b, lds, stream, n, skip, keep_factor, S, r, status, sd); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_cholesky_lds_quadform(int n, const NormalDims& nd, const FactorBuffers& F, const Step2Dev& sd,
                                        double* qf_part, int nqf, hipStream_t stream)
{
    hipLaunchKernelGGL(step2_chol_quadform_kernel, dim3(1 + (nqf + 3)/4), dim3(1024), chol_lds_bytes(n), stream,
                       n, F.status, sd, nd, F.S, F.r, qf_part, nqf, (const double*)factor_S_packed(F, n));
    return hipGetLastError();
}
hipError_t launch_cholesky_global(int n, const int* skip, double* S, double* r, int* status, hipStream_t stream)
{
    hipLaunchKernelGGL(schur_cholesky_solve_global_kernel, dim3(1), dim3(1024), 0, stream, n, skip, S, r, status);
    return hipGetLastError();
}

} // namespace mrcal_amd
