// The Schur complement onto the camera block: block elimination, the reduction, the SYRK kernels
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "solver_kernel_decls.hpp"

namespace mrcal_amd {


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
__device__ __forceinline__
void schur_reduce_body(const NormalDims& nd, const OpDev& O, double lambda, int add_g /* r starts from g_S (else from 0) */,
                       int nslots, const double* __restrict__ Spart,
                       double* __restrict__ S, double* __restrict__ r, int block,
                       const unsigned char* __restrict__ live = NULL /* [nslots][npairs]: the slot holds the tile (sparse SYRK); NULL: all do */,
                       double* __restrict__ iso = NULL /* given: S goes out COMPACTED by O.cperm (LcholCompact): the coupled
                                                         variables' n' x n' matrix with its rhs as row n', the isolated pairs' blocks here */,
                       int* __restrict__ err = NULL /* with iso: set to 3 if an entry that the compaction has no place for is not zero */,
                       double* __restrict__ ndMA = NULL, double* __restrict__ ndMB = NULL /* with iso and an active plan in O.ndp: the two sides' matrices (lchol_nd_*) */,
                       double* __restrict__ Spk = NULL /* without iso: a second, PACKED copy of the lower triangle ([j(j+1)/2 + i], the one-workgroup
                                                          Cholesky's own layout: what it loads is then 78 KB in a row instead of 140 pieces of rows) */)
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
            if(iso == NULL) { S[(size_t)j*nd.Nc + i] = v; if(Spk != NULL) Spk[(((size_t)j*(j + 1)) >> 1) + i] = v; }
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
            if(iso == NULL) { r[i] = v; if(Spk != NULL) Spk[(((size_t)nd.Nc*(nd.Nc + 1)) >> 1) + i] = v; }      // (the rhs: row Nc of the packed copy)
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
                         int ride_finish, Step2Dev sd, double* __restrict__ Spk)
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
                          (cperm_cur != NULL && O.cperm != NULL) ? iso : (double*)NULL, err, ndMA, ndMB, Spk);
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
int launch_syrk(const NormalDims& nd, const BlockRanges& br, const int* skip, const FactorBuffers& F,
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

} // namespace mrcal_amd
