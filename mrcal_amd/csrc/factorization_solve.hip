// Solves against a kept factorization (the CHOLMOD_factorization equivalent) and the CSR products of the uncertainty code
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "solver_kernel_decls.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// solves against a kept factorization (the CHOLMOD_factorization equivalent):
// (JtJ) x = b with N = JtJ = [A B; Bt D] factored as in launch_factor_local()
// + schur_cholesky (keep_factor): the E blocks' L_e in LD, Wt = L_e^-1 Bt_e,
// and the Cholesky factor of the Schur complement in the lower triangle of S.
//   y_e = L_e^-1 b_e ;  r = b_S - Wt^T y ;  x_S = S^-1 r ;  x_e = L_e^-T (y_e - Wt_e x_S)
////////////////////////////////////////////////////////////////////////////////

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
} // namespace mrcal_amd
