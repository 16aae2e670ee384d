// chol_factor_diag16_pair(): the 16 x 16 diagonal block by two waves, timed alone and checked against a host Cholesky
// (L and X = L^-T).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrcal_amd/csrc -o /tmp/diag16_pair_bench tools/exp/diag16_pair_bench.hip
__device__ double g_pair_dbg[64];
#define CHOL_PAIR_DEBUG g_pair_dbg
#include "../../mrcal_amd/csrc/solver_kernels.hip"
#include "diag16_pair.hpp"
#include <cstdio>
#include <cmath>
__global__ __launch_bounds__(128)
void diag16_pair_bench_kernel(const double* __restrict__ Ain, double* __restrict__ out, double* __restrict__ Xout, long long* cyc, int nrep, int jb)
{
    __shared__ __attribute__((aligned(16))) double A[16*17];
    __shared__ __attribute__((aligned(16))) double W[16*17];
    __shared__ __attribute__((aligned(16))) double X[16*CHOL_XLD];
    __shared__ __attribute__((aligned(16))) double ex[CHOL_PAIR_LDS_DOUBLES];
    __shared__ double sinkbuf[128];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15;
    for(int i = t; i < 16*17; i += 128) A[i] = Ain[i];
    for(int i = t; i < CHOL_PAIR_LDS_DOUBLES; i += 128) ex[i] = 0.0;
    __syncthreads();
    bool bad = false;
    const long long t0 = clock64();
    for(int rep = 0; rep < nrep; rep++)
    {
        if(wave == 0) for(int i = lane; i < 16*17; i += 64) W[i] = A[i];
        double* rowL = &W[((r16 < jb) ? r16 : 0)*17];
        double* sink = sinkbuf + t;
        const bool mine = lane < 16 && r16 < jb;
        bad |= mrcal_amd::chol_factor_diag16_pair(wave, lane, jb, rowL, X, ex, rep + 1,
                                                  [&](int c) -> double* { return (mine && c <= r16) ? rowL + c : sink; });
        __syncthreads();
    }
    const long long t1 = clock64();
    if(t == 0) { cyc[0] = (t1 - t0)/nrep; cyc[1] = bad; }
    for(int i = t; i < 16*17; i += 128) out[i] = W[i];
    for(int i = t; i < 16*CHOL_XLD; i += 128) Xout[i] = X[i];
}
int main()
{
    double h[16*17];
    for(int i = 0; i < 16; i++) for(int j = 0; j < 17; j++) h[i*17 + j] = (i == j) ? 20.0 + i : 1.0/(1 + i + j);
    double *dA, *dout, *dX; long long* dc;
    (void)hipMalloc(&dA, sizeof(h)); (void)hipMalloc(&dout, sizeof(h)); (void)hipMalloc(&dX, 16*CHOL_XLD*8); (void)hipMalloc(&dc, 16);
    (void)hipMemcpy(dA, h, sizeof(h), hipMemcpyHostToDevice);
    for(int jb = 16; jb >= 12; jb -= 4)
    {
        for(int k = 0; k < 2; k++) { diag16_pair_bench_kernel<<<1,128>>>(dA, dout, dX, dc, 200, jb); (void)hipDeviceSynchronize(); }
        long long c[2]; (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        double o[16*17], x[16*CHOL_XLD]; (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); (void)hipMemcpy(x, dX, sizeof(x), hipMemcpyDeviceToHost);
        double L[16][16] = {{0}}; double err = 0.0, errx = 0.0;
        for(int j = 0; j < 16; j++)
        {
            if(j >= jb) { L[j][j] = 1.0; continue; }
            double d = h[j*17+j]; for(int k = 0; k < j; k++) d -= L[j][k]*L[j][k]; L[j][j] = sqrt(d);
            for(int i = j+1; i < jb; i++) { double v = h[i*17+j]; for(int k = 0; k < j; k++) v -= L[i][k]*L[j][k]; L[i][j] = v/L[j][j]; }
        }
        for(int i = 0; i < jb; i++) for(int j = 0; j <= i; j++) { const double e = fabs(o[i*17+j] - L[i][j]); err = (e == e) ? fmax(err, e) : 1e300; }
        // X[k][c] = (L^-T)[k][c]: X^T L^T... check  sum_k L[i][k] X[c][k]... X = L^-T  <=>  L^T X = I  <=>  sum_k L[k][i] X[k][c] = delta_ic
        for(int i = 0; i < 16; i++) for(int c = 0; c < 16; c++)
        {
            double sum = 0.0; for(int k = 0; k < 16; k++) sum += L[k][i]*x[k*CHOL_XLD + c];
            { const double e = fabs(sum - (i == c ? 1.0 : 0.0)); errx = (e == e) ? fmax(errx, e) : 1e300; }
        }
        { double dbg[64]; (void)hipMemcpyFromSymbol(dbg, HIP_SYMBOL(g_pair_dbg), sizeof(dbg)); printf("pivots:"); for(int i = 0; i < 16; i++) printf(" %.4g(%g)", dbg[i], dbg[16+i]); printf("\n"); }
        printf("chol_factor_diag16_pair (FAST %d, jb %d): %lld cycles per call (incl. a 272-double LDS copy and a barrier), not-positive flag %lld, max |L - host| %.3g, max |L^T X - I| %.3g\n",
               CHOL_PAIR_FAST, jb, c[0], c[1], err, errx);
    }
    return 0;
}
