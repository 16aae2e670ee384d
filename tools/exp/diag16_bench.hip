// chol_factor_diag16() alone: one wave factoring a 16 x 16 block held in LDS, N times, cycles per call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrcal_amd/csrc -o /tmp/diag16_bench tools/exp/diag16_bench.hip
#include "../../mrcal_amd/csrc/solver_kernels.hip"
#include <cstdio>
#include <cmath>
__global__ __launch_bounds__(64)
void diag16_bench_kernel(const double* __restrict__ Ain, double* __restrict__ out, long long* cyc, int nrep)
{
    __shared__ __attribute__((aligned(16))) double A[16*17];
    __shared__ __attribute__((aligned(16))) double W[16*17];
    __shared__ __attribute__((aligned(16))) double X[16*CHOL_XLD];
    __shared__ __attribute__((aligned(16))) double cb[6*64];
    const int lane = threadIdx.x, r16 = lane & 15;
    for(int i = lane; i < 16*17; i += 64) A[i] = Ain[i];
    __syncthreads();
    bool bad = false;
    const long long t0 = clock64();
    for(int rep = 0; rep < nrep; rep++)
    {
        for(int i = lane; i < 16*17; i += 64) W[i] = A[i];
        double* rowL = &W[r16*17];
        double* sink = cb + 5*64 + lane;
        bad |= mrcal_amd::chol_factor_diag16(lane, 16, rowL, X, cb, [&](int c) -> double* { return (lane < 16) ? rowL + c : sink; });
    }
    const long long t1 = clock64();
    if(lane == 0) { cyc[0] = (t1 - t0)/nrep; cyc[1] = bad; }
    for(int i = lane; i < 16*17; i += 64) out[i] = W[i];
}
int main()
{
    double h[16*17];
    for(int i = 0; i < 16; i++) for(int j = 0; j < 17; j++) h[i*17 + j] = (i == j) ? 20.0 + i : 1.0/(1 + i + j);
    double *dA, *dout; long long* dc;
    (void)hipMalloc(&dA, sizeof(h)); (void)hipMalloc(&dout, sizeof(h)); (void)hipMalloc(&dc, 16);
    (void)hipMemcpy(dA, h, sizeof(h), hipMemcpyHostToDevice);
    for(int k = 0; k < 2; k++) { diag16_bench_kernel<<<1,64>>>(dA, dout, dc, 200); (void)hipDeviceSynchronize(); }
    long long c[2]; (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    double o[16*17]; (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    double sum = 0.0; for(int i = 0; i < 16; i++) for(int j = 0; j <= i; j++) sum += o[i*17+j]*(1 + i + 3*j);
    // (the factor against a plain host Cholesky of the same matrix)
    double L[16][16] = {{0}}; double err = 0.0;
    for(int j = 0; j < 16; j++) { double d = h[j*17+j]; for(int k = 0; k < j; k++) d -= L[j][k]*L[j][k]; L[j][j] = sqrt(d);
        for(int i = j+1; i < 16; i++) { double v = h[i*17+j]; for(int k = 0; k < j; k++) v -= L[i][k]*L[j][k]; L[i][j] = v/L[j][j]; } }
    for(int i = 0; i < 16; i++) for(int j = 0; j <= i; j++) err = fmax(err, fabs(o[i*17+j] - L[i][j]));
    printf("chol_factor_diag16: %lld cycles per call (incl. a 272-double LDS copy), not-positive flag %lld, L[0][0] = %.6f (sqrt(20) = 4.472136), L[15][15] = %.6f, weighted sum %.15g, max |L - host Cholesky| %.3g\n", c[0], c[1], o[0], o[15*17+15], sum, err);
    return 0;
}
