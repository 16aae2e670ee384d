// Experiment: cost of launching 8000 one-wave workgroups with 13.8 KB of LDS each
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(_e)); exit(1);} }while(0)
__global__ __launch_bounds__(64) void k_trivial(double* out)
{
    extern __shared__ double t[];
    t[threadIdx.x] = 1.0;
    if(out == (double*)1) out[0] = t[0];
}
// n FP64 FMAs per lane, dependent chains of 4
__global__ __launch_bounds__(64) void k_fma(double* out, int n, double a)
{
    extern __shared__ double t[];
    double x0 = threadIdx.x, x1 = a, x2 = 2*a, x3 = 3*a;
    for(int i=0;i<n;i+=4) { x0 = x0*a + 1.0; x1 = x1*a + 1.0; x2 = x2*a + 1.0; x3 = x3*a + 1.0; }
    if(x0+x1+x2+x3 == 12345.0) out[0] = x0;
}
typedef double double4_t __attribute__((ext_vector_type(4)));
// n f64 MFMAs per wave, 3 independent accumulators
__global__ __launch_bounds__(64) void k_mfma(double* out, int n, double a)
{
    double4_t c0 = {0,0,0,0}, c1 = c0, c2 = c0;
    double x = a + threadIdx.x;
    for(int i=0;i<n;i+=3)
    {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c0, 0,0,0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c1, 0,0,0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c2, 0,0,0);
    }
    if(c0[0]+c1[1]+c2[2] == 12345.0) out[0] = c0[0];
}
// n f64 4x4x4 (4 blocks) MFMAs per wave, 7 independent accumulators
__global__ __launch_bounds__(64) void k_mfma4(double* out, int n, double a)
{
    double c[7] = {0,0,0,0,0,0,0};
    double x = a + threadIdx.x;
    for(int i=0;i<n;i+=7)
    {
#pragma unroll
        for(int j=0;j<7;j++) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, c[j], 0,0,0);
    }
    if(c[0]+c[1]+c[2]+c[3]+c[4]+c[5]+c[6] == 12345.0) out[0] = c[0];
}
// both: half of the waves (by block parity of b/8) FMA, the others MFMA
__global__ __launch_bounds__(64) void k_mixed(double* out, int nf, int nm, double a)
{
    if((blockIdx.x >> 3) & 1)
    {
        double x0 = threadIdx.x, x1 = a, x2 = 2*a, x3 = 3*a;
        for(int i=0;i<nf;i+=4) { x0 = x0*a + 1.0; x1 = x1*a + 1.0; x2 = x2*a + 1.0; x3 = x3*a + 1.0; }
        if(x0+x1+x2+x3 == 12345.0) out[0] = x0;
    }
    else
    {
        double4_t c0 = {0,0,0,0}, c1 = c0, c2 = c0;
        double x = a + threadIdx.x;
        for(int i=0;i<nm;i+=3)
        {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c0, 0,0,0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c1, 0,0,0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c2, 0,0,0);
        }
        if(c0[0]+c1[1]+c2[2] == 12345.0) out[0] = c0[0];
    }
}
template<class F> static float timeit(F f, int n)
{
    hipEvent_t e0,e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for(int i=0;i<n;i++) f();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
    return ms/n*1e3;
}
int main()
{
    double* out; CHECK(hipMalloc(&out, 1024));
    printf("trivial 8000x64, 13.8KB lds: %.2f us\n", timeit([&]{ hipLaunchKernelGGL(k_trivial, dim3(8000), dim3(64), 13824, 0, out); }, 20));
    printf("trivial 8000x64, 0 lds     : %.2f us\n", timeit([&]{ hipLaunchKernelGGL(k_trivial, dim3(8000), dim3(64), 512, 0, out); }, 20));
    // 8192 waves = 8 per SIMD
    const int NW = 8192;
    float tf = timeit([&]{ hipLaunchKernelGGL(k_fma, dim3(NW), dim3(64), 0, 0, out, 4000, 1.0000001); }, 10);
    printf("fma64: %d waves x 4000 FMA: %.2f us -> %.2f cycles/FMA/wave at 2.4GHz (8 waves/SIMD)\n", NW, tf, tf*1e-6*2.4e9/(8*4000));
    float tm = timeit([&]{ hipLaunchKernelGGL(k_mfma, dim3(NW), dim3(64), 0, 0, out, 300, 1.0000001); }, 10);
    printf("mfma64: %d waves x 300 MFMA: %.2f us -> %.2f cycles/MFMA/SIMD (8 waves/SIMD)\n", NW, tm, tm*1e-6*2.4e9/(8*300));
    float t4 = timeit([&]{ hipLaunchKernelGGL(k_mfma4, dim3(NW), dim3(64), 0, 0, out, 700, 1.0000001); }, 10);
    printf("mfma64 4x4x4: %d waves x 700 MFMA: %.2f us -> %.2f cycles/MFMA/SIMD (8 waves/SIMD)\n", NW, t4, t4*1e-6*2.4e9/(8*700));
    // mixed: 4 waves/SIMD each kind, sized to take the same time alone
    float tx = timeit([&]{ hipLaunchKernelGGL(k_mixed, dim3(NW), dim3(64), 0, 0, out, 4000, 300, 1.0000001); }, 10);
    printf("mixed (half the waves 4000 FMA, half 300 MFMA): %.2f us   [alone: fma %.2f, mfma %.2f at half the waves]\n", tx, tf/2, tm/2);
    return 0;
}
