// Dependent-issue latency of the instructions on the pivot chain of chol_factor_diag16(): one wave, a chain of N
// dependent operations, cycles by s_memtime.   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_latency fp64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
template<int MODE>
__global__ void chain(double* out, long long* cyc, double a, double b)
{
    double x = a + threadIdx.x*1e-9;
    asm volatile("s_nop 0" : "+v"(x));
    const long long t0 = clock64();
    asm volatile("s_nop 0" : "+v"(x) : "s"(t0));
#pragma unroll
    for(int i = 0; i < N; i++)
    {
        if(MODE == 0) x = fma(x, b, a);                               // v_fma_f64
        if(MODE == 1) x = x*b;                                        // v_mul_f64
        if(MODE == 2) x = __builtin_amdgcn_rsq(x) + a;                // v_rsq_f64 + v_add_f64
        if(MODE == 3)
        {
            union { double d; int i[2]; } u; u.d = x;                 // v_readlane x2 -> sgpr pair -> v_fma
            u.i[0] = __builtin_amdgcn_readlane(u.i[0], 5); u.i[1] = __builtin_amdgcn_readlane(u.i[1], 5);
            x = fma(u.d, b, a);
        }
        if(MODE == 4) { float f = (float)x; f = fmaf(f, 1.0001f, 0.5f); x = f; }   // cvt + v_fma_f32 + cvt
    }
    asm volatile("s_nop 0" : "+v"(x));
    // (the timer read is ordered behind the chain by making it wait for a value that depends on x)
    const int xl = __builtin_amdgcn_readfirstlane((int)(x*0.0));
    long long t1;
    asm volatile("s_nop 4\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "s"(xl));
    out[threadIdx.x] = x;
    if(threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double* out; long long* cyc; (void)hipMalloc(&out, 64*8); (void)hipMalloc(&cyc, 8);
    const char* names[] = { "v_fma_f64", "v_mul_f64", "v_rsq_f64 + v_add_f64", "2 x v_readlane + v_fma_f64", "cvt f64->f32, v_fma_f32, cvt back" };
    for(int m = 0; m < 5; m++)
    {
        for(int rep = 0; rep < 2; rep++)
        {
            if(m == 0) chain<0><<<1,64>>>(out, cyc, 1.0, 0.999);
            if(m == 1) chain<1><<<1,64>>>(out, cyc, 1.0, 0.999);
            if(m == 2) chain<2><<<1,64>>>(out, cyc, 1.0, 0.999);
            if(m == 3) chain<3><<<1,64>>>(out, cyc, 1.0, 0.999);
            if(m == 4) chain<4><<<1,64>>>(out, cyc, 1.0, 0.999);
            (void)hipDeviceSynchronize();
        }
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-36s %6.1f cycles per link\n", names[m], (double)c/N);
    }
    return 0;
}
