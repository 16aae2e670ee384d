// What a store-only stream can reach on this GPU: the ceiling of the board
// kernel, whose traffic is ~90% writes of Jacobian values. Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_store_bw tools/probe_store_bw.hip && /tmp/probe_store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

template<int MODE>   // 0: 8-byte stores, 1: 16-byte stores, 2: 16-byte nontemporal, 3: 8-byte, each wave writes 24 consecutive doubles per row like a CSR row
__global__ __launch_bounds__(256) void fill(double* __restrict__ p, size_t n, double v)
{
    const size_t tid = (size_t)blockIdx.x*blockDim.x + threadIdx.x, nth = (size_t)gridDim.x*blockDim.x;
    if(MODE == 0)      for(size_t i = tid; i < n; i += nth) p[i] = v + i;
    else if(MODE == 1) for(size_t i = tid; i < n/2; i += nth) ((double2*)p)[i] = make_double2(v + i, v);
    else if(MODE == 2) for(size_t i = tid; i < n/2; i += nth) { __builtin_nontemporal_store(v + i, &p[2*i]); __builtin_nontemporal_store(v, &p[2*i+1]); }
    else if(MODE == 3) for(size_t i = tid; i < n; i += nth) __builtin_nontemporal_store(v + i, &p[i]);
}
template<int MODE> __global__ __launch_bounds__(256) void copy(const double* __restrict__ s, double* __restrict__ p, size_t n)
{
    const size_t tid = (size_t)blockIdx.x*blockDim.x + threadIdx.x, nth = (size_t)gridDim.x*blockDim.x;
    for(size_t i = tid; i < n/2; i += nth) ((double2*)p)[i] = ((const double2*)s)[i];
}
template<int MODE> __global__ __launch_bounds__(256) void rd(const double* __restrict__ s, double* __restrict__ p, size_t n)
{
    const size_t tid = (size_t)blockIdx.x*blockDim.x + threadIdx.x, nth = (size_t)gridDim.x*blockDim.x;
    double acc = 0; for(size_t i = tid; i < n/2; i += nth) { double2 a = ((const double2*)s)[i]; acc += a.x + a.y; }
    if(acc == 1.2345) p[0] = acc;
}
template<typename F> static void timeit(const char* name, double bytes, F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int i = 0; i < 3; i++) f();
    float best = 1e9, sum = 0;
    for(int i = 0; i < 20; i++) { CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms; }
    printf("%-46s avg %.1f us  best %.1f us  -> %.2f TB/s (best %.2f)\n", name, sum/20*1e3, best*1e3, bytes/(sum/20*1e-3)/1e12, bytes/(best*1e-3)/1e12);
}
int main()
{
    const size_t n = 37200080ull + 1600080ull;      // J values + x, doubles: the board kernel's output
    double *p, *s; CK(hipMalloc(&p, n*8)); CK(hipMalloc(&s, n*8)); CK(hipMemset(s, 0, n*8));
    const double B = n*8.0;
    for(int grid : {1024, 2048, 4096, 16384, 65536})
    {
        char nm[128];
        snprintf(nm, sizeof(nm), "fill  8B stores, grid %d", grid);      timeit(nm, B, [&]{ hipLaunchKernelGGL(fill<0>, dim3(grid), dim3(256), 0, 0, p, n, 1.0); });
        snprintf(nm, sizeof(nm), "fill 16B stores, grid %d", grid);      timeit(nm, B, [&]{ hipLaunchKernelGGL(fill<1>, dim3(grid), dim3(256), 0, 0, p, n, 1.0); });
        snprintf(nm, sizeof(nm), "fill 2x8B nontemporal, grid %d", grid); timeit(nm, B, [&]{ hipLaunchKernelGGL(fill<2>, dim3(grid), dim3(256), 0, 0, p, n, 1.0); });
        snprintf(nm, sizeof(nm), "fill 8B nontemporal, grid %d", grid);  timeit(nm, B, [&]{ hipLaunchKernelGGL(fill<3>, dim3(grid), dim3(256), 0, 0, p, n, 1.0); });
    }
    timeit("hipMemsetAsync", B, [&]{ CK(hipMemsetAsync(p, 0, n*8, 0)); });
    timeit("copy 16B (read+write, bytes counted twice)", 2*B, [&]{ hipLaunchKernelGGL(copy<0>, dim3(8192), dim3(256), 0, 0, s, p, n); });
    timeit("read 16B", B, [&]{ hipLaunchKernelGGL(rd<0>, dim3(8192), dim3(256), 0, 0, s, p, n); });
    return 0;
}
