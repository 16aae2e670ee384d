// Experiment (round 6): how fast does the chip START one-wave workgroups that look like the board kernel's
// (64 threads, 18 KB of LDS, 240 VGPRs => two per SIMD, 2048 slots)? Every wave stamps the wall clock (100 MHz) when
// it starts and when it ends after spinning `work` microseconds; the host prints the start-time distribution and the
// span of the launch. Also with 2 and 4 waves per workgroup (same LDS per wave) and with small register footprints.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_rate tools/exp/dispatch_rate.hip && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHECK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(_e)); exit(1);} }while(0)

template<int NV>
__device__ __forceinline__ void body(unsigned long long* ts, int work_ticks, double* sink)
{
    extern __shared__ double lds[];
    const unsigned long long t0 = wall_clock64();
    // keep NV doubles live across the spin: the register footprint of the board kernel
    double r[NV];
#pragma unroll
    for(int i=0;i<NV;i++) r[i] = (double)(threadIdx.x + i);
    lds[threadIdx.x] = r[0];
    while((long long)(wall_clock64() - t0) < work_ticks)
    {
#pragma unroll
        for(int i=0;i<NV;i++) r[i] = r[i]*1.0000001 + 1e-9;
    }
    double s = 0;
#pragma unroll
    for(int i=0;i<NV;i++) s += r[i];
    if(s == 12345.678) sink[0] = s + lds[0];
    const int w = blockIdx.x*(blockDim.x/64) + threadIdx.x/64;
    if((threadIdx.x & 63) == 0) { ts[2*w] = t0; ts[2*w+1] = wall_clock64(); }
}
__global__ __launch_bounds__(64)  void k_big1(unsigned long long* ts, int wt, double* sink)   { body<110>(ts, wt, sink); }
__global__ __launch_bounds__(128) void k_big2(unsigned long long* ts, int wt, double* sink)   { body<110>(ts, wt, sink); }
__global__ __launch_bounds__(256) void k_big4(unsigned long long* ts, int wt, double* sink)   { body<110>(ts, wt, sink); }
__global__ __launch_bounds__(64)  void k_small1(unsigned long long* ts, int wt, double* sink) { body<8>(ts, wt, sink); }

template<class K>
static void run(const char* name, K kernel, int nwaves, int wpb, int lds_per_wave, double work_us)
{
    unsigned long long* ts; double* sink;
    CHECK(hipMalloc(&ts, (size_t)nwaves*16)); CHECK(hipMalloc(&sink, 64));
    std::vector<unsigned long long> h(2*(size_t)nwaves);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for(int rep=0; rep<3; rep++)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(nwaves/wpb), dim3(64*wpb), (size_t)lds_per_wave*wpb, 0, ts, (int)(work_us*100), sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipMemcpy(h.data(), ts, (size_t)nwaves*16, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> st(nwaves);
    for(int i=0;i<nwaves;i++) { t0 = std::min(t0, h[2*i]); t1 = std::max(t1, h[2*i+1]); }
    for(int i=0;i<nwaves;i++) st[i] = (h[2*i] - t0)*0.01;
    std::sort(st.begin(), st.end());
    printf("%-10s %5d waves (%d/wg, %5d B LDS/wave) work %4.1f us: events %6.1f us, first start -> last end %6.1f us; "
           "starts at 10/50/90/99/100 %%: %5.2f %5.2f %5.2f %5.2f %5.2f us\n",
           name, nwaves, wpb, lds_per_wave, work_us, ms*1e3, (t1 - t0)*0.01,
           st[nwaves/10], st[nwaves/2], st[nwaves*9/10], st[nwaves*99/100], st[nwaves-1]);
    CHECK(hipFree(ts)); CHECK(hipFree(sink));
}
int main()
{
    for(double work : {0.0, 5.0, 14.0})
        for(int n : {1024, 1600, 2048, 3200, 8000})
        {
            run("big 1/wg",   k_big1,   n, 1, 18432, work);
            run("big 2/wg",   k_big2,   n, 2, 18432, work);
            run("big 4/wg",   k_big4,   n, 4, 18432, work);
            run("small 1/wg", k_small1, n, 1, 18432, work);
            run("small nolds", k_small1, n, 1, 512, work);
        }
    return 0;
}
