// Experiment: HBM write rate of the store patterns the board kernel could use.
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(_e)); exit(1);} }while(0)

// total: NOBS observations x 4800 doubles (200 rows x 24)
constexpr int ROWLEN = 24;
constexpr int NROWS  = 200;
constexpr int OBSLEN = ROWLEN*NROWS;

// A: fully coalesced, one wave per observation
__global__ __launch_bounds__(64) void k_coalesced(double* out, double v)
{
    double* o = out + (size_t)blockIdx.x*OBSLEN;
    for(int e = 2*threadIdx.x; e < OBSLEN; e += 128)
    {
        double2 t; t.x = v + e; t.y = v - e;
        *reinterpret_cast<double2*>(&o[e]) = t;
    }
}
// A4: fully coalesced, 4 waves (256 threads) per block, 4 observations per block
__global__ __launch_bounds__(256) void k_coalesced256(double* out, double v)
{
    double* o = out + (size_t)blockIdx.x*OBSLEN*4;
    for(int e = 2*threadIdx.x; e < OBSLEN*4; e += 512)
    {
        double2 t; t.x = v + e; t.y = v - e;
        *reinterpret_cast<double2*>(&o[e]) = t;
    }
}
// B: row per lane (192 B contiguous per lane), lane stride 192 B; 64 rows per pass
__global__ __launch_bounds__(64) void k_row_per_lane(double* out, double v)
{
    double* o = out + (size_t)blockIdx.x*OBSLEN;
    for(int r0 = 0; r0 < NROWS; r0 += 64)
    {
        const int r = r0 + threadIdx.x;
        if(r < NROWS)
        {
            double* row = o + (size_t)r*ROWLEN;
#pragma unroll
            for(int c=0;c<ROWLEN;c+=2)
            {
                double2 t; t.x = v + c; t.y = v - r;
                *reinterpret_cast<double2*>(&row[c]) = t;
            }
        }
    }
}
// C: corner per lane (2 rows = 384 B contiguous per lane)
__global__ __launch_bounds__(64) void k_corner_per_lane(double* out, double v)
{
    double* o = out + (size_t)blockIdx.x*OBSLEN;
    for(int p0 = 0; p0 < NROWS/2; p0 += 64)
    {
        const int p = p0 + threadIdx.x;
        if(p < NROWS/2)
        {
            double* row = o + (size_t)p*2*ROWLEN;
#pragma unroll
            for(int c=0;c<2*ROWLEN;c+=2)
            {
                double2 t; t.x = v + c; t.y = v - p;
                *reinterpret_cast<double2*>(&row[c]) = t;
            }
        }
    }
}
// D: through LDS: lanes write rows to LDS (odd stride), then coalesced copy-out; nw waves per block
template<int NW>
__global__ __launch_bounds__(64*NW) void k_lds(double* out, double v)
{
    __shared__ double tile[NW][128*25];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* o = out + ((size_t)blockIdx.x*NW + w)*OBSLEN;
    double* T = tile[w];
    for(int p0 = 0; p0 < NROWS/2; p0 += 64)
    {
        const int p = p0 + lane;
        const int np = (NROWS/2 - p0 < 64) ? NROWS/2 - p0 : 64;
        if(p < NROWS/2)
        {
#pragma unroll
            for(int c=0;c<ROWLEN;c++)
            {
                T[(2*lane)*25 + c]   = v + c;
                T[(2*lane+1)*25 + c] = v - c;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int nelem = 2*np*ROWLEN;
        double* oo = o + (size_t)2*p0*ROWLEN;
        for(int e = 2*lane; e < nelem; e += 128)
        {
            const int r = e / ROWLEN, c = e - r*ROWLEN;  // ROWLEN even: both in the same row
            double2 t; t.x = T[r*25 + c]; t.y = T[r*25 + c + 1];
            *reinterpret_cast<double2*>(&oo[e]) = t;
        }
        __builtin_amdgcn_wave_barrier();
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
    return ms/n;
}
int main()
{
    const int NOBS = 8000;
    double* out; size_t bytes = (size_t)NOBS*OBSLEN*8;
    CHECK(hipMalloc(&out, bytes));
    auto rep = [&](const char* name, float ms){ printf("%-28s %8.2f us  %7.1f GB/s\n", name, ms*1e3, bytes/1e9/(ms*1e-3)); };
    rep("memset",          timeit([&]{ hipMemsetAsync(out, 0, bytes, 0); }, 20));
    rep("coalesced 1 wave/obs",  timeit([&]{ hipLaunchKernelGGL(k_coalesced, dim3(NOBS), dim3(64), 0, 0, out, 1.0); }, 20));
    rep("coalesced 256 thr",     timeit([&]{ hipLaunchKernelGGL(k_coalesced256, dim3(NOBS/4), dim3(256), 0, 0, out, 1.0); }, 20));
    rep("row per lane (192B)",   timeit([&]{ hipLaunchKernelGGL(k_row_per_lane, dim3(NOBS), dim3(64), 0, 0, out, 1.0); }, 20));
    rep("corner per lane (384B)",timeit([&]{ hipLaunchKernelGGL(k_corner_per_lane, dim3(NOBS), dim3(64), 0, 0, out, 1.0); }, 20));
    rep("via LDS, 1 wave/blk",   timeit([&]{ hipLaunchKernelGGL(k_lds<1>, dim3(NOBS), dim3(64), 0, 0, out, 1.0); }, 20));
    rep("via LDS, 4 wave/blk",   timeit([&]{ hipLaunchKernelGGL(k_lds<4>, dim3(NOBS/4), dim3(256), 0, 0, out, 1.0); }, 20));
    return 0;
}
