// Does a consumer kernel find the producer kernel's data in its XCD's L2? Kernel A: workgroup i
// writes a 16 KB chunk i. Kernel B: workgroup i reads chunk (i + shift) % n, 16 bytes per lane per
// load, 4 loads in flight, and the time of B is measured. shift 0: the chunk its own XCD (WG i ->
// XCD i % 8) produced; shift 1: a neighbour XCD's; shift 8: same XCD, another CU's.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/xcd_locality tools/exp/xcd_locality.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)
constexpr int CHUNK = 2048;      // doubles per workgroup: 16 KB
__global__ __launch_bounds__(256) void produce(double* p, double v)
{
    double2* q = (double2*)(p + (size_t)blockIdx.x*CHUNK);
    for(int i = threadIdx.x; i < CHUNK/2; i += 256) q[i] = make_double2(v + i, v);
}
__global__ __launch_bounds__(256) void consume(const double* p, double* out, int shift, int n)
{
    const double2* q = (const double2*)(p + (size_t)((blockIdx.x + shift) % n)*CHUNK);
    double acc = 0;
    double2 a = q[threadIdx.x], b = q[threadIdx.x + 256], c = q[threadIdx.x + 512], d = q[threadIdx.x + 768];
    acc = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    if(acc == 1.2345) out[0] = acc;
}
int main()
{
    for(int n : {256, 2048, 16384})
    {
        double *p, *o; CK(hipMalloc(&p, (size_t)n*CHUNK*8)); CK(hipMalloc(&o, 64));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for(int shift : {0, 1, 3, 8, 9, 64})
        {
            float best = 1e9, sum = 0;
            for(int it = 0; it < 12; it++)
            {
                hipLaunchKernelGGL(produce, dim3(n), dim3(256), 0, 0, p, (double)it);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(consume, dim3(n), dim3(256), 0, 0, p, o, shift, n);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if(it >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("n %6d (%6.1f MB) shift %2d: consume avg %.2f us best %.2f us\n", n, n*CHUNK*8/1e6, shift, sum/10*1e3, best*1e3);
        }
        hipFree(p); hipFree(o);
    }
    return 0;
}
