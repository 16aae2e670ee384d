// Experiment (round 6): what does it cost to hand a TILE from one workgroup to another inside a
// launch on an otherwise idle chip - the regime a persistent Cholesky chain would run in (one chain workgroup fed by
// tile workgroups), as opposed to the 5.4 MB publish of round 5's fused prologue that LEDGER R5.2 priced it with?
// Two workgroups of 512 threads play ping-pong with a tile of T bytes: the sender stores the tile, raises a flag;
// the receiver polls the flag, reads the whole tile (into registers, summed so that nothing is optimized away), then
// answers with its own tile. 400 hops; the time per hop is the total over the hops. Two forms of the hand-off
// (cdna_hip_programming.md Guideline 16):
//   sc1    16-byte write-through stores, s_waitcnt vmcnt(0), sc1 flag store | sc1 poll, sc1 16-byte loads
//   fence  plain stores, __syncthreads, lane 0: release fence (agent), vmcnt(0), relaxed flag | relaxed poll, acquire fence, plain loads
// and two placements: the partner on the SAME XCD (blocks 0 and 8: workgroups go to the XCDs round-robin) or on the next
// one (blocks 0 and 1). The other blocks of the grid exit at once.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff tools/exp/handoff_32k.hip && /tmp/handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(_e)); exit(1);} }while(0)
typedef double d2 __attribute__((ext_vector_type(2)));
#define NT 512

template<bool SC1>
__device__ __forceinline__ void send(double* tile, int n2 /* 16-byte pieces */, unsigned* flag, unsigned value, double seed)
{
    for(int i = threadIdx.x; i < n2; i += NT)
    {
        d2 v; v.x = seed + i; v.y = seed - i;
        if(SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(tile + 2*i), "v"(v) : "memory");
        else    *reinterpret_cast<d2*>(tile + 2*i) = v;
    }
    if(SC1)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if(threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
        __syncthreads();
        if(threadIdx.x == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template<bool SC1>
__device__ __forceinline__ double receive(const double* tile, int n2, unsigned* flag, unsigned value)
{
    if(threadIdx.x == 0)
    {
        int spins = 0;
        while(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
        if(!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    double s = 0.0;
    // (all of a thread's loads asked for before any is used: up to 8 in flight)
    d2 v[8];
    for(int i0 = threadIdx.x; i0 < n2; i0 += 8*NT)
    {
#pragma unroll
        for(int u = 0; u < 8; u++)
        {
            const int i = i0 + u*NT;
            if(i < n2)
            {
                if(SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(tile + 2*i) : "memory");
                else    v[u] = *reinterpret_cast<const d2*>(tile + 2*i);
            }
        }
        if(SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for(int u = 0; u < 8; u++) if(i0 + u*NT < n2) s += v[u].x + v[u].y;
    }
    return s;
}
template<bool SC1>
__global__ __launch_bounds__(NT) void pingpong(double* tiles, unsigned* flags, int n2, int partner_block, int nhops,
                                               unsigned long long* ticks, double* sink)
{
    const bool first = blockIdx.x == 0, second = (int)blockIdx.x == partner_block;
    if(!first && !second) return;
    double* mine   = tiles + (first ? 0 : 2*(size_t)n2);
    double* theirs = tiles + (first ? 2*(size_t)n2 : 0);
    unsigned* fmine = flags + (first ? 0 : 32), *ftheirs = flags + (first ? 32 : 0);
    double acc = 0.0;
    const unsigned long long t0 = wall_clock64();
    for(int k = 1; k <= nhops; k++)
    {
        if(first) { send<SC1>(mine, n2, fmine, (unsigned)k, (double)k);      acc += receive<SC1>(theirs, n2, ftheirs, (unsigned)k); }
        else      { acc += receive<SC1>(theirs, n2, ftheirs, (unsigned)k);   send<SC1>(mine, n2, fmine, (unsigned)k, acc*1e-30); }
        __syncthreads();
    }
    if(first && threadIdx.x == 0) *ticks = wall_clock64() - t0;
    if(acc == 12345.678) sink[0] = acc;
}
int main()
{
    double* tiles; unsigned* flags; unsigned long long* ticks; double* sink;
    CHECK(hipMalloc(&tiles, 4*(size_t)(1 << 20))); CHECK(hipMalloc(&flags, 256)); CHECK(hipMalloc(&ticks, 8)); CHECK(hipMalloc(&sink, 8));
    const int nhops = 200;
    for(int partner : {8, 1})
        for(int bytes : {64, 4096, 32768, 65536, 262144})
            for(int sc1 = 1; sc1 >= 0; sc1--)
            {
                double best = 1e30;
                for(int rep = 0; rep < 3; rep++)
                {
                    CHECK(hipMemset(flags, 0, 256));
                    if(sc1) hipLaunchKernelGGL(pingpong<true>,  dim3(16), dim3(NT), 0, 0, tiles, flags, bytes/16, partner, nhops, ticks, sink);
                    else    hipLaunchKernelGGL(pingpong<false>, dim3(16), dim3(NT), 0, 0, tiles, flags, bytes/16, partner, nhops, ticks, sink);
                    CHECK(hipDeviceSynchronize());
                    unsigned long long t; CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
                    const double us = t*0.01/(2.0*nhops);
                    if(us < best) best = us;
                }
                printf("partner block %d (%s XCD), tile %6d B, %-5s : %6.2f us per hop (store + flag + poll + read)\n",
                       partner, partner == 8 ? "same" : "next", bytes, sc1 ? "sc1" : "fence", best);
            }
    return 0;
}
