// v_permlane32_swap_b32 (gfx950): which lanes of which operand trade places. hipcc --offload-arch=gfx950 -O2 -o /tmp/pl tools/exp/permlane32_swap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* a, unsigned* b)
{
    unsigned x = a[threadIdx.x], y = b[threadIdx.x];
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    a[threadIdx.x] = r[0]; b[threadIdx.x] = r[1];
}
int main()
{
    unsigned ha[64], hb[64], *a, *b;
    for(int i=0;i<64;i++) { ha[i] = i; hb[i] = 100 + i; }
    hipMalloc(&a, 256); hipMalloc(&b, 256);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b);
    hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost);
    printf("first result  (was lane):      "); for(int i=0;i<64;i+=8) printf(" [%d]=%u", i, ha[i]); printf("\n");
    printf("second result (was 100+lane):  "); for(int i=0;i<64;i+=8) printf(" [%d]=%u", i, hb[i]); printf("\n");
    return 0;
}
