// which way does DPP row_ror:n move data? (dev probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out)
{
    const int l = threadIdx.x;
    out[l]      = __builtin_amdgcn_update_dpp(0, l, 0x120 + 4,  0xf, 0xf, false);
    out[64 + l] = __builtin_amdgcn_update_dpp(0, l, 0x120 + 12, 0xf, 0xf, false);
}
int main()
{
    int* d; int h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1,64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_ror:4  lanes 0..19: "); for(int i=0;i<20;i++) printf("%d ", h[i]);      printf("\n");
    printf("row_ror:12 lanes 0..19: "); for(int i=0;i<20;i++) printf("%d ", h[64+i]);   printf("\n");
    return 0;
}
