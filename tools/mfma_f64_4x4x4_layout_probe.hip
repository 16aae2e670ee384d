// Determines the lane layout of v_mfma_f64_4x4x4f64 (4 blocks of D = A(4x4) B(4x4))
// empirically, with no hypothesis: for every (lane of A, lane of B) pair that
// is nonzero, which lane of D receives the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const double* a_in, const double* b_in, double* d_out)
{
    int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], acc, 0, 0, 0);
    d_out[l] = acc;
}
int main()
{
    double *a, *b, *d;
    hipMalloc(&a, 64*8); hipMalloc(&b, 64*8); hipMalloc(&d, 64*8);
    double ha[64], hb[64], hd[64];
    // A lane la carries 2^la-ish unique tags is overkill; do 64 runs, one per A lane,
    // with B = distinct primes per lane: D[ld] = sum of A[la]*B[lb] identifies lb
    int contrib[64][64]; // contrib[ld][la] = lb or -1
    for(int i=0;i<64;i++) for(int j=0;j<64;j++) contrib[i][j] = -1;
    for(int la=0; la<64; la++)
    {
        for(int l=0;l<64;l++) { ha[l] = (l==la) ? 1.0 : 0.0; hb[l] = (double)(l+1); }
        hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
        hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
        for(int ld=0; ld<64; ld++)
            if(hd[ld] != 0.0) contrib[ld][la] = (int)hd[ld] - 1;
    }
    // print: for each D lane, the (la,lb) pairs
    for(int ld=0; ld<64; ld++)
    {
        printf("D lane %2d <- ", ld);
        for(int la=0; la<64; la++) if(contrib[ld][la] >= 0) printf("(A%2d,B%2d) ", la, contrib[ld][la]);
        printf("\n");
    }
    // test the hypothesis: block b = l/16 for all three; A lane = 16b + 4k + i ; B lane = 16b + 4k + j ; D lane = 16b + 4i + j  (and variants)
    const char* names[4] = { "A:4k+i B:4k+j D:4i+j", "A:4k+i B:4k+j D:4j+i", "A:4i+k B:4j+k D:4i+j", "A:4i+k B:4j+k D:4j+i" };
    for(int h=0; h<4; h++)
    {
        int bad = 0;
        for(int bl=0; bl<4; bl++) for(int i=0;i<4;i++) for(int j=0;j<4;j++) for(int k=0;k<4;k++)
        {
            int la = 16*bl + ((h<2) ? 4*k+i : 4*i+k);
            int lb = 16*bl + ((h<2) ? 4*k+j : 4*j+k);
            int ld = 16*bl + ((h%2==0) ? 4*i+j : 4*j+i);
            if(contrib[ld][la] != lb) bad++;
        }
        printf("hypothesis %s: %d mismatches\n", names[h], bad);
    }
    return 0;
}
