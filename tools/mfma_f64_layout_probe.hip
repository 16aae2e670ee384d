// Determines the lane/register layout of v_mfma_f64_16x16x4_f64 empirically:
// D = A(16x4) B(4x16). Prints, for A and B, which (row,col) each lane feeds, and
// for D which (i,j) each (lane,reg) holds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* a_in, const double* b_in, double* d_out)
{
    int l = threadIdx.x;
    double4_t acc = {0,0,0,0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_in[l], b_in[l], acc, 0, 0, 0);
    for(int v=0; v<4; v++) d_out[v*64+l] = acc[v];
}
int main()
{
    double *a, *b, *d;
    hipMalloc(&a, 64*8); hipMalloc(&b, 64*8); hipMalloc(&d, 256*8);
    double ha[64], hb[64], hd[256];
    // hypothesis: lane l feeds A[l%16][l/16], B[l/16][l%16]; D[4*(l/16)+v][l%16] at (l,v)
    // test: A[i][k] = i+1 + 100*(k+1) ; B[k][j] = (j+1) + 1000*(k+1)  under the hypothesis
    double A[16][4], B[4][16];
    for(int i=0;i<16;i++) for(int k=0;k<4;k++) A[i][k] = (i+1) + 0.125*(k+1);
    for(int k=0;k<4;k++) for(int j=0;j<16;j++) B[k][j] = (j+1)*0.5 + 3.0*(k+1);
    for(int l=0;l<64;l++) { ha[l] = A[l%16][l/16]; hb[l] = B[l/16][l%16]; }
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for(int l=0;l<64;l++) for(int v=0;v<4;v++)
    {
        double got = hd[v*64+l];
        int i1 = 4*(l/16)+v, j1 = l%16;      // hypothesis 1
        int i2 = (l/16)+4*v, j2 = l%16;      // hypothesis 2
        double r1=0, r2=0;
        for(int k=0;k<4;k++) { r1 += A[i1][k]*B[k][j1]; r2 += A[i2][k]*B[k][j2]; }
        if(got != r1) bad1++;
        if(got != r2) bad2++;
    }
    printf("hypothesis1 (i=4*(l/16)+v, j=l%%16): %d mismatches; hypothesis2 (i=(l/16)+4v): %d mismatches\n", bad1, bad2);
    // also transposed hypotheses
    int bad3=0, bad4=0;
    for(int l=0;l<64;l++) for(int v=0;v<4;v++)
    {
        double got = hd[v*64+l];
        int j3 = 4*(l/16)+v, i3 = l%16;
        int j4 = (l/16)+4*v, i4 = l%16;
        double r3=0, r4=0;
        for(int k=0;k<4;k++) { r3 += A[i3][k]*B[k][j3]; r4 += A[i4][k]*B[k][j4]; }
        if(got != r3) bad3++;
        if(got != r4) bad4++;
    }
    printf("transposed: h3 %d, h4 %d\n", bad3, bad4);
    return 0;
}
