// accuracy of v_rsq_f64 and of its Newton refinements (the pivot chain of the Cholesky kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o0, double* o1, double* o2, double* o1b, int n)
{
    int i = blockIdx.x*blockDim.x + threadIdx.x; if(i >= n) return;
    double p = x[i];
    double r = __builtin_amdgcn_rsq(p);
    o0[i] = r;
    double hp = -0.5*p;
    double r1 = r*fma(hp, r*r, 1.5);
    o1[i] = r1;
    double r2 = r1*fma(hp, r1*r1, 1.5);
    o2[i] = r2;
    // one step in the "residual" form: e = 1 - p r^2 (fma), r' = r + r*(e/2)
    double e = fma(-p*r, r, 1.0);
    o1b[i] = fma(r*0.5, e, r);
}
int main()
{
    const int n = 1<<20;
    std::vector<double> x(n), a(n), b(n), c(n), d(n);
    srand(1);
    for(int i = 0; i < n; i++) x[i] = exp((rand()/(double)RAND_MAX - 0.5)*60.0);
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n*8); hipMalloc(&d0, n*8); hipMalloc(&d1, n*8); hipMalloc(&d2, n*8); hipMalloc(&d3, n*8);
    hipMemcpy(dx, x.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n*8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n*8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n*8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for(int i = 0; i < n; i++)
    {
        long double t = 1.0L/sqrtl((long double)x[i]);
        e0 = fmax(e0, fabs((double)((a[i] - t)/t))); e1 = fmax(e1, fabs((double)((b[i] - t)/t)));
        e2 = fmax(e2, fabs((double)((c[i] - t)/t))); e3 = fmax(e3, fabs((double)((d[i] - t)/t)));
    }
    printf("max rel err: raw %.3g | 1 Newton %.3g | 2 Newton %.3g | 1 step residual form %.3g   (eps = %.3g)\n", e0, e1, e2, e3, 2.2e-16);
    return 0;
}
