// accuracy of v_rsq_f64 on gfx950, raw and after 1 and 2 Newton steps (dev tool)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/rsq rsq_f64_probe.hip && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* out, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if(i >= n) return;
    const double p = x[i];
    double rd = __builtin_amdgcn_rsq(p);
    out[3*i+0] = rd;
    rd = rd*(1.5 - 0.5*p*rd*rd);
    out[3*i+1] = rd;
    rd = rd*(1.5 - 0.5*p*rd*rd);
    out[3*i+2] = rd;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(3*n);
    std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-30., 30.);
    for(auto& v : x) v = std::exp(u(g));
    double *dx, *dout;
    hipMalloc(&dx, n*8); hipMalloc(&dout, 3*n*8);
    hipMemcpy(dx, x.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 3*n*8, hipMemcpyDeviceToHost);
    double e[3] = {0,0,0};
    for(int i=0;i<n;i++)
        for(int j=0;j<3;j++)
        {
            const long double ref = 1.0L/sqrtl((long double)x[i]);
            const double err = (double)fabsl(((long double)o[3*i+j] - ref)/ref);
            if(err > e[j]) e[j] = err;
        }
    printf("max relative error of v_rsq_f64: raw %.3e, 1 Newton %.3e, 2 Newton %.3e\n", e[0], e[1], e[2]);
    return 0;
}
