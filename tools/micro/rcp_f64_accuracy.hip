// Accuracy of v_rcp_f64 and of one / two Newton refinements on gfx950 (max relative error over 1M random inputs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p = x[i];
    double a = __builtin_amdgcn_rcp(p);
    r0[i] = a;
    a = fma(a, fma(-p, a, 1.0), a); r1[i] = a;
    a = fma(a, fma(-p, a, 1.0), a); r2[i] = a;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (int i = 0; i < n; i++) { double m = 1.0 + std::fabs(u(g)); x[i] = std::ldexp(u(g) < 0 ? -m : m, (int)(u(g) * 40)); }
    double *dx, *d0, *d1, *d2;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d0, n * 8); (void)hipMalloc(&d1, n * 8); (void)hipMalloc(&d2, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    (void)hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / (long double)x[i];
        e0 = std::fmax(e0, (double)std::fabs(((long double)a[i] - t) / t)); e1 = std::fmax(e1, (double)std::fabs(((long double)b[i] - t) / t));
        e2 = std::fmax(e2, (double)std::fabs(((long double)c[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err %.3e; + 1 Newton step %.3e; + 2 Newton steps %.3e\n", e0, e1, e2);
    return 0;
}
