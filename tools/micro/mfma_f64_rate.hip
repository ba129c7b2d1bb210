// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on gfx950 (one wave per SIMD, 256 CUs).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters) {
    v4d c[CHAINS];
    for (int j = 0; j < CHAINS; j++) c[j] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < CHAINS; j++) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
    }
    double s = 0.0;
    for (int j = 0; j < CHAINS; j++) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters) {
    double c[CHAINS];
    for (int j = 0; j < CHAINS; j++) c[j] = threadIdx.x * 1e-9 * j;
    double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < CHAINS; j++) c[j] = fma(c[j], a, b);
    }
    double s = 0.0;
    for (int j = 0; j < CHAINS; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
double time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; hipMalloc(&out, 256 * 1024 * sizeof(double));
    const int iters = 20000, blocks = 256;   // one 256-thread block per CU: one wave per SIMD
    for (int waves = 1; waves <= 2; waves++) {
        const int nb = blocks * waves;
        double m1 = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(nb), dim3(256), 0, 0, out, iters); });
        double m4 = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(nb), dim3(256), 0, 0, out, iters); });
        double f1 = time_ms([&] { hipLaunchKernelGGL(k_fma<1>, dim3(nb), dim3(256), 0, 0, out, iters); });
        double f8 = time_ms([&] { hipLaunchKernelGGL(k_fma<8>, dim3(nb), dim3(256), 0, 0, out, iters); });
        const double simd = 256.0 * 4.0;
        printf("waves/SIMD %d: mfma f64 16x16x4: dependent chain %.1f ns/instr, 4 chains %.1f ns/instr (%.1f TFLOP/s); v_fma_f64: dependent %.2f ns/instr, 8 chains %.2f ns/instr (%.1f TFLOP/s)\n",
               waves, m1 * 1e6 / iters / waves, m4 * 1e6 / (4.0 * iters) / waves, 2048.0 * 4 * iters * simd * waves / (m4 * 1e-3) / 1e12,
               f1 * 1e6 / iters / waves, f8 * 1e6 / (8.0 * iters) / waves, 128.0 * 8 * iters * simd * waves / (f8 * 1e-3) / 1e12);
    }
    return 0;
}
