// Micro-benchmark of the Newton-system solver of the rollout kernel in isolation: n x n system + one right-hand side per
// wavefront, the occupancy of the real kernel (one 64-lane workgroup per trajectory, 19.5 KB of LDS each -> 8 per CU,
// two waves per SIMD), `reps` solves per wave.  Variants:
//   0  gj_rows<N>      (mvi_core.hpp: one row per lane, pivot row broadcast with v_readlane, 28 pivot steps)
//   1  gj_panel<N>     (mvi_core.hpp: matrix in the v_mfma_f64_16x16x4 accumulator layout on all 64 lanes, panels of four
//                       columns factored one row per lane, trailing update on the matrix cores)
// Prints ns per solve (wall, per wave slot), cycles at 2.4 GHz, and the residual / difference of the solutions.
//   tools/micro/build.sh && tools/micro/bin/gj_bench [n] [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mvi_core.hpp"

#ifndef GJ_N
#define GJ_N 28
#endif

template <int VARIANT, bool TRACE>
__global__ __launch_bounds__(64, 2) void k_solve(int n_arg, int ld_arg, int reps, const double *A_in, double *x_out, int *piv_out, long long *cycles) {
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(GJ_STATIC_N)     // like the system-specialised kernel: sizes are compile-time constants
    constexpr int n = GJ_N, ld = (GJ_N + 1) | 1;
#else
    const int n = n_arg, ld = ld_arg;
#endif
    double *lds = tg_lds_base();
    const int lane = threadIdx.x;
    const double *src = A_in + (size_t)(blockIdx.x % 64) * n * (n + 1);
    int *trace = (int *)(lds + 2 * n * ld);
    // pristine copy in the second half: every repetition solves the same system
    for (int e = lane; e < n * (n + 1); e += 64) lds[n * ld + (e / (n + 1)) * ld + e % (n + 1)] = src[e];
    if (lane < 32) trace[lane] = -1;
    __syncthreads();
    bool ok = true;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        for (int e = lane; e < n * ld; e += 64) lds[e] = lds[n * ld + e];
        __syncthreads();
        if (VARIANT == 0) ok &= tg::Core<64>::gj_rows<GJ_N, TRACE>(true, lds, n, ld, lane, trace);
        else ok &= tg::Core<64>::gj_panel<GJ_N, TRACE>(true, lds, n, ld, lane, lds + 2 * n * ld + 32, trace);
        __syncthreads();
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (blockIdx.x < 64) {
        if (lane < n) { x_out[blockIdx.x * 32 + lane] = lds[lane * ld + n]; piv_out[blockIdx.x * 32 + lane] = trace[lane]; }
        if (lane == 0) { cycles[blockIdx.x] = t1 - t0; piv_out[blockIdx.x * 32 + 31] = ok ? 0 : 1; }
    }
#endif
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : GJ_N, reps = argc > 2 ? std::atoi(argv[2]) : 2000;
    const int ld = (n + 1) | 1;
    // workgroups resident per CU: 8 (two waves per SIMD, the rollout kernel's occupancy) or 4 (one wave per SIMD: the solver's own latency)
    const int per_cu = argc > 3 ? std::atoi(argv[3]) : 8;
    const int grid = 256 * per_cu * 2;   // two full rounds
    std::vector<double> A(64 * (size_t)n * (n + 1));
    srand(12345);
    auto rnd = []() { return 2.0 * rand() / RAND_MAX - 1.0; };
    for (int m = 0; m < 64; m++) {
        double *M = &A[(size_t)m * n * (n + 1)];
        // KKT-like: diagonally heavy leading block, a few "constraint" rows / columns with zeros on the diagonal
        const int nc = n >= 8 ? 6 : 0, nd = n - nc;
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= n; j++) {
                double v = 0.3 * rnd();
                if (i < nd && j < nd && i == j) v += 2.0 + rnd();
                if (i >= nd && j >= nd && j < n) v = 0.0;
                if ((i >= nd || (j >= nd && j < n)) && (rand() % 3)) v = 0.0;
                M[i * (n + 1) + j] = v * (m % 4 == 3 && i % 5 == 0 ? 1e3 : 1.0);
            }
        for (int i = nd; i < n; i++) { M[i * (n + 1) + (i - nd) * 3 % nd] = 1.0 + 0.1 * rnd(); M[((i - nd) * 3 % nd) * (n + 1) + i] = -1.0 + 0.1 * rnd(); }
    }
    double *dA, *dx; int *dp; long long *dc;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dx, 64 * 32 * 8 * 2); hipMalloc(&dp, 64 * 32 * 4 * 2); hipMalloc(&dc, 64 * 8 * 2);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = per_cu >= 8 ? 19520 : 160 * 1024 / per_cu - 512;    // 19520 B = the puppet's LDS slice: 8 workgroups per CU
    if (2 * n * ld * 8 + 128 + 256 + 1024 > (int)lds) { printf("n too large for the benchmark's LDS slice\n"); return 1; }
    std::vector<double> x(2 * 64 * 32); std::vector<int> piv(2 * 64 * 32); std::vector<long long> cyc(2 * 64);
    for (int variant = 0; variant < 2; variant++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0);
            if (lds > 64 * 1024) { hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
            if (variant == 0) hipLaunchKernelGGL((k_solve<0, false>), dim3(grid), dim3(64), lds, 0, n, ld, reps, dA, dx, dp, dc);
            else hipLaunchKernelGGL((k_solve<1, false>), dim3(grid), dim3(64), lds, 0, n, ld, reps, dA, dx + 64 * 32, dp + 64 * 32, dc + 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(cyc.data(), dc, cyc.size() * 8, hipMemcpyDeviceToHost);
        // one traced solve for the pivot rows and the solution
        if (variant == 0) hipLaunchKernelGGL((k_solve<0, true>), dim3(64), dim3(64), 19520, 0, n, ld, 1, dA, dx, dp, dc);
        else hipLaunchKernelGGL((k_solve<1, true>), dim3(64), dim3(64), 19520, 0, n, ld, 1, dA, dx + 64 * 32, dp + 64 * 32, dc + 64);
        hipDeviceSynchronize();
        hipMemcpy(x.data(), dx, x.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(piv.data(), dp, piv.size() * 4, hipMemcpyDeviceToHost);
        // residual of every one of the 64 systems
        double worst = 0.0; int bad = 0;
        for (int m = 0; m < 64; m++) {
            const double *M = &A[(size_t)m * n * (n + 1)], *xm = &x[(variant * 64 + m) * 32];
            double rn = 0.0, bn = 0.0;
            for (int i = 0; i < n; i++) {
                double s = -M[i * (n + 1) + n], sc = 0.0;
                for (int j = 0; j < n; j++) { s += M[i * (n + 1) + j] * xm[j]; sc = std::fmax(sc, std::fabs(M[i * (n + 1) + j] * xm[j])); }
                rn = std::fmax(rn, std::fabs(s) / (sc + 1e-300)); bn += 1;
            }
            worst = std::fmax(worst, rn);
            bad += piv[(variant * 64 + m) * 32 + 31];
        }
        double cavg = 0; for (int m = 0; m < 64; m++) cavg += (double)cyc[variant * 64 + m] / 64.0;
        // grid = 2 rounds of 2048 resident waves -> wall time of one solve of a resident wave = ms / (2 * reps)
        printf("variant %d (%s): %d waves/CU n=%d  %.3f ms  -> %.0f ns per solve per resident wave (%.0f cycles @2.4GHz); s_memtime ticks/solve %.0f; worst scaled residual %.2e; not-ok %d\n",
               variant, variant == 0 ? "gj_rows" : "gj_panel", per_cu, n, best, best * 1e6 / (2.0 * reps), best * 1e6 / (2.0 * reps) * 2.4, cavg / reps, worst, bad);
    }
    double dmax = 0.0; int pdiff = 0;
    for (int m = 0; m < 64; m++) for (int i = 0; i < n; i++) {
        dmax = std::fmax(dmax, std::fabs(x[m * 32 + i] - x[(64 + m) * 32 + i]) / (1e-300 + std::fabs(x[m * 32 + i])));
        pdiff += piv[m * 32 + i] != piv[(64 + m) * 32 + i];
    }
    printf("max relative difference of the two solutions %.2e; pivot rows that differ %d of %d\n", dmax, pdiff, 64 * n);
    return 0;
}
