// Micro-benchmark / self-check of the structured Newton solve (trep_amd/csrc/bbd.hpp, gj_bbd) against gj_panel on KKT systems
// with the puppet's pattern: 6 trunk configs, four blocks of four configs, six constraints (two on the trunk alone, two on
// trunk + a whole block, two on trunk + three configs of a block).  Same harness as gj_bench.hip: one 64-lane workgroup per
// system, the rollout kernel's occupancy (19.5 KB of LDS -> 8 per CU), `reps` solves per wave.
//   tools/micro/build.sh && tools/micro/bin/bbd_bench [reps] [waves per CU]
// Also prints how many workgroups of a given dynamic-LDS size the runtime puts on a CU (the LDS allocation granule decides whether
// the plan tables fit next to the 19 520 B slice).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#if defined(TG_BBD_STAMPS)
namespace tg { __device__ long long tg_bbd_stamps[8]; }
#endif
#include "mvi_core.hpp"
#include "bbd.hpp"

constexpr int NF = 28, ND = 22, LD = 29;
#ifndef BB_NG
#define BB_NG 4
#define BB_NB 7
#define BB_T 12
#endif
struct TVar { __host__ __device__ constexpr int operator[](int i) const { return i < 6 ? i : 16 + i; } };   // trunk 0..5, constraints 22..27

template <int VARIANT>
__global__ __launch_bounds__(64, 2) void k_solve(int reps, const double *A_in, const int *plan_tab, double *x_out, int *ok_out, long long *cycles) {
#if defined(__HIP_DEVICE_COMPILE__)
    double *lds = tg_lds_base();
    const int lane = threadIdx.x;
    const double *src = A_in + (size_t)(blockIdx.x % 64) * NF * (NF + 1);
    double *pristine = lds + NF * LD, *scratch = lds + 2 * NF * LD;
    int *tab = (int *)(scratch + 256);
    for (int e = lane; e < NF * (NF + 1); e += 64) pristine[(e / (NF + 1)) * LD + e % (NF + 1)] = src[e];
    for (int e = lane; e < 128; e += 64) tab[e] = plan_tab[e];
    __syncthreads();
    bool ok = true;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        for (int e = lane; e < NF * LD; e += 64) lds[e] = pristine[e];
        __syncthreads();
        if (VARIANT == 0) ok &= tg::Core<64>::gj_panel<28, false>(true, lds, NF, LD, lane, scratch, nullptr);
        else ok &= tg::gj_bbd<NF, LD, BB_NG, BB_NB, BB_T>(lds, tg::bbd_rows<BB_NG + BB_NB>(tab, lane), scratch, lane, TVar{});
        __syncthreads();
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (blockIdx.x < 64) {
        if (lane < NF) x_out[blockIdx.x * 32 + lane] = lds[lane * LD + NF];
        if (lane == 0) { cycles[blockIdx.x] = t1 - t0; ok_out[blockIdx.x] = ok ? 1 : 0; }
    }
#endif
}

__global__ void k_empty() {}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int per_cu = argc > 2 ? std::atoi(argv[2]) : 8;
    // ---- LDS granule probe
    for (int bytes : {19520, 19776, 19968, 20032, 20224, 20480}) {
        int nblk = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void *>(&k_solve<1>), 64, (size_t)bytes);
        printf("dynamic LDS %d B -> %d workgroups of 64 per CU\n", bytes, nblk);
    }
    // ---- pattern and plan
    std::vector<unsigned char> pat(NF * NF, 0);
    auto blk = [](int i) { return i < 6 ? -1 : (i - 6) / 4; };
    for (int i = 0; i < ND; i++)
        for (int j = 0; j < ND; j++) pat[i * NF + j] = (blk(i) < 0 || blk(j) < 0 || blk(i) == blk(j)) ? 1 : 0;
    for (int c = 0; c < 6; c++) {
        const int row = ND + c;
        for (int j = 0; j < 6; j++) pat[row * NF + j] = pat[j * NF + row] = 1;
        if (c >= 2) for (int m = 0; m < (c < 4 ? 4 : 3); m++) { const int j = 6 + 4 * (c - 2) + m; pat[row * NF + j] = pat[j * NF + row] = 1; }
    }
    const tg::BbdPlan plan = tg::bbd_plan(NF, ND, pat);
    printf("plan ok %d: groups %d, own %d, border %d, trailing %d; trailing variables:", plan.ok, plan.g, plan.ng, plan.nb, plan.t);
    for (int i = 0; i < plan.t; i++) printf(" %d", plan.tvar[i]);
    printf("\n");
    if (!plan.ok || plan.ng != BB_NG || plan.nb != BB_NB || plan.t != BB_T) { printf("unexpected plan\n"); return 1; }
    for (int i = 0; i < plan.t; i++) if (plan.tvar[i] != TVar{}[i]) { printf("unexpected trailing order\n"); return 1; }
    // ---- matrices: -M/dt + small unsymmetric part, M SPD with the pattern; Dh entries O(1)
    std::vector<double> A(64 * (size_t)NF * (NF + 1), 0.0);
    srand(4242);
    auto rnd = []() { return 2.0 * rand() / RAND_MAX - 1.0; };
    const double dt = 0.01;
    for (int m = 0; m < 64; m++) {
        double *M = &A[(size_t)m * NF * (NF + 1)];
        // M = sum over "bodies" of v v^T with v supported on trunk + one block (or the trunk alone)
        std::vector<double> mass(ND * ND, 0.0);
        for (int b = 0; b < 10; b++) {
            const int limb = b < 2 ? -1 : (b - 2) / 2;
            for (int rep = 0; rep < 6; rep++) {
                std::vector<double> v(ND, 0.0);
                for (int i = 0; i < ND; i++) if (blk(i) < 0 || blk(i) == limb) v[i] = rnd();
                for (int i = 0; i < ND; i++) for (int j = 0; j < ND; j++) mass[i * ND + j] += 0.3 * v[i] * v[j];
            }
        }
        for (int i = 0; i < ND; i++)
            for (int j = 0; j < ND; j++) M[i * (NF + 1) + j] = pat[i * NF + j] ? -mass[i * ND + j] / dt + 0.05 * rnd() : 0.0;
        for (int c = ND; c < NF; c++)
            for (int j = 0; j < ND; j++) if (pat[c * NF + j]) { const double d = rnd(); M[c * (NF + 1) + j] = d + 1e-3 * rnd(); M[j * (NF + 1) + c] = -(d + 1e-3 * rnd()); }
        for (int i = 0; i < NF; i++) M[i * (NF + 1) + NF] = rnd() * (i < ND ? 1.0 : 1e-3);
    }
    double *dA, *dx; int *dp, *dtab; long long *dc;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dx, 64 * 32 * 8 * 2); hipMalloc(&dp, 64 * 4 * 2); hipMalloc(&dc, 64 * 8 * 2); hipMalloc(&dtab, 128 * 4);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtab, plan.tab, 128 * 4, hipMemcpyHostToDevice);
    const size_t lds = per_cu >= 8 ? 19520 : 160 * 1024 / per_cu - 512;
    const int grid = 256 * per_cu * 2;
    std::vector<double> x(2 * 64 * 32); std::vector<int> okv(2 * 64); std::vector<long long> cyc(2 * 64);
    for (int variant = 0; variant < 2; variant++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL((k_solve<0>), dim3(grid), dim3(64), lds, 0, reps, dA, dtab, dx, dp, dc);
            else hipLaunchKernelGGL((k_solve<1>), dim3(grid), dim3(64), lds, 0, reps, dA, dtab, dx + 64 * 32, dp + 64, dc + 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(cyc.data(), dc, cyc.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(x.data(), dx, x.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(okv.data(), dp, okv.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0.0; int bad = 0;
        for (int m = 0; m < 64; m++) {
            const double *M = &A[(size_t)m * NF * (NF + 1)], *xm = &x[(variant * 64 + m) * 32];
            for (int i = 0; i < NF; i++) {
                double s = -M[i * (NF + 1) + NF], sc = std::fabs(M[i * (NF + 1) + NF]);
                for (int j = 0; j < NF; j++) { s += M[i * (NF + 1) + j] * xm[j]; sc = std::fmax(sc, std::fabs(M[i * (NF + 1) + j] * xm[j])); }
                worst = std::fmax(worst, std::fabs(s) / (sc + 1e-300));
            }
            bad += okv[variant * 64 + m] ? 0 : 1;
        }
        double cavg = 0; for (int m = 0; m < 64; m++) cavg += (double)cyc[variant * 64 + m] / 64.0;
        printf("variant %d (%s): %d waves/CU  %.3f ms  -> %.0f ns per solve per resident wave; s_memtime ticks/solve %.0f; worst scaled residual %.2e; not-ok %d\n",
               variant, variant == 0 ? "gj_panel" : "gj_bbd", per_cu, best, best * 1e6 / (2.0 * reps), cavg / reps, worst, bad);
    }
#if defined(TG_BBD_STAMPS)
    {   // build with EXTRA=-DTG_BBD_STAMPS: cycles of workgroup 0 between the stages of gj_bbd (3 timed launches x reps solves)
        long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(tg::tg_bbd_stamps), sizeof(st));
        const char *names[6] = {"stage 0: tables, rows -> registers", "stage 1: own columns", "stage 2: Schur updates (LDS atomics)", "stage 2: trailing rows + U", "stage 2: trailing elimination", "stage 3: back-substitution, stores"};
        for (int i = 0; i < 6; i++) printf("  %-40s %8.0f cycles per solve\n", names[i], (double)st[i] / (3.0 * reps));
    }
#endif
    double dmax = 0.0;
    for (int m = 0; m < 64; m++) for (int i = 0; i < NF; i++)
        dmax = std::fmax(dmax, std::fabs(x[m * 32 + i] - x[(64 + m) * 32 + i]) / (1e-300 + std::fabs(x[m * 32 + i])));
    printf("max relative difference of the two solutions %.2e\n", dmax);
    return 0;
}
