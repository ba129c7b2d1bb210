// dopt.hip -- device-side primitives of the discrete trajectory optimiser (the direct caller of the
// MidpointVI path: reference trep/discopt/dlqr.py:9-81, dcost.py:5-118, doptimizer.py:249-506).
//
// One optimisation problem ("seed") = one trajectory X [N+1][nX], U [N][nU] with its linearisation
// A [N][nX][nX], B [N][nX][nU].  The seed axis is the parallel axis: every kernel below runs one
// workgroup per seed (the k axis of a Riccati / adjoint / tangent sweep is inherently serial), so
// S seeds occupy S CUs; the dense nX x nX work of one seed is spread over the 256 lanes of its
// workgroup with the matrices resident in LDS (P, A_k: 2 x 51 KB for the 40-DOF puppet, nX = 80).
// MI355X has the same fp64 rate on the vector and the matrix pipes and the operands here are
// 80 x 80 at most, so the products are register-tiled VALU code (4x4 tiles from LDS), not MFMA.
//
// All entry points take DEVICE pointers and run on the device's default stream, which is ordered
// with the (blocking) streams of the tg_batch objects.
#include <hip/hip_runtime.h>

#include <array>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "../../include/trep_amd.h"

namespace tg_detail {
int fail(int code, const std::string &msg);
}
// ROCm device-library wavefront reduction (DPP based)
extern "C" __device__ __attribute__((const)) unsigned int __ockl_wfred_max_u32(unsigned int);

namespace {

using tg_detail::fail;
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(TG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int LQ_T = 256;  // threads per workgroup of the sweep kernels
#if !defined(LQM_THREADS)
#define LQM_THREADS 512
#endif
constexpr int LQM_T = LQM_THREADS; // ... of the matrix-core sweep: two wavefronts per SIMD hide the 200-cycle latency of a dependent f64 MFMA

// Diagnostic build only (-DTG_PROFILE): cycles per phase of the LQ sweep, workgroup 0 thread 0, read by tg_lq_profile.
#if defined(TG_PROFILE)
__device__ long long g_lq_prof[8];
#define LQ_STAMP(i) do { if (blockIdx.x == 0 && tid == 0) { long long t_ = (long long)__builtin_amdgcn_s_memtime(); lq_acc[i] += t_ - lq_last; lq_last = t_; } } while (0)
#else
#define LQ_STAMP(i) ((void)0)
#endif

__host__ __device__ inline int round_up(int n, int m) { return (n + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------------
// Time-varying LQ backward sweep (dlqr.py:41-81; with q = r = S = 0 it is solve_tv_lqr, :9-38):
//   gamma = R_k + B'PB,  Kpart = B'PA + S_k',  [C_k | K_k] = gamma^-1 [B'b + r_k | Kpart],
//   b <- q_k - K'r_k + (A' - K'B') b,  P <- Q_k + A'PA - Kpart'K,  P <- (P + P')/2.
// Model weights: Q_k = Qc(+k stride) [+ HZ_k[0:nxh,0:nxh]], S_k = HZ_k[0:nxh, nxh:], R_k = Rc [+ HZ_k[nxh:,nxh:]]
// where HZ_k is the z-contracted second derivative of the dynamics (tg_batch_deriv2_contract_device),
// i.e. the Newton model of doptimizer.py:319-345 is assembled on the fly.
//
// P (later P A) and A_k live in LDS with the leading dimension padded to a multiple of the tile size TS
// (padding stays zero); every thread owns ONE TS x TS tile of the nX x nX products in registers, so the
// 256 threads cover nX <= 16*TS exactly (TS = 5 for the 40-DOF puppet: 16 x 16 tiles of 5 x 5).
// A_{k-1}, B_{k-1} are prefetched into registers (RI x CI / PB values per thread) while step k computes.
// ------------------------------------------------------------------------------------------------------
template <int TS, int RI, int CI, int PB>
__global__ __launch_bounds__(LQ_T) void k_tv_lq(const tg_lq_problem a) {
    extern __shared__ double lds[];
    __shared__ int s_sing, s_rowof[64];
    const int tid = threadIdx.x;
    const int s = a.select_dev ? a.select_dev[blockIdx.x] : blockIdx.x;
    const int nX = a.nX, nU = a.nU, N = a.horizon;
    const int ldx = round_up(nX, TS), nT = ldx / TS, ntiles = nT * nT;
    const int ldw = nU + 1 + ldx;  // [gamma | B'b + r | Kpart]
    double *Pm = lds, *Am = Pm + ldx * ldx, *Bm = Am + ldx * ldx, *BtP = Bm + ldx * nU, *Kp = BtP + nU * ldx;
    double *G = Kp + nU * ldx, *bv = G + nU * ldw, *bn = bv + ldx, *wv = bn + ldx, *rv = wv + nU, *fac = rv + nU;
    const int lds_doubles = (int)(fac + nU - lds);
    for (int i = tid; i < lds_doubles; i += LQ_T) lds[i] = 0.0;
    if (tid == 0) s_sing = 0;
    __syncthreads();

    const size_t sN = (size_t)s * N;
    const bool affine = a.q_dev != nullptr;
    const int nxh = a.hz_nx, hzR = a.hz_R;
    // terminal condition P_N = Qf, b_N = q_N -- or, for a sweep over the steps [k_begin, k_end) of the horizon (tg_lq_problem::k_*), the
    // (P, b) an earlier launch left at step k_end (its P0 / b0 outputs)
    const int kb = a.k_begin, ke = a.k_end > 0 ? a.k_end : N;
    if (a.Pt_dev) {
        const double *Pt = a.Pt_dev + (size_t)s * nX * nX;
        for (int e = tid; e < nX * nX; e += LQ_T) Pm[(e / nX) * ldx + e % nX] = Pt[e];
        if (affine && a.bt_dev) for (int i = tid; i < nX; i += LQ_T) bv[i] = a.bt_dev[(size_t)s * nX + i];
    } else {
        const double *Qf = a.Qf_dev + (size_t)s * a.Qf_seed_stride;
        for (int e = tid; e < nX * nX; e += LQ_T) Pm[(e / nX) * ldx + e % nX] = Qf[e];
        if (affine) for (int i = tid; i < nX; i += LQ_T) bv[i] = a.q_dev[(sN + s + N) * nX + i];
    }
    // prefetch mapping: 8 row groups x 32 columns per pass (256-byte row segments)
    const int pr0 = tid >> 5, pc0 = tid & 31;
    double preA[RI * CI], preB[PB];
    auto prefetch = [&](int k) {
        const double *Ak = a.A_dev + (sN + k) * (size_t)nX * nX, *Bk = a.B_dev + (sN + k) * (size_t)nX * nU;
#pragma unroll
        for (int ri = 0; ri < RI; ri++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) {
                const int r = pr0 + 8 * ri, c = pc0 + 32 * ci;
                if (r < nX && c < nX) preA[ri * CI + ci] = Ak[r * nX + c];
            }
#pragma unroll
        for (int i = 0; i < PB; i++) { const int e = tid + i * LQ_T; if (e < nX * nU) preB[i] = Bk[e]; }
    };
    auto commit = [&]() {
#pragma unroll
        for (int ri = 0; ri < RI; ri++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) {
                const int r = pr0 + 8 * ri, c = pc0 + 32 * ci;
                if (r < nX && c < nX) Am[r * ldx + c] = preA[ri * CI + ci];
            }
#pragma unroll
        for (int i = 0; i < PB; i++) { const int e = tid + i * LQ_T; if (e < nX * nU) Bm[e] = preB[i]; }
    };
    prefetch(ke - 1);
    commit();
    __syncthreads();

    const bool has_tile = tid < ntiles;
    const int i0 = TS * (tid / nT), j0 = TS * (tid % nT);
    double acc[TS * TS];
#if defined(TG_PROFILE)
    long long lq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lq_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int k = ke - 1; k >= kb; k--) {
        if (k > kb) prefetch(k - 1);
        const double *hz = a.hz_dev ? a.hz_dev + (sN + k) * (size_t)hzR * hzR : nullptr;
        if (a.b_next_dev && affine) for (int i = tid; i < nX; i += LQ_T) a.b_next_dev[(sN + k) * nX + i] = bv[i];    // b_{k+1}
        // ---- phase 1: PA tile (registers), B'P, B'b ----------------------------------------------------
        if (has_tile) {
#pragma unroll
            for (int e = 0; e < TS * TS; e++) acc[e] = 0.0;
#pragma unroll 2
            for (int m = 0; m < nX; m++) {
                double pv[TS], av[TS];
#pragma unroll
                for (int i = 0; i < TS; i++) pv[i] = Pm[m * ldx + i0 + i];   // P is symmetric: row m instead of column m (contiguous, no bank conflicts)
#pragma unroll
                for (int j = 0; j < TS; j++) av[j] = Am[m * ldx + j0 + j];
#pragma unroll
                for (int i = 0; i < TS; i++)
#pragma unroll
                    for (int j = 0; j < TS; j++) acc[TS * i + j] += pv[i] * av[j];
            }
        }
        for (int o = tid; o < nU * nT; o += LQ_T) {  // B'P, 1 x TS tiles
            const int u = o / nT, c0 = TS * (o % nT);
            double c[TS];
#pragma unroll
            for (int j = 0; j < TS; j++) c[j] = 0.0;
#pragma unroll 4
            for (int i = 0; i < nX; i++) {
                const double bb = Bm[i * nU + u];
#pragma unroll
                for (int j = 0; j < TS; j++) c[j] += bb * Pm[i * ldx + c0 + j];
            }
#pragma unroll
            for (int j = 0; j < TS; j++) BtP[u * ldx + c0 + j] = c[j];
        }
        if (affine && tid < nU) {
            double w = 0.0;
            for (int i = 0; i < nX; i++) w += Bm[i * nU + tid] * bv[i];
            wv[tid] = w;
            rv[tid] = a.r_dev[(sN + k) * nU + tid];
        }
        __syncthreads();
        LQ_STAMP(0);
        // ---- phase 2: PA -> LDS (over P), gamma, Kpart = (B'P) A + S' ------------------------------------
        if (has_tile) {
#pragma unroll
            for (int i = 0; i < TS; i++)
#pragma unroll
                for (int j = 0; j < TS; j++) Pm[(i0 + i) * ldx + j0 + j] = acc[TS * i + j];
        }
        for (int o = tid; o < nU * nU; o += LQ_T) {
            const int u = o / nU, v = o % nU;
            double g = a.R_dev[(size_t)s * a.R_seed_stride + (size_t)k * a.R_step_stride + o];
            if (hz) g += hz[(size_t)(nxh + u) * hzR + nxh + v];
#pragma unroll 4
            for (int j = 0; j < nX; j++) g += BtP[u * ldx + j] * Bm[j * nU + v];
            G[u * ldw + v] = g;
        }
        if (tid < nU) { const double rw = affine ? wv[tid] + rv[tid] : 0.0; rv[tid] = rw; G[tid * ldw + nU] = rw; }   // r_k + B'b
        for (int o = tid; o < nU * nT; o += LQ_T) {
            const int u = o / nT, c0 = TS * (o % nT);
            double c[TS];
#pragma unroll
            for (int j = 0; j < TS; j++) c[j] = (hz && c0 + j < nxh) ? hz[(size_t)(c0 + j) * hzR + nxh + u] : 0.0;
#pragma unroll 4
            for (int i = 0; i < nX; i++) {
                const double bp = BtP[u * ldx + i];
#pragma unroll
                for (int j = 0; j < TS; j++) c[j] += bp * Am[i * ldx + c0 + j];
            }
#pragma unroll
            for (int j = 0; j < TS; j++) { Kp[u * ldx + c0 + j] = c[j]; G[u * ldw + nU + 1 + c0 + j] = c[j]; }
        }
        __syncthreads();
        LQ_STAMP(1);
        // ---- phase 3: [C | K] = gamma^-1 [.|.]: Gauss-Jordan with partial pivoting, rows never move ------------
        // Per pivot ONE phase: every wavefront finds the pivot of column p itself (lane i holds |G[i][p]| of the
        // rows not used yet -- the rows LAPACK's getrf searches -- and a wave max picks the row), then thread
        // (half h, column c) eliminates column p+1+c in its half of the rows with all loads ahead of the stores.
        // Rows are normalised and put in variable order at the end (into the dead B'P buffer and `wv`).
        {
            const int lane = tid & 63, half = tid >> 7, col = tid & 127;
            const int h0 = half ? (nU + 1) / 2 : 0, h1 = half ? nU : (nU + 1) / 2;
            unsigned long long used = 0ull;
            for (int p = 0; p < nU; p++) {
                // the magnitude only ranks candidates: single precision with the lane in the low mantissa bits, one
                // 32-bit wave max (DPP) instead of a shuffle butterfly
                float mf = 0.0f;
                if (lane < nU && !((used >> lane) & 1ull)) mf = (float)fabs(G[lane * ldw + p]);
                const unsigned int key = __ockl_wfred_max_u32((__float_as_uint(mf) & ~0x3Fu) | (unsigned int)(63 - lane));
                const int r = 63 - (int)(key & 0x3Fu);
                if (!(__uint_as_float(key & ~0x3Fu) > 0.0f)) { if (tid == 0) s_sing = 1; }
                used |= 1ull << r;
                if (tid == 0) s_rowof[p] = r;
                const double piv = G[r * ldw + p];
                double inv = __builtin_amdgcn_rcp(piv);           // hardware seed + two Newton steps: the multipliers
                inv = fma(inv, fma(-piv, inv, 1.0), inv);         // need not be correctly rounded
                inv = fma(inv, fma(-piv, inv, 1.0), inv);
                for (int c = p + 1 + col; c < ldw; c += 128) {
                    const double pk = G[r * ldw + c];
                    for (int i0 = h0; i0 < h1; i0 += 4) {
                        double l[4], a4[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int i = i0 + q < h1 ? i0 + q : h1 - 1;
                            l[q] = G[i * ldw + p]; a4[q] = G[i * ldw + c];
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int i = i0 + q;
                            if (i < h1 && i != r) G[i * ldw + c] = fma(-(l[q] * inv), pk, a4[q]);
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (tid < nU) fac[tid] = 1.0 / G[s_rowof[tid] * ldw + tid];
        __syncthreads();
        double *Ks = BtP, *Cs = wv;   // K_k [nU][ldx], C_k [nU]
        for (int o = tid; o < nU * ldx; o += LQ_T) {
            const int u = o / ldx, j = o % ldx;
            Ks[o] = G[s_rowof[u] * ldw + nU + 1 + j] * fac[u];
        }
        if (tid < nU) Cs[tid] = G[s_rowof[tid] * ldw + nU] * fac[tid];
        __syncthreads();
        LQ_STAMP(2);
        // ---- phase 4: outputs K_k, C_k; new P tile (registers), new b ----------------------------------------
        {
            double *Ko = a.K_dev + (sN + k) * (size_t)nU * nX;
            for (int o = tid; o < nU * nX; o += LQ_T) Ko[o] = Ks[(o / nX) * ldx + o % nX];
            if (a.C_dev && tid < nU) a.C_dev[(sN + k) * nU + tid] = Cs[tid];
        }
        if (has_tile) {
            const double *Qk = a.Q_dev + (size_t)s * a.Q_seed_stride + (size_t)k * a.Q_step_stride;
#pragma unroll
            for (int i = 0; i < TS; i++)
#pragma unroll
                for (int j = 0; j < TS; j++) {
                    const int r = i0 + i, c = j0 + j;
                    double q0 = (r < nX && c < nX) ? Qk[(size_t)r * nX + c] : 0.0;
                    if (hz && r < nxh && c < nxh) q0 += hz[(size_t)r * hzR + c];
                    acc[TS * i + j] = q0;
                }
#pragma unroll 2
            for (int m = 0; m < nX; m++) {  // + A' (PA)
                double av[TS], pv[TS];
#pragma unroll
                for (int i = 0; i < TS; i++) av[i] = Am[m * ldx + i0 + i];
#pragma unroll
                for (int j = 0; j < TS; j++) pv[j] = Pm[m * ldx + j0 + j];
#pragma unroll
                for (int i = 0; i < TS; i++)
#pragma unroll
                    for (int j = 0; j < TS; j++) acc[TS * i + j] += av[i] * pv[j];
            }
            for (int u = 0; u < nU; u++) {  // - Kpart' K
                double kv[TS], gv[TS];
#pragma unroll
                for (int i = 0; i < TS; i++) kv[i] = Kp[u * ldx + i0 + i];
#pragma unroll
                for (int j = 0; j < TS; j++) gv[j] = Ks[u * ldx + j0 + j];
#pragma unroll
                for (int i = 0; i < TS; i++)
#pragma unroll
                    for (int j = 0; j < TS; j++) acc[TS * i + j] -= kv[i] * gv[j];
            }
        }
        if (affine) for (int i = tid; i < nX; i += LQ_T) {
            double v = a.q_dev[(sN + s + k) * nX + i];
            for (int m = 0; m < nX; m++) v += Am[m * ldx + i] * bv[m];
            for (int u = 0; u < nU; u++) v -= Ks[u * ldx + i] * rv[u];
            bn[i] = v;
        }
        __syncthreads();
        LQ_STAMP(3);
        // ---- phase 5: P <- new tile, b <- new b, next A, B into LDS ---------------------------------------------
        if (has_tile) {
#pragma unroll
            for (int i = 0; i < TS; i++)
#pragma unroll
                for (int j = 0; j < TS; j++) Pm[(i0 + i) * ldx + j0 + j] = acc[TS * i + j];
        }
        if (affine) for (int i = tid; i < nX; i += LQ_T) bv[i] = bn[i];
        if (k > kb) commit();
        __syncthreads();
        LQ_STAMP(4);
        // ---- phase 6: P <- (P + P')/2, one thread per unordered pair --------------------------------------------
        for (int e = tid; e < nX * nX; e += LQ_T) {
            const int i = e / nX, j = e % nX;
            if (i < j) {
                const double v = 0.5 * (Pm[i * ldx + j] + Pm[j * ldx + i]);
                Pm[i * ldx + j] = v; Pm[j * ldx + i] = v;
            }
        }
        __syncthreads();
        LQ_STAMP(5);
    }
#if defined(TG_PROFILE)
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 8; i++) g_lq_prof[i] = lq_acc[i];
#endif
    if (a.P0_dev) for (int e = tid; e < nX * nX; e += LQ_T) a.P0_dev[(size_t)s * nX * nX + e] = Pm[(e / nX) * ldx + e % nX];
    if (a.b0_dev && affine) for (int i = tid; i < nX; i += LQ_T) a.b0_dev[(size_t)s * nX + i] = bv[i];
    if (a.status_dev && tid == 0) a.status_dev[s] = (s_sing || (a.Pt_dev && a.status_dev[s] != TG_OK)) ? TG_SINGULAR : TG_OK;      // (a later chunk of a chunked sweep keeps an earlier chunk's verdict)
}

// ------------------------------------------------------------------------------------------------------
// The same sweep on the matrix cores.
//
// k_tv_lq above is bound by LDS latency: its products are register-tiled VALU loops (10 LDS reads per 25 FMAs) run by
// ONE wavefront per SIMD (153 KB of LDS = one workgroup per CU), and its nU x nU solve is 18 barrier-separated LDS
// passes: 206 k cycles per Riccati step for the 40-DOF puppet where the flops alone need about 20 k.  Here
//   * every product -- P A, P B, B'(P B), B'(P A), A'(P A), Kpart' K -- is a chain of v_mfma_f64_16x16x4_f64 over
//     16 x 16 output tiles (wave w owns tiles w, w+4, ...): two LDS reads feed 2048 flops instead of 5, and all
//     operand reads are row segments of 16 consecutive doubles (P is symmetric, so P' rows serve as P columns;
//     P B is stored instead of B'P for the same reason) -- no bank conflicts, no transposes;
//   * the solve [C | K] = gamma^-1 [B'b + r | Kpart] runs in registers without a single barrier: each wavefront
//     takes the nU columns of gamma plus its quarter of the 1 + nX right-hand-side columns, one column per lane,
//     rows in registers; every wave repeats the (identical) pivot search on its own copy of gamma.  LAPACK's pivot
//     rule (largest magnitude in the column, rows never move) as before.
// Same arithmetic up to summation order; tests/test_gpu_discopt_device.py holds both kernels to the numpy sweep.
// ------------------------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));

// One wavefront: Gauss-Jordan on [gamma | its slice of the right-hand sides], one column per lane, NR >= nU rows in
// registers.  G [nU][ldw] = [gamma | rhs(1 + nX)] is only READ (every wave loads gamma and its own slice); solutions go
// to Cs [nU] (rhs 0) and Ks [nU][ldx] (rhs 1 + j), in variable order.  scr: 3 * NR doubles of per-wave LDS scratch.
template <int NR>
__device__ __forceinline__ void lq_gj_wave(const double *G_generic, int ldw, int nU, int rhs_lo, int rhs_n, double *Ks_generic, int ldx,
                                        double *Cs_generic, double *scr_generic, int lane, int *sing) {
    typedef __attribute__((address_space(3))) double lds_double;
    const lds_double *G = (const lds_double *)G_generic;
    lds_double *Ks = (lds_double *)Ks_generic, *Cs = (lds_double *)Cs_generic, *colbuf = (lds_double *)scr_generic, *dinv = colbuf + NR;
    __attribute__((address_space(3))) int *var = (__attribute__((address_space(3))) int *)(colbuf + 2 * NR);
    const bool is_rhs = lane >= nU && lane < nU + rhs_n;
    const int gcol = lane < nU ? lane : nU + rhs_lo + (lane - nU);       // column of G this lane holds
    const bool have = lane < nU || is_rhs;
    double a[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) a[i] = (have && i < nU) ? G[i * ldw + gcol] : 0.0;
    unsigned int used = 0u;
    bool ok = true;
    for (int k = 0; k < nU; k++) {
        if (lane == k) {
#pragma unroll
            for (int i = 0; i < NR; i++) colbuf[i] = a[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        float mf = 0.0f;
        if (lane < nU && !((used >> lane) & 1u)) mf = (float)fabs(colbuf[lane]);
        const unsigned int key = __ockl_wfred_max_u32((__float_as_uint(mf) & ~0x3Fu) | (unsigned int)(63 - lane));
        const int r = __builtin_amdgcn_readfirstlane(63 - (int)(key & 0x3Fu));
        if (!(__uint_as_float(key & ~0x3Fu) > 0.0f)) ok = false;
        used |= 1u << r;
        // the whole pivot column in registers first (wave-uniform addresses: LDS broadcasts, all in flight together); a
        // load inside a `(i == r) ? ... : ...` arm becomes a scalar branch with its own s_waitcnt per row
        double cb[NR];
#pragma unroll
        for (int i = 0; i < NR; i++) cb[i] = colbuf[i];
        // this lane's pivot-row entry and the pivot: r is wave-uniform but not a compile-time register index.  A chain of
        // uniform branches picks them (a 0/1-weighted FMA sum would be NR dependent fp64 FMAs, ~30 cycles each)
        double p = 0.0, piv = 1.0;
#pragma unroll
        for (int i = 0; i < NR; i++) if (i == r) { p = a[i]; piv = cb[i]; }
        double inv = __builtin_amdgcn_rcp(piv);            // seed + one cubic refinement: three dependent fp64 operations
        { const double e = fma(-piv, inv, 1.0); inv = fma(inv, fma(e, e, e), inv); }
        if (lane == 0) { var[r] = k; dinv[r] = inv; }
        const double q = ok ? -(inv * p) : 0.0;    // a_i <- a_i - (c_i / pivot) p for the other rows
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const double upd = fma(cb[i], q, a[i]);
            a[i] = (i == r) ? a[i] : upd;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    if (!ok && lane == 0) *sing = 1;
    if (is_rhs) {
        const int g = rhs_lo + (lane - nU);
#pragma unroll
        for (int i = 0; i < NR; i++) {
            if (i < nU) {
                const int v = var[i];
                const double x = a[i] * dinv[i];
                if (g == 0) Cs[v] = x; else Ks[v * ldx + g - 1] = x;
            }
        }
    }
}

// The same solve with one matrix ROW per lane (lanes 0 .. NR-1) and the columns in registers: NR columns of gamma plus SL right-hand-side
// columns of this wave's slice.  A pivot step is then: one 32-lane DPP max over |a_ik| (LAPACK's rule: largest magnitude of the
// column among the rows not used yet; near ties within 2^-17 go to the lower row), the pivot row broadcast with v_readlane (two
// entries ahead of their two FMAs), one FMA per remaining column -- no LDS traffic, no dynamic register index (the column-per-lane
// variant above needs the pivot ROW out of a register array and pays ~2.3 k cycles per step for it).  Rows / columns nU .. NR-1 are
// an identity block, so all NR steps run without a branch on nU.  ~4x faster than lq_gj_wave on the 18 x 18 puppet systems.
template <int NR, int SL>
__device__ __forceinline__ void lq_gj_rows(const double *G_generic, int ldw, int nU, int rhs_lo, int rhs_n, double *Ks_generic, int ldx,
                                           double *Cs_generic, int lane, int *sing) {
    typedef __attribute__((address_space(3))) double lds_double;
    const lds_double *G = (const lds_double *)G_generic;
    lds_double *Ks = (lds_double *)Ks_generic, *Cs = (lds_double *)Cs_generic;
    const bool mine = lane < nU, ident = lane >= nU && lane < NR;
    double m[NR], b[SL];
#pragma unroll
    for (int j = 0; j < NR; j++) m[j] = mine ? (j < nU ? G[lane * ldw + j] : 0.0) : ((ident && j == lane) ? 1.0 : 0.0);
#pragma unroll
    for (int c = 0; c < SL; c++) b[c] = (mine && c < rhs_n) ? G[lane * ldw + nU + rhs_lo + c] : 0.0;
    int mycol = -1;
    double diag = 1.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const bool free_row = (mine || ident) && mycol < 0;
        const float cand = free_row ? (float)fabs(m[k]) : 0.0f;
        unsigned int key = (__float_as_uint(cand) & ~0x3Fu) | (unsigned int)(63 - lane);
        // max over each 16-lane row (DPP row_shr 1, 2, 4, 8), then over the two rows that hold matrix rows
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true));
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x112, 0xf, 0xf, true));
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x114, 0xf, 0xf, true));
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x118, 0xf, 0xf, true));
        const unsigned int best = max((unsigned int)__builtin_amdgcn_readlane((int)key, 15), (unsigned int)__builtin_amdgcn_readlane((int)key, 31));
        const int src = 63 - (int)(best & 0x3Fu);
        if (!(__uint_as_float(best & ~0x3Fu) > 0.0f)) ok = false;
        auto bcast = [&](double v) -> double {
            const long long w = __double_as_longlong(v);
            const int lo = __builtin_amdgcn_readlane((int)(w & 0xFFFFFFFFLL), src), hi = __builtin_amdgcn_readlane((int)(w >> 32), src);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
        const bool is_piv = lane == src;
        const double piv = bcast(m[k]);
        double inv = __builtin_amdgcn_rcp(piv);            // seed + one cubic refinement (the multipliers need not be correctly rounded)
        { const double e = fma(-piv, inv, 1.0); inv = fma(inv, fma(e, e, e), inv); }
        const double l = (ok && !is_piv) ? m[k] * inv : 0.0;
#pragma unroll
        for (int j = k + 1; j < NR; j += 2) {
            const double p0 = bcast(m[j]), p1 = bcast(m[j + 1 < NR ? j + 1 : j]);
            m[j] = fma(-l, p0, m[j]);
            if (j + 1 < NR) m[j + 1] = fma(-l, p1, m[j + 1]);
        }
#pragma unroll
        for (int c = 0; c < SL; c += 2) {
            const double p0 = bcast(b[c]), p1 = bcast(b[c + 1 < SL ? c + 1 : c]);
            b[c] = fma(-l, p0, b[c]);
            if (c + 1 < SL) b[c + 1] = fma(-l, p1, b[c + 1]);
        }
        if (is_piv) { mycol = k; diag = m[k]; }
    }
    if (!ok && lane == 0) *sing = 1;
    if (mine && mycol >= 0 && mycol < nU) {       // row `lane` was the pivot of variable mycol: its right-hand sides / pivot are that variable's solution
        const double dinv = 1.0 / diag;
#pragma unroll
        for (int c = 0; c < SL; c++) {
            if (c < rhs_n) {
                const int g = rhs_lo + c;
                const double x = b[c] * dinv;
                if (g == 0) Cs[mycol] = x; else Ks[mycol * ldx + g - 1] = x;
            }
        }
    }
}

// The same elimination in two parts, so that the 18 dependent pivot steps (search -> reciprocal -> multipliers) are done ONCE per Riccati
// step instead of once per wavefront.  lq_factor_rows: the matrix alone (one row per lane): pivot lane and multipliers of every step go to
// LDS -- the multiplier of (row i, step k) over the dead entry G[i][k], the pivot lanes / solved columns / reciprocal pivots to `fac`
// (32 doubles dinv | 32 ints solved column per row | NR ints pivot lane per step).  lq_apply_rows: a wave replays the steps on its slice of
// right-hand sides: per step one multiplier read, and per column two v_readlane + one FMA.  Same operations in the same order per column
// as lq_gj_rows: the results are bit-identical.
template <int NR>
__device__ __forceinline__ void lq_factor_rows(double *G_generic, int ldw, int nU, int lane, double *fac_generic, int *sing) {
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) int lds_int;
    lds_double *G = (lds_double *)G_generic, *dinv_s = (lds_double *)fac_generic;
    lds_int *col_s = (lds_int *)(dinv_s + 32), *piv_s = col_s + 32;
    const bool mine = lane < nU, ident = lane >= nU && lane < NR;
    double m[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) m[j] = mine ? (j < nU ? G[lane * ldw + j] : 0.0) : ((ident && j == lane) ? 1.0 : 0.0);
    int mycol = -1;
    double diag = 1.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NR; k++) if (k < nU) {      // (steps nU .. NR-1 are the identity block: they change nothing; a uniform branch per unrolled step)
        const bool free_row = (mine || ident) && mycol < 0;
        const float cand = free_row ? (float)fabs(m[k]) : 0.0f;
        unsigned int key = (__float_as_uint(cand) & ~0x3Fu) | (unsigned int)(63 - lane);
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true));
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x112, 0xf, 0xf, true));
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x114, 0xf, 0xf, true));
        key = max(key, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)key, 0x118, 0xf, 0xf, true));
        const unsigned int best = max((unsigned int)__builtin_amdgcn_readlane((int)key, 15), (unsigned int)__builtin_amdgcn_readlane((int)key, 31));
        const int src = 63 - (int)(best & 0x3Fu);
        if (!(__uint_as_float(best & ~0x3Fu) > 0.0f)) ok = false;
        auto bcast = [&](double v) -> double {
            const long long w = __double_as_longlong(v);
            const int lo = __builtin_amdgcn_readlane((int)(w & 0xFFFFFFFFLL), src), hi = __builtin_amdgcn_readlane((int)(w >> 32), src);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
        const bool is_piv = lane == src;
        const double piv = bcast(m[k]);
        double inv = __builtin_amdgcn_rcp(piv);
        { const double e = fma(-piv, inv, 1.0); inv = fma(inv, fma(e, e, e), inv); }
        const double l = (ok && !is_piv) ? m[k] * inv : 0.0;
#pragma unroll
        for (int j = k + 1; j < NR; j += 2) {
            const double p0 = bcast(m[j]), p1 = bcast(m[j + 1 < NR ? j + 1 : j]);
            m[j] = fma(-l, p0, m[j]);
            if (j + 1 < NR) m[j + 1] = fma(-l, p1, m[j + 1]);
        }
        if (is_piv) { mycol = k; diag = m[k]; }
        if (mine) G[lane * ldw + k] = l;
        if (lane == 0) piv_s[k] = src;
    }
    if (!ok && lane == 0) *sing = 1;
    if (lane < 32) { col_s[lane] = mycol; dinv_s[lane] = 1.0 / diag; }
}

// gamma = R_k + B'PB is symmetric and, with a positive definite R_k, positive definite: it can be eliminated in index order without a
// pivot search.  Same layout as lq_factor_rows (one row per lane, the columns in registers) with the pivot of step k taken from lane k:
// the broadcasts are v_readlane with a constant lane, and the per-step chain is reciprocal -> multiplier -> first FMA instead of
// candidate -> four DPP maxima -> two v_readlane -> v_readfirstlane -> reciprocal -> ...  (the pivoted factorisation is ~20 k cycles of
// the ~70 k of a Riccati step, executed by one wave while seven wait at the barrier).  Every pivot is guarded -- |pivot| must exceed
// 2^-20 of the largest entry of its row (the Newton model's R_k + HZ_uu can be indefinite far from the optimum) -- and nothing is
// written before all guards have passed: on failure the caller runs lq_factor_rows on the untouched matrix.  Writes the same tables as
// lq_factor_rows (pivot lane of step k = k, row i solves column i), so lq_apply_rows replays it unchanged.
template <int NR>
__device__ __forceinline__ bool lq_factor_rows_spd(double *G_generic, int ldw, int nU, int lane, double *fac_generic) {
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) int lds_int;
    lds_double *G = (lds_double *)G_generic, *dinv_s = (lds_double *)fac_generic;
    lds_int *col_s = (lds_int *)(dinv_s + 32), *piv_s = col_s + 32;
    const bool mine = lane < nU, ident = lane >= nU && lane < NR;
    double m[NR], amax = 0.0;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        m[j] = mine ? (j < nU ? G[lane * ldw + j] : 0.0) : ((ident && j == lane) ? 1.0 : 0.0);
        amax = fmax(amax, fabs(m[j]));
    }
    const double guard = 9.5367431640625e-07 * amax;   // 2^-20 of the row's largest entry
    double myinv = 1.0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NR; k++) if (k < nU) {
        auto bcast = [&](double v) -> double {
            const long long w = __double_as_longlong(v);
            const int lo = __builtin_amdgcn_readlane((int)(w & 0xFFFFFFFFLL), k), hi = __builtin_amdgcn_readlane((int)(w >> 32), k);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
        const bool is_piv = lane == k;
        bad = bad || (is_piv && !(fabs(m[k]) > guard));
        const double piv = bcast(m[k]);
        double inv = __builtin_amdgcn_rcp(piv);
        { const double e = fma(-piv, inv, 1.0); inv = fma(inv, fma(e, e, e), inv); }
        const double l = is_piv ? 0.0 : m[k] * inv;
#pragma unroll
        for (int j = k + 1; j < NR; j += 2) {
            const double p0 = bcast(m[j]), p1 = bcast(m[j + 1 < NR ? j + 1 : j]);
            m[j] = fma(-l, p0, m[j]);
            if (j + 1 < NR) m[j + 1] = fma(-l, p1, m[j + 1]);
        }
        myinv = is_piv ? inv : myinv;
        m[k] = is_piv ? m[k] : l;           // column k is dead below and above the pivot: keep the multiplier there
    }
    if (__any(bad ? 1 : 0)) return false;
    if (mine) {
#pragma unroll
        for (int k = 0; k < NR; k++) if (k < nU) G[lane * ldw + k] = lane == k ? 0.0 : m[k];
    }
    if (lane < nU) piv_s[lane] = lane;
    if (lane < 32) { col_s[lane] = lane < nU ? lane : -1; dinv_s[lane] = myinv; }
    return true;
}

// The same unpivoted factorisation with TWO matrix rows per lane (rows 2 i and 2 i + 1 in lane i < NR / 2 <= 16): the pivot row of step k then sits
// in lane k / 2 of the first 16-lane DPP row, and the elimination is `v_fmac_f64_dpp ... row_newbcast:k/2` -- one instruction per (row, column)
// instead of two v_readlane + one FMA, no SGPR hazards.  And it leaves gamma^-1 itself (in-place Gauss-Jordan inversion, over gamma in G) instead of
// the multipliers: the replay of the elimination on the 1 + nX right-hand sides -- 18 steps of two v_readlane + one FMA per column on 18 of 64
// lanes, the longest phase of the step (17 k cycles) -- becomes a product gamma^-1 [r | Kpart] on the matrix cores (lq_apply_inverse).  Same guards
// as lq_factor_rows_spd; nothing is written unless all of them passed (the caller then falls back to the pivoted factorisation + replay).
template <int L>
__device__ __forceinline__ double lq_bcast16(double v) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + L, 0xf, 0xf, true));
}
template <int L>
__device__ __forceinline__ void lq_fmac16(double &a, double b, double c) {      // a += (lane L's b) * c
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c), "n"(L));
}
template <int K, int NR>
__device__ __forceinline__ void lq_spd2_steps(double (&m0)[NR], double (&m1)[NR], int lane, int nU, double &inv0, double &inv1) {
    if constexpr (K < NR) {
        if (K < nU) {      // (uniform)
            constexpr int L = K >> 1;
            constexpr bool ODD = (K & 1) != 0;
            const double piv = lq_bcast16<L>(ODD ? m1[K] : m0[K]);
            double inv = __builtin_amdgcn_rcp(piv);
            { const double e = fma(-piv, inv, 1.0); inv = fma(inv, fma(e, e, e), inv); }
            const bool pl = lane == L;
            double n0 = -(m0[K] * inv), n1 = -(m1[K] * inv);      // minus the multipliers of the lane's two rows
            if (ODD) { n1 = pl ? 0.0 : n1; inv1 = pl ? inv : inv1; } else { n0 = pl ? 0.0 : n0; inv0 = pl ? inv : inv0; }
            // IN-PLACE Gauss-Jordan inversion: column K of the matrix is dead after this step and becomes column K of the inverse's
            // numerator (pivot row: 1, other rows: minus their multiplier); the columns j < K already are such columns and are updated
            // like the live matrix columns j > K
#pragma unroll
            for (int j = 0; j < NR; j++) {
                if (j == K) continue;
                if (ODD) { lq_fmac16<L>(m0[j], m1[j], n0); lq_fmac16<L>(m1[j], m1[j], n1); }
                else { lq_fmac16<L>(m1[j], m0[j], n1); lq_fmac16<L>(m0[j], m0[j], n0); }
            }
            m0[K] = (!ODD && pl) ? 1.0 : n0; m1[K] = (ODD && pl) ? 1.0 : n1;
        }
        lq_spd2_steps<K + 1, NR>(m0, m1, lane, nU, inv0, inv1);
    }
}
template <int NR>
__device__ __forceinline__ bool lq_factor_rows_spd2(double *G_generic, int ldw, int nU, int lane, double *fac_generic) {
    static_assert(NR % 2 == 0 && NR <= 32, "two rows per lane on one 16-lane DPP row");
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) int lds_int;
    lds_double *G = (lds_double *)G_generic, *dinv_s = (lds_double *)fac_generic;
    lds_int *col_s = (lds_int *)(dinv_s + 32), *piv_s = col_s + 32;
    const int r0 = 2 * (lane & 15), r1 = r0 + 1;
    const bool have = lane < NR / 2, in0 = have && r0 < nU, in1 = have && r1 < nU;
    double m0[NR], m1[NR], a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const double x0 = G[(in0 ? r0 : 0) * ldw + (j < nU ? j : 0)], x1 = G[(in1 ? r1 : 0) * ldw + (j < nU ? j : 0)];
        m0[j] = (in0 && j < nU) ? x0 : ((have && !in0 && j == r0) ? 1.0 : 0.0);
        m1[j] = (in1 && j < nU) ? x1 : ((have && !in1 && j == r1) ? 1.0 : 0.0);
        a0 = fmax(a0, fabs(m0[j])); a1 = fmax(a1, fabs(m1[j]));
    }
    double inv0 = 0.0, inv1 = 0.0;
    lq_spd2_steps<0, NR>(m0, m1, lane, nU, inv0, inv1);
    // the guards: |pivot| > 2^-20 of the row's largest entry, from the reciprocal kept by each row (a zero pivot gives inf / NaN: fails)
    const bool bad = (in0 && !(fabs(inv0) * (9.5367431640625e-07 * a0) < 1.0)) || (in1 && !(fabs(inv1) * (9.5367431640625e-07 * a1) < 1.0));
    if (__any(bad ? 1 : 0)) return false;
#pragma unroll
    for (int k = 0; k < NR; k++) if (k < nU) {      // gamma^-1 [row][k] = numerator / pivot of the row
        if (in0) G[r0 * ldw + k] = m0[k] * inv0;
        if (in1) G[r1 * ldw + k] = m1[k] * inv1;
    }
    (void)piv_s; (void)col_s; (void)dinv_s;
    return true;
}

template <int NR, int SL>
__device__ __forceinline__ void lq_apply_rows(const double *G_generic, int ldw, int nU, int rhs_lo, int rhs_n, double *Ks_generic, int ldx,
                                              double *Cs_generic, int lane, const double *fac_generic) {
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) int lds_int;
    const lds_double *G = (const lds_double *)G_generic, *dinv_s = (const lds_double *)fac_generic;
    const lds_int *col_s = (const lds_int *)(dinv_s + 32), *piv_s = col_s + 32;
    lds_double *Ks = (lds_double *)Ks_generic, *Cs = (lds_double *)Cs_generic;
    const bool mine = lane < nU;
    double b[SL];
#pragma unroll
    for (int c = 0; c < SL; c++) b[c] = (mine && c < rhs_n) ? G[lane * ldw + nU + rhs_lo + c] : 0.0;
    for (int k = 0; k < nU; k++) {
        const int src = __builtin_amdgcn_readfirstlane(piv_s[k]);
        const double l = mine ? G[lane * ldw + k] : 0.0;
        auto bcast = [&](double v) -> double {
            const long long w = __double_as_longlong(v);
            const int lo = __builtin_amdgcn_readlane((int)(w & 0xFFFFFFFFLL), src), hi = __builtin_amdgcn_readlane((int)(w >> 32), src);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
#pragma unroll
        for (int c = 0; c < SL; c += 2) {
            const double p0 = bcast(b[c]), p1 = bcast(b[c + 1 < SL ? c + 1 : c]);
            b[c] = fma(-l, p0, b[c]);
            if (c + 1 < SL) b[c + 1] = fma(-l, p1, b[c + 1]);
        }
    }
    const int mycol = lane < 32 ? col_s[lane] : -1;
    if (mine && mycol >= 0 && mycol < nU) {
        const double dinv = dinv_s[lane];
#pragma unroll
        for (int c = 0; c < SL; c++) {
            if (c < rhs_n) {
                const int g = rhs_lo + c;
                const double x = b[c] * dinv;
                if (g == 0) Cs[mycol] = x; else Ks[mycol * ldx + g - 1] = x;
            }
        }
    }
}

struct LqLayout {   // LDS layout of k_tv_lq_mfma in doubles
    int ldx, nUp, ldw, Pm, Am, Bm, Kp, Ks, G, bv, bn, wv, rv, scr, fac, total;
    __host__ __device__ LqLayout(int nX, int nU) {
        ldx = round_up(nX, 16); nUp = round_up(nU, 4); ldw = nU + 1 + ldx;
        int o = 0;
        Pm = o; o += ldx * ldx; Am = o; o += ldx * ldx; Bm = o; o += ldx * nU;
        Kp = o; o += nUp * ldx;
        Ks = o; o += (nUp * ldx > ldx * nUp ? nUp * ldx : ldx * nUp);   // K_k [nUp][ldx]; before the solve: P B [ldx][nUp]
        G = o; o += nU * ldw; bv = o; o += ldx; bn = o; o += ldx; wv = o; o += nUp; rv = o; o += nUp;
        scr = o; o += nU > 32 ? (LQM_T / 64) * 3 * round_up(nU, 4) : 0;     // scratch of lq_gj_wave (more than 32 inputs) only
        fac = o; o += nU <= 32 ? 32 + 16 + 16 : 0;                          // lq_factor_rows: reciprocal pivots, solved column per row, pivot lane per step
        total = o;
    }
};

template <int NT, int NR>   // 16 x 16 tiles per dimension: nX <= 16 NT; rows of the input solve in registers: nU <= NR
__global__ __launch_bounds__(LQM_T) void k_tv_lq_mfma(const tg_lq_problem a) {
    extern __shared__ double lds[];
    __shared__ int s_sing;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int s = a.select_dev ? a.select_dev[blockIdx.x] : blockIdx.x;
    const int nX = a.nX, nU = a.nU, N = a.horizon;
    const LqLayout L(16 * NT, nU);
    constexpr int ldx = 16 * NT, NTILES = NT * NT, NW = LQM_T / 64, TMAX = (NTILES + NW - 1) / NW;
    const int nUp = L.nUp, ldw = L.ldw, NUT = (nU + 15) >> 4;
    double *Pm = lds + L.Pm, *Am = lds + L.Am, *Bm = lds + L.Bm, *Kp = lds + L.Kp, *Ks = lds + L.Ks, *PB = Ks, *G = lds + L.G;
    double *bv = lds + L.bv, *bn = lds + L.bn, *wv = lds + L.wv, *rv = lds + L.rv, *scr = lds + L.scr + wave * 3 * NR, *fac = lds + L.fac;
    for (int i = tid; i < L.total; i += LQM_T) lds[i] = 0.0;
    if (tid == 0) s_sing = 0;
    __syncthreads();
    const size_t sN = (size_t)s * N;
    const bool affine = a.q_dev != nullptr;
    const int nxh = a.hz_nx, hzR = a.hz_R;
    const int kb = a.k_begin, ke = a.k_end > 0 ? a.k_end : N;      // the steps of this launch (tg_lq_problem::k_*)
    if (a.Pt_dev) {
        const double *Pt = a.Pt_dev + (size_t)s * nX * nX;
        for (int e = tid; e < nX * nX; e += LQM_T) Pm[(e / nX) * ldx + e % nX] = Pt[e];
        if (affine && a.bt_dev) for (int i = tid; i < nX; i += LQM_T) bv[i] = a.bt_dev[(size_t)s * nX + i];
    } else {
        const double *Qf = a.Qf_dev + (size_t)s * a.Qf_seed_stride;
        for (int e = tid; e < nX * nX; e += LQM_T) Pm[(e / nX) * ldx + e % nX] = Qf[e];
        if (affine) for (int i = tid; i < nX; i += LQM_T) bv[i] = a.q_dev[(sN + s + N) * nX + i];
    }
    // next A_k, B_k: global -> registers while the step computes -> LDS at its end (8 row groups x 32 columns per pass)
    constexpr int RG = LQM_T / 32, RI = (ldx + RG - 1) / RG, CI = (ldx + 31) / 32, PBN = (ldx * 32 + LQM_T - 1) / LQM_T;
    const int pr0 = tid >> 5, pc0 = tid & 31;   // LQM_T / 32 row groups x 32 columns per pass
    double preA[RI * CI], preB[PBN];
    auto prefetch = [&](int k) {
        const double *Ak = a.A_dev + (sN + k) * (size_t)nX * nX, *Bk = a.B_dev + (sN + k) * (size_t)nX * nU;
#pragma unroll
        for (int ri = 0; ri < RI; ri++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) {
                const int r = pr0 + RG * ri, c = pc0 + 32 * ci;
                if (r < nX && c < nX) preA[ri * CI + ci] = Ak[r * nX + c];
            }
#pragma unroll
        for (int i = 0; i < PBN; i++) { const int e = tid + i * LQM_T; if (e < nX * nU) preB[i] = Bk[e]; }
    };
    auto commit = [&]() {
#pragma unroll
        for (int ri = 0; ri < RI; ri++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) {
                const int r = pr0 + RG * ri, c = pc0 + 32 * ci;
                if (r < nX && c < nX) Am[r * ldx + c] = preA[ri * CI + ci];
            }
#pragma unroll
        for (int i = 0; i < PBN; i++) { const int e = tid + i * LQM_T; if (e < nX * nU) Bm[e] = preB[i]; }
    };
    prefetch(ke - 1);
    commit();
    __syncthreads();
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    v4d acc[TMAX];
    const int rhs_total = 1 + nX, slice = (rhs_total + NW - 1) / NW;
#if defined(TG_PROFILE)
    long long lq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lq_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int k = ke - 1; k >= kb; k--) {
        if (k > kb) prefetch(k - 1);
        const double *hz = a.hz_dev ? a.hz_dev + (sN + k) * (size_t)hzR * hzR : nullptr;
        if (a.b_next_dev && affine) for (int i = tid; i < nX; i += LQM_T) a.b_next_dev[(sN + k) * nX + i] = bv[i];    // b_{k+1}
        // ---- phase 1: P A tiles (registers), P B -> LDS, B'b ---------------------------------------------------
        // all tiles of the wave at once: TMAX independent accumulation chains (a dependent f64 MFMA issues every ~200 cycles,
        // independent ones every ~64: tools/micro/mfma_f64_rate.hip)
        {
            int ti_[TMAX], tj_[TMAX];
            bool ok_[TMAX];
#pragma unroll
            for (int i = 0; i < TMAX; i++) {
                const int t = wave + NW * i;
                ok_[i] = t < NTILES; ti_[i] = ok_[i] ? t / NT : 0; tj_[i] = ok_[i] ? t % NT : 0; acc[i] = zero4;
            }
#pragma unroll 2
            for (int k0 = 0; k0 < ldx; k0 += 4) {
                double av[TMAX], bw[TMAX];
#pragma unroll
                for (int i = 0; i < TMAX; i++) {
                    av[i] = Pm[(k0 + lk) * ldx + 16 * ti_[i] + lr];      // P[i][k] = P[k][i]
                    bw[i] = Am[(k0 + lk) * ldx + 16 * tj_[i] + lr];
                }
#pragma unroll
                for (int i = 0; i < TMAX; i++) if (ok_[i]) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bw[i], acc[i], 0, 0, 0);
            }
        }
        for (int t = wave; t < NT * NUT; t += NW) {
            const int ti = t / NUT, tu = t % NUT, u = 16 * tu + lr;
            v4d c = zero4;
#pragma unroll 4
            for (int k0 = 0; k0 < ldx; k0 += 4) {
                const double av = Pm[(k0 + lk) * ldx + 16 * ti + lr];
                const double bw = u < nU ? Bm[(k0 + lk) * nU + u] : 0.0;
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bw, c, 0, 0, 0);
            }
            if (u < nU) {
#pragma unroll
                for (int r = 0; r < 4; r++) PB[(16 * ti + lk + 4 * r) * nUp + u] = c[r];
            }
        }
        // B'b rides along with gamma = R + B'(P B) when P B has a spare padding column (nU not a multiple of 4): b goes into column nU
        // of the staged P B, and column nU of the gamma tiles comes out as B'b (phase 2b) -- no serial dot products
        const bool bb_in_tile = affine && nUp > nU;
        if (affine && tid < nU) {
            if (!bb_in_tile) {
                double w = 0.0;
                for (int i = 0; i < nX; i++) w += Bm[i * nU + tid] * bv[i];
                wv[tid] = w;
            }
            rv[tid] = a.r_dev[(sN + k) * nU + tid];
        }
        if (bb_in_tile) for (int i = tid; i < ldx; i += LQM_T) PB[i * nUp + nU] = i < nX ? bv[i] : 0.0;
        __syncthreads();
        LQ_STAMP(0);
        // ---- phase 2a: P A -> LDS (over P) -------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TMAX; i++) {
            const int t = wave + NW * i;
            if (t < NTILES) {
                const int ti = t / NT, tj = t % NT;
#pragma unroll
                for (int r = 0; r < 4; r++) Pm[(16 * ti + lk + 4 * r) * ldx + 16 * tj + lr] = acc[i][r];
            }
        }
        __syncthreads();
        // The upper-triangle tiles of the new P that this wave owns (phases 3 / 4).  (Loading their weights Q_k here, a phase early, takes the
        // load latency off the tile chains -- 16.5 k -> 7 k cycles -- but the factorising wave needs ~20 k anyway, and the accumulators live
        // through phase 2b cost phase 1 its registers: 38.4 instead of 35.6 us per step.)
        constexpr bool SPLIT = NR <= 32;
        constexpr int NTRI = NT * (NT + 1) / 2, TW = SPLIT ? NW - 1 : NW, TSYM = (NTRI + TW - 1) / TW;
        static_assert(TSYM <= TMAX, "accumulator array too small for the upper-triangle tiles");
        const double *Qk = a.Q_dev + (size_t)s * a.Q_seed_stride + (size_t)k * a.Q_step_stride;
        int ti_[TMAX], tj_[TMAX];
        bool ok_[TMAX];
#pragma unroll
        for (int i = 0; i < TMAX; i++) {
            const int t = wave + TW * i;
            ok_[i] = i < TSYM && t < NTRI && wave < TW;
            // t-th tile of the upper triangle, row by row: row r holds NT - r tiles
            int tr = 0, tt = ok_[i] ? t : 0;
#pragma unroll
            for (int r = 0; r < NT; r++) if (tt >= NT - r && tr == r) { tt -= NT - r; tr = r + 1; }
            ti_[i] = tr; tj_[i] = tr + tt;
        }
        // ---- phase 2b: gamma = R + B'(P B), Kpart = B'(P A) + S' -> [gamma | r + B'b | Kpart] ------------------------
        for (int t = wave; t < NUT * (NUT + NT); t += NW) {
            const int tu = t / (NUT + NT), tc = t % (NUT + NT), u = 16 * tu + lr;
            const bool is_gamma = tc < NUT;
            const int tj = is_gamma ? tc : tc - NUT, col = 16 * tj + lr;
            v4d c = zero4;
#pragma unroll 4
            for (int k0 = 0; k0 < ldx; k0 += 4) {
                const double av = u < nU ? Bm[(k0 + lk) * nU + u] : 0.0;       // B'[u][k]
                const double bw = is_gamma ? (col < nU + (bb_in_tile ? 1 : 0) ? PB[(k0 + lk) * nUp + col] : 0.0) : Pm[(k0 + lk) * ldx + col];
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bw, c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int uo = 16 * tu + lk + 4 * r;      // output row (input index u), column `col`
                if (uo < nU) {
                    if (is_gamma) {
                        if (col < nU) {
                            double g = c[r] + a.R_dev[(size_t)s * a.R_seed_stride + (size_t)k * a.R_step_stride + uo * nU + col];
                            if (hz) g += hz[(size_t)(nxh + uo) * hzR + nxh + col];
                            G[uo * ldw + col] = g;
                        } else if (bb_in_tile && col == nU) {       // r_k + B'b
                            const double rw = c[r] + rv[uo];
                            rv[uo] = rw; G[uo * ldw + nU] = rw;
                        }
                    } else {
                        double v = c[r];
                        if (hz && col < nxh) v += hz[(size_t)col * hzR + nxh + uo];
                        Kp[uo * ldx + col] = v; G[uo * ldw + nU + 1 + col] = v;
                    }
                }
            }
        }
        if (tid < nU && !bb_in_tile) { const double rw = affine ? wv[tid] + rv[tid] : 0.0; rv[tid] = rw; G[tid * ldw + nU] = rw; }   // r_k + B'b
        __syncthreads();
        LQ_STAMP(1);
        // ---- phase 3 / 4 ----------------------------------------------------------------------------------------------------
        // The new P tiles = Q + A'(P A) - Kpart' K are symmetric (Q_k, A'(P A) with symmetric P, Kpart' gamma^-1 Kpart): only the
        // NT (NT + 1) / 2 tiles on and above the diagonal are computed, phase 6 mirrors them.  Up to 32 inputs: the LAST wave factorises
        // gamma (lq_factor_rows: the 18 dependent pivot steps once per step, not once per wave) WHILE the other waves accumulate
        // Q + A'(P A) -- that part does not need the gains --; then every wave replays the steps on its slice of the right-hand
        // sides (lq_apply_rows) and the tile owners add -Kpart' K.  More inputs: every wave solves its slice itself (lq_gj_wave).
        auto tiles_q_apa = [&]() {      // acc = Q_k (+ curvature) + A'(P A)
#pragma unroll
            for (int i = 0; i < TMAX; i++) {
                v4d c = zero4;
                const int col = 16 * tj_[i] + lr;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * ti_[i] + lk + 4 * r;
                    if (ok_[i] && row < nX && col < nX) {
                        c[r] = Qk[(size_t)row * nX + col];
                        if (hz && row < nxh && col < nxh) c[r] += hz[(size_t)row * hzR + col];
                    }
                }
                acc[i] = c;
            }
#pragma unroll 2
            for (int k0 = 0; k0 < ldx; k0 += 4) {
                double av[TMAX], bw[TMAX];
#pragma unroll
                for (int i = 0; i < TMAX; i++) {
                    av[i] = Am[(k0 + lk) * ldx + 16 * ti_[i] + lr];      // A'[i][k] = A[k][i]
                    bw[i] = Pm[(k0 + lk) * ldx + 16 * tj_[i] + lr];      // (P A)[k][j]
                }
#pragma unroll
                for (int i = 0; i < TMAX; i++) if (ok_[i]) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bw[i], acc[i], 0, 0, 0);
            }
        };
        auto tiles_minus_kpk = [&]() {  // acc -= Kpart' K
            for (int u0 = 0; u0 < nUp; u0 += 4) {
                double av[TMAX], bw[TMAX];
#pragma unroll
                for (int i = 0; i < TMAX; i++) {
                    av[i] = -Kp[(u0 + lk) * ldx + 16 * ti_[i] + lr];     // -Kpart'[i][u]
                    bw[i] = Ks[(u0 + lk) * ldx + 16 * tj_[i] + lr];
                }
#pragma unroll
                for (int i = 0; i < TMAX; i++) if (ok_[i]) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bw[i], acc[i], 0, 0, 0);
            }
        };
        const int lo = wave * slice, nrhs = lo < rhs_total ? (rhs_total - lo < slice ? rhs_total - lo : slice) : 0;
        if constexpr (SPLIT) {
            if (wave == NW - 1) lq_factor_rows<NR>(G, ldw, nU, lane, fac, &s_sing);
            else tiles_q_apa();
            LQ_STAMP(6);
            for (int o = tid; o < (nUp - nU) * ldx; o += LQM_T) Ks[nU * ldx + o] = 0.0;   // padding rows (the buffer held P B)
            __syncthreads();
            lq_apply_rows<NR, (16 * NT + 1 + NW - 1) / NW>(G, ldw, nU, lo, nrhs, Ks, ldx, wv, lane, fac);
            __syncthreads();
            LQ_STAMP(2);
        } else {
            lq_gj_wave<NR>(G, ldw, nU, lo, nrhs, Ks, ldx, wv, scr, lane, &s_sing);
            LQ_STAMP(6);
            for (int o = tid; o < (nUp - nU) * ldx; o += LQM_T) Ks[nU * ldx + o] = 0.0;   // padding rows (the buffer held P B)
            __syncthreads();
            LQ_STAMP(2);
            tiles_q_apa();
        }
        double *Cs = wv;
        {   // outputs K_k, C_k
            double *Ko = a.K_dev + (sN + k) * (size_t)nU * nX;
            for (int o = tid; o < nU * nX; o += LQM_T) Ko[o] = Ks[(o / nX) * ldx + o % nX];
            if (a.C_dev && tid < nU) a.C_dev[(sN + k) * nU + tid] = Cs[tid];
        }
        tiles_minus_kpk();
        if (affine) for (int i = tid; i < nX; i += LQM_T) {     // new b = q_k + A'b - K'(r_k + B'b): four independent chains (the LDS reads in flight together)
            double v0 = a.q_dev[(sN + s + k) * nX + i], v1 = 0.0, v2 = 0.0, v3 = 0.0;
            int m = 0;
            for (; m + 4 <= nX; m += 4) {
                v0 = fma(Am[m * ldx + i], bv[m], v0); v1 = fma(Am[(m + 1) * ldx + i], bv[m + 1], v1);
                v2 = fma(Am[(m + 2) * ldx + i], bv[m + 2], v2); v3 = fma(Am[(m + 3) * ldx + i], bv[m + 3], v3);
            }
            for (; m < nX; m++) v0 = fma(Am[m * ldx + i], bv[m], v0);
            int u = 0;
            for (; u + 2 <= nU; u += 2) { v1 = fma(-Ks[u * ldx + i], rv[u], v1); v2 = fma(-Ks[(u + 1) * ldx + i], rv[u + 1], v2); }
            for (; u < nU; u++) v3 = fma(-Ks[u * ldx + i], rv[u], v3);
            bn[i] = (v0 + v1) + (v2 + v3);
        }
        __syncthreads();
        LQ_STAMP(3);
        // ---- phase 5: P <- new tiles, b <- new b, next A, B into LDS ---------------------------------------------
#pragma unroll
        for (int i = 0; i < TMAX; i++) {
            if (ok_[i]) {
#pragma unroll
                for (int r = 0; r < 4; r++) Pm[(16 * ti_[i] + lk + 4 * r) * ldx + 16 * tj_[i] + lr] = acc[i][r];
            }
        }
        if (affine) for (int i = tid; i < nX; i += LQM_T) bv[i] = bn[i];
        if (k > kb) commit();
        __syncthreads();
        LQ_STAMP(4);
        // ---- phase 6: the lower triangle is the mirror of the upper one -------------------------------------------------
        // along diagonals: (i, i + d) and (i + d, i) are both strided by ldx + 1 doubles over the lanes (no bank conflicts; a
        // row-wise pass reads the transposed element with stride ldx = 16-way conflicts).  Inside the diagonal tiles both
        // triangles were computed: there the two are averaged as before.
        for (int d = 1 + wave; d < nX; d += NW)
            for (int i = lane; i < nX - d; i += 64) {
                const bool diag_tile = (i >> 4) == ((i + d) >> 4);
                const double up = Pm[i * ldx + i + d];
                const double v = diag_tile ? 0.5 * (up + Pm[(i + d) * ldx + i]) : up;
                Pm[i * ldx + i + d] = v; Pm[(i + d) * ldx + i] = v;
            }
        __syncthreads();
        LQ_STAMP(5);
    }
#if defined(TG_PROFILE)
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 8; i++) g_lq_prof[i] = lq_acc[i];
#endif
    if (a.P0_dev) for (int e = tid; e < nX * nX; e += LQM_T) a.P0_dev[(size_t)s * nX * nX + e] = Pm[(e / nX) * ldx + e % nX];
    if (a.b0_dev && affine) for (int i = tid; i < nX; i += LQM_T) a.b0_dev[(size_t)s * nX + i] = bv[i];
    if (a.status_dev && tid == 0) a.status_dev[s] = (s_sing || (a.Pt_dev && a.status_dev[s] != TG_OK)) ? TG_SINGULAR : TG_OK;      // (a later chunk of a chunked sweep keeps an earlier chunk's verdict)
}

// ------------------------------------------------------------------------------------------------------
// The same sweep for problems with the block structure of DSystem.fdx / fdu (tg_lq_problem::ds_*; reference
// trep/discopt/dsystem.py:284-317): states [Qd | Qk | p | v], inputs [u | rho].  A_k has dense rows for Qd and p only
// (over the Q and p columns); its Qk rows and v columns are zero and a v row holds one entry, A[v_m][Qk_m] (= -1/dt).
// B_k has dense Qd / p rows; a Qk row and a v row hold one entry each, in column rho_m.  So
//   * only the dense rows are staged: AD [KC][ldc], KC = 2 round_up(nd, 4) "compact" rows -- the Qd block padded to a multiple of four
//     with the first Qk rows (zero rows), then the p block padded with the first v rows -- ldc = 16 NTC columns covering Q and p.
//     Every k-loop of the dense kernel (20 steps at nX = 80) becomes KC / 4 steps (12), every A-shaped result has NTC (4) instead of
//     NT (5) column tiles: 622 instead of 1355 matrix-core instructions per Riccati step at the puppet's sizes;
//   * the single-entry rows that are not among the compact rows enter as one or two extra terms per output element;
//   * the v columns of P A, Kpart and K are zero and are never computed; the v rows / columns of the new P are those of Q_k;
//   * AD is double buffered and filled by global_load_lds_dwordx4 -- global memory straight into LDS, no staging registers (the dense
//     kernel holds A_{k-1}, B_{k-1} in 40 VGPRs through the whole step and is the worse for it: 67 spilled VGPRs) and no commit pass;
//     B_k is dead after the gamma / Kpart tiles and is refilled in place the same way; inside a step the barriers only order LDS
//     traffic (no vmcnt wait: the loads of step k-1 have the whole of step k to land), the step's last barrier waits for them;
//   * gamma = R_k + B'PB is factorised without a pivot search (lq_factor_rows_spd), the pivoted factorisation being the fallback;
//   * Kpart is kept once, as the right-hand-side block of G.
// Same arithmetic otherwise (the products sum the same non-zero terms in compact-row order).
// ------------------------------------------------------------------------------------------------------
struct LqDsLayout {
    int ldx, nUp, ldw, nd4, KC, ldc, lda, Pm, AD, adsz, Bm, avs, Ks, G, bv, bn, wv, rv, rn, fac, total;
    __host__ __device__ LqDsLayout(int ldx_, int nU, int nd, int nq) {
        ldx = ldx_; nUp = round_up(nU, 4); ldw = nU + 1 + ldx; nd4 = round_up(nd, 4); KC = 2 * nd4; ldc = 16 * ((nq + nd + 15) >> 4);
        // row stride of AD: a ds_read_b64 serves lanes 0-31 (two rows of 16 doubles) in one cycle if the rows sit in opposite halves of the
        // 64 banks, i.e. stride = 16 mod 32 doubles (ldx = 80 is; ldc = 64 is not: the padding columns are never loaded)
        lda = (ldc % 32 == 16) ? ldc : ldc + 16;
        int o = 0;
        Pm = o; o += ldx * ldx;
        adsz = round_up(KC * lda, 128); AD = o; o += 2 * adsz;            // two buffers, each a whole number of 1 KB load instructions
        Bm = o; o += round_up(ldx * nU, 128);
        avs = o; o += 64;                                                   // A[v_m][Qk_m], two buffers of 32 (nk <= 31)
        Ks = o; o += nUp * ldx;                                             // K_k [nUp][ldx]; before the solve: P B [ldx][nUp]
        G = o; o += nUp * ldw;
        bv = o; o += ldx; bn = o; o += ldx; wv = o; o += nUp; rv = o; o += nUp; rn = o; o += nUp;
        fac = o; o += 32 + 16 + 16 + 2;      // (+ the form flag of lq_factor_call)
        total = o;
    }
};

// Barrier inside a step: orders the waves' plain LDS traffic only.  A workgroup-scope release fence would also wait for the
// global_load_lds fills in flight (they write LDS, so the compiler counts them in: s_waitcnt vmcnt(0) -- ~8 k cycles at the first
// barrier after their issue); those land in the OTHER buffer and are awaited by the step's last barrier (__syncthreads).
#define LQ_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// (the pivoted fallback as a call of its own: its code is only fetched when a guard failed)
template <int NR>
__device__ __noinline__ void lq_factor_pivoted_call(double *G, int ldw, int nU, int lane, double *fac, int *sing) {
    ldw = __builtin_amdgcn_readfirstlane(ldw); nU = __builtin_amdgcn_readfirstlane(nU);
    lq_factor_rows<NR>(G, ldw, nU, lane, fac, sing);
}

// gamma's factorisation as a CALL: one wave of eight runs it, and inlined its 2 x 20 matrix registers (unpivoted attempt + pivoted fallback)
// are part of the sweep kernel's register allocation (158 spilled VGPRs)
template <int NR>
__device__ __noinline__ void lq_factor_call(double *G, int ldw, int nU, int lane, double *fac, int *sing) {
    // (arguments of a call travel in vector registers: say that the sizes are wave-uniform, or every `k < nU` becomes a divergent branch)
    ldw = __builtin_amdgcn_readfirstlane(ldw); nU = __builtin_amdgcn_readfirstlane(nU);
    // fac[64] (as an int): 1 = G's gamma block holds gamma^-1 (lq_apply_inverse), 0 = the multipliers of the pivoted factorisation (lq_apply_rows)
    const bool inverse = lq_factor_rows_spd2<NR>(G, ldw, nU, lane, fac);
    if (!inverse) lq_factor_pivoted_call<NR>(G, ldw, nU, lane, fac, sing);
    if (lane == 0) *reinterpret_cast<int *>(fac + 64) = inverse ? 1 : 0;
}

// Launch arguments re-read per phase: the struct sits at the head of the kernel-argument segment; reading it through a laundered
// constant-address-space pointer at the head of every phase keeps its ~40 scalars -- and everything derived from them -- from being
// hoisted out of the step loop and held (then spilled: 416 SGPR spills, 90 VGPRs) for the whole sweep.  K$-resident scalar loads.
typedef const __attribute__((address_space(4))) tg_lq_problem KLq;
__device__ __forceinline__ KLq &lq_fresh_args() {
    KLq *p = (KLq *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
}
// Everything a phase of k_tv_lq_ds derives from the arguments (sizes, LDS offsets, structure helpers), declared at the head of each phase
#define LQ_DS_PHASE                                                                                                                   \
    KLq &a = lq_fresh_args();                                                                                                         \
    const int nX = a.nX, nU = a.nU, N = a.horizon;                                                                                    \
    const int nd = a.ds_nd, nq = a.ds_nd + a.ds_nk, nu = a.ds_nu, nv0 = nq + nd;                                                      \
    const LqDsLayout L(ldx, nU, nd, nq);                                                                                              \
    const int nUp = L.nUp, ldw = L.ldw, NUT = (nU + 15) >> 4, nd4 = L.nd4, KC = L.KC, ldc = L.ldc, lda = L.lda, NTC = ldc >> 4, mx = nd4 - nd; \
    double *Pm = lds + L.Pm, *Bm = lds + L.Bm, *Ks = lds + L.Ks, *PB = Ks, *G = lds + L.G;                                            \
    double *bv = lds + L.bv, *bn = lds + L.bn, *wv = lds + L.wv, *rv = lds + L.rv, *fac = lds + L.fac;                                \
    const size_t sN = (size_t)s * N;                                                                                                  \
    const bool affine = a.q_dev != nullptr;                                                                                           \
    const int nxh = a.hz_nx, hzR = a.hz_R;                                                                                            \
    const double *AD = lds + L.AD + cur * L.adsz, *avs = lds + L.avs + cur * 32;                                                      \
    const double *hz = a.hz_dev ? a.hz_dev + (sN + k) * (size_t)hzR * hzR : nullptr;                                                  \
    const bool bb_in_tile = affine && nUp > nU;                                                                                       \
    /* global row / P column of compact row cr */                                                                                     \
    auto grow = [&](int cr) { return cr < nd4 ? cr : nq + (cr - nd4); };                                                              \
    /* single-entry rows outside the compact rows: v row of Qk column j of A; Qk / v rows of rho column u of B (or -1) */             \
    auto a_extra = [&](int j) { return (j >= nd + mx && j < nq) ? nv0 + (j - nd) : -1; };                                             \
    auto b_extra_q = [&](int u) { return (u >= nu + mx && u < nU) ? nd + (u - nu) : -1; };                                            \
    auto b_extra_v = [&](int u) { return (u >= nu + mx && u < nU) ? nv0 + (u - nu) : -1; };                                           \
    /* global memory -> LDS, 16 bytes per lane, 1 KB per wave instruction; the instructions of a fill are dealt to the waves */       \
    auto fill_AD = [&](int kk, double *dst) {                                                                                         \
        const double *Ak = a.A_dev + (sN + kk) * (size_t)nX * nX;                                                                     \
        const int npairs = (KC * lda) >> 1;                                                                                           \
        for (int i = wave; i * 64 < npairs; i += NW) {                                                                                \
            const int e = 2 * (i * 64 + lane), row = e / lda, col = e - row * lda, gr = grow(row);                                    \
            if (e < 2 * npairs && gr < nX && col < ldc && col < nX)                                                                   \
                __builtin_amdgcn_global_load_lds((gl_void *)(Ak + (size_t)gr * nX + col), (lds_void *)(dst + i * 128), 16, 0, 0);     \
        }                                                                                                                             \
    };                                                                                                                                \
    auto fill_B = [&](int kk) {                                                                                                       \
        const double *Bk = a.B_dev + (sN + kk) * (size_t)nX * nU;                                                                     \
        const int npairs = (nX * nU) >> 1;                                                                                            \
        for (int i = wave; i * 64 < npairs; i += NW) {                                                                                \
            const int e = 2 * (i * 64 + lane);                                                                                        \
            if (e < 2 * npairs) __builtin_amdgcn_global_load_lds((gl_void *)(Bk + e), (lds_void *)(Bm + i * 128), 16, 0, 0);          \
        }                                                                                                                             \
    };                                                                                                                                \
    auto load_av = [&](int kk) { return tid < nq - nd ? a.A_dev[((sN + kk) * (size_t)nX + nv0 + tid) * nX + nd + tid] : 0.0; };       \
    (void)N; (void)nu; (void)lda; (void)nUp; (void)ldw; (void)NUT; (void)KC; (void)NTC; (void)Pm; (void)PB; (void)G; (void)bv; (void)bn; (void)wv; (void)rv; (void)fac; \
    (void)nxh; (void)AD; (void)avs; (void)hz; (void)bb_in_tile

template <int NT, int NR>
__global__ __launch_bounds__(LQM_T) void k_tv_lq_ds(const tg_lq_problem a0) {
    extern __shared__ double lds[];
    __shared__ int s_sing;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gl_void;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    constexpr int ldx = 16 * NT, NW = LQM_T / 64;
    const int s = __builtin_amdgcn_readfirstlane(a0.select_dev ? a0.select_dev[blockIdx.x] : (int)blockIdx.x);
    const int kb = a0.k_begin, ke = a0.k_end > 0 ? a0.k_end : a0.horizon;      // the steps of this launch (tg_lq_problem::k_*)
    int cur = 0, k = ke - 1;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    constexpr int T1 = 4;                              // chains per wave in phase 1 (P A and P B tiles in one list: at most 32 tiles, see tg_tv_lq)
    // phase 3: wave 0 factorises gamma; the wave that shares its SIMD (a workgroup's waves are dealt to the four SIMDs in turn: wave 4) stays out of
    // the tile work -- every dependent fp64 operation of the factorisation otherwise queues behind a 64-cycle matrix-core instruction of its neighbour
    constexpr int NTRI = NT * (NT + 1) / 2, TW = NW - 2, TSYM = (NTRI + TW - 1) / TW;
    constexpr int TMAX = T1 > TSYM ? T1 : TSYM;
    v4d acc[TMAX];
    {
        LQ_DS_PHASE;
        for (int i = tid; i < L.total; i += LQM_T) lds[i] = 0.0;
        if (tid == 0) s_sing = 0;
        __syncthreads();
        if (a.Pt_dev) {
            const double *Pt = a.Pt_dev + (size_t)s * nX * nX;
            for (int e = tid; e < nX * nX; e += LQM_T) Pm[(e / nX) * ldx + e % nX] = Pt[e];
            if (affine && a.bt_dev) for (int i = tid; i < nX; i += LQM_T) bv[i] = a.bt_dev[(size_t)s * nX + i];
        } else {
            const double *Qf = a.Qf_dev + (size_t)s * a.Qf_seed_stride;
            for (int e = tid; e < nX * nX; e += LQM_T) Pm[(e / nX) * ldx + e % nX] = Qf[e];
            if (affine) for (int i = tid; i < nX; i += LQM_T) bv[i] = a.q_dev[(sN + s + N) * nX + i];
        }
        fill_AD(ke - 1, lds + L.AD);
        fill_B(ke - 1);
        if (tid < nq - nd) lds[L.avs + tid] = load_av(ke - 1);
        if (affine && tid < nU) lds[L.rn + tid] = a.r_dev[(sN + ke - 1) * nU + tid];
        __syncthreads();
    }
    // the upper-triangle tiles of the new P that this wave owns in phases 3 .. 5 (wave 0 factorises gamma meanwhile)
    int ti_[TSYM], tj_[TSYM];
    bool ok_[TSYM], mm_[TSYM];
    {
        LQ_DS_PHASE;
#pragma unroll
        for (int i = 0; i < TSYM; i++) {
            const int t = (wave < 4 ? wave - 1 : wave - 2) + TW * i;
            ok_[i] = t < NTRI && wave != 0 && wave != 4;
            int tr = 0, tt = ok_[i] ? t : 0;
#pragma unroll
            for (int r = 0; r < NT; r++) if (tt >= NT - r && tr == r) { tt -= NT - r; tr = r + 1; }
            ti_[i] = tr; tj_[i] = tr + tt;
            mm_[i] = ok_[i] && tr + tt < NTC;       // tiles in the v columns are Q_k alone
        }
    }
#if defined(TG_PROFILE)
    long long lq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lq_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (; k >= kb; k--) {
        double pav = 0.0, prn = 0.0;      // A[v_m][Qk_m] and r of step k - 1: requested at the head of the step, committed at its end
        {   // ---- phase 1: T = P A (NT x NTC tiles) and P B (NT x NUT tiles) as one list of chains ------------------------------------------
            LQ_DS_PHASE;
            if (k > kb) { fill_AD(k - 1, lds + L.AD + (cur ^ 1) * L.adsz); pav = load_av(k - 1); if (affine && tid < nU) prn = a.r_dev[(sN + k - 1) * nU + tid]; }
            if (a.b_next_dev && affine) for (int i = tid; i < nX; i += LQM_T) a.b_next_dev[(sN + k) * nX + i] = bv[i];    // b_{k+1}
            const int n_pa = NT * NTC, n_all = n_pa + NT * NUT;
            int t1_[T1], tc_[T1], bs_[T1];
            const double *bp_[T1];
            bool o1_[T1], pa_[T1];
#pragma unroll
            for (int i = 0; i < T1; i++) {
                const int t = wave + NW * i;
                o1_[i] = t < n_all; pa_[i] = t < n_pa;
                const int tt = o1_[i] ? (pa_[i] ? t : t - n_pa) : 0, w = pa_[i] ? NTC : NUT;
                t1_[i] = tt / w; tc_[i] = 16 * (tt - (tt / w) * w) + lr; acc[i] = zero4;
                bp_[i] = (pa_[i] ? AD : Bm) + tc_[i]; bs_[i] = pa_[i] ? lda : nU;
            }
            // operands of step k0 + 4 are requested before the matrix-core instructions of step k0 issue (two register sets, unrolled by two)
            auto ld = [&](int k0, double (&av)[T1], double (&bw)[T1]) {
                // no branch in this loop (a guarded load or matrix-core instruction becomes a basic block of its own, with a full
                // s_waitcnt in front of every one): chains without a tile recompute tile 0 into an accumulator nobody stores, and the
                // padding columns of a P B tile read whatever follows the row -- column j of a product depends on column j of B alone
                const int g0 = (grow(k0) + lk);
#pragma unroll
                for (int i = 0; i < T1; i++) {
                    av[i] = Pm[g0 * ldx + 16 * t1_[i] + lr];      // P[i][g] = P[g][i]
                    bw[i] = bp_[i][(pa_[i] ? k0 + lk : g0) * bs_[i]];
                }
            };
            auto mm = [&](const double (&av)[T1], const double (&bw)[T1]) {
#pragma unroll
                for (int i = 0; i < T1; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bw[i], acc[i], 0, 0, 0);
            };
            {
                double a0[T1], b0[T1], a1[T1], b1[T1];
                ld(0, a0, b0);
                int k0 = 0;
                for (; k0 + 8 <= KC; k0 += 8) {
                    ld(k0 + 4, a1, b1);
                    mm(a0, b0);
                    if (k0 + 8 < KC) ld(k0 + 8, a0, b0);
                    mm(a1, b1);
                }
                if (k0 < KC) mm(a0, b0);
            }

            // the single-entry rows outside the compact rows: one (P A tile: the v row of a Qk column) or two (P B tile: the Qk and the v row of
            // a rho column) extra terms per element.  Branch-free -- an element without such a row reads row 0 with a zero coefficient --
            // and all loads ahead of the FMAs: as guarded code this was ~50 basic blocks of one LDS round trip each (6.5 k cycles)
            {
                int e1[T1], e2[T1];
                double c1[T1], c2[T1], p1[T1][4], p2[T1][4];
#pragma unroll
                for (int i = 0; i < T1; i++) {
                    const int col = tc_[i];
                    const int ra = a_extra(col), rq = col < nU ? b_extra_q(col) : -1, rx = col < nU ? b_extra_v(col) : -1;
                    e1[i] = pa_[i] ? (ra >= 0 ? ra : 0) : (rq >= 0 ? rq : 0);
                    e2[i] = (!pa_[i] && rx >= 0) ? rx : 0;
                    const double xa = avs[(ra >= 0 ? col - nd : 0)], xq = Bm[(rq >= 0 ? rq : 0) * nU + (col < nU ? col : 0)], xv = Bm[(rx >= 0 ? rx : 0) * nU + (col < nU ? col : 0)];
                    c1[i] = pa_[i] ? (ra >= 0 ? xa : 0.0) : (rq >= 0 ? xq : 0.0);
                    c2[i] = (!pa_[i] && rx >= 0) ? xv : 0.0;
                }
#pragma unroll
                for (int i = 0; i < T1; i++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * t1_[i] + lk + 4 * r;
                        p1[i][r] = Pm[row * ldx + e1[i]]; p2[i][r] = Pm[row * ldx + e2[i]];
                    }
#pragma unroll
                for (int i = 0; i < T1; i++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[i][r] = fma(p2[i][r], c2[i], fma(p1[i][r], c1[i], acc[i][r]));
            }
            if (affine && tid < nU) {
                if (!bb_in_tile) {
                    double w = 0.0;
                    for (int i = 0; i < nX; i++) w += Bm[i * nU + tid] * bv[i];
                    wv[tid] = w;
                }
                rv[tid] = lds[L.rn + tid];      // r_k: fetched during the previous step
            }
            LQ_LDS_SYNC();
            LQ_STAMP(0);
            // ---- phase 2a: P A over P (its columns < ldc), P B and -- in its padding column -- b -------------------------------------
#pragma unroll
            for (int i = 0; i < T1; i++) {
                if (!o1_[i]) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * t1_[i] + lk + 4 * r;
                    if (pa_[i]) Pm[row * ldx + tc_[i]] = acc[i][r];
                    else if (tc_[i] < nU) PB[row * nUp + tc_[i]] = acc[i][r];
                }
            }
            if (bb_in_tile) for (int i = tid; i < ldx; i += LQM_T) PB[i * nUp + nU] = i < nX ? bv[i] : 0.0;
            LQ_LDS_SYNC();
        }
        {   // ---- phase 2b: gamma = R + B'(P B), Kpart = B'(P A) + S' -> G = [gamma | r + B'b | Kpart] ----------------------------------------
            LQ_DS_PHASE;
            // a wave's tiles (two at the puppet's sizes) run side by side, each with its k-loop split into two partial sums: four independent
            // chains of KC / 8 matrix-core instructions instead of two tiles x KC / 4 dependent ones one after the other
            constexpr int T2 = 2;
            const int n2 = NUT * (NUT + NTC);
            for (int tb = wave; tb < n2; tb += T2 * NW) {
                int tu_[T2], col_[T2];
                bool on_[T2], gam_[T2], gcol_[T2];
                v4d c0[T2], c1[T2];
#pragma unroll
                for (int i = 0; i < T2; i++) {
                    const int t = tb + NW * i;
                    on_[i] = t < n2;
                    const int tt = on_[i] ? t : 0, tu = tt / (NUT + NTC), tc = tt - tu * (NUT + NTC);
                    gam_[i] = tc < NUT; tu_[i] = tu; col_[i] = 16 * (gam_[i] ? tc : tc - NUT) + lr;
                    gcol_[i] = col_[i] < nU + (bb_in_tile ? 1 : 0);
                    c0[i] = zero4; c1[i] = zero4;
                }
                auto ld = [&](int k0, double (&av)[T2], double (&bw)[T2]) {
                    const int g0 = grow(k0) + lk;
#pragma unroll
                    for (int i = 0; i < T2; i++) {
                        // (branch-free like phase 1: rows u >= nU of B' and columns past gamma's are garbage in, garbage out, never stored)
                        av[i] = Bm[g0 * nU + 16 * tu_[i] + lr];       // B'[u][g]
                        bw[i] = (gam_[i] ? PB + g0 * nUp : Pm + g0 * ldx)[col_[i]];
                    }
                };
                {
                    double a0[T2], b0[T2], a1[T2], b1[T2];
                    ld(0, a0, b0);
                    int k0 = 0;
                    for (; k0 + 8 <= KC; k0 += 8) {
                        ld(k0 + 4, a1, b1);
#pragma unroll
                        for (int i = 0; i < T2; i++) c0[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[i], c0[i], 0, 0, 0);
                        if (k0 + 8 < KC) ld(k0 + 8, a0, b0);
#pragma unroll
                        for (int i = 0; i < T2; i++) c1[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[i], c1[i], 0, 0, 0);
                    }
                    if (k0 < KC) {
#pragma unroll
                        for (int i = 0; i < T2; i++) c0[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[i], c0[i], 0, 0, 0);
                    }
                }
                // rows Qk_m / v_m of B outside the compact rows: (B'X)[rho_m][.] += B[Qk_m][rho_m] X[Qk_m][.] + B[v_m][rho_m] X[v_m][.]
                // (branch-free, all loads first: see phase 1)
                double ex[T2][4];
                {
                    double bq[T2][4], bx[T2][4], xq[T2][4], xx[T2][4];
#pragma unroll
                    for (int i = 0; i < T2; i++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int uo = 16 * tu_[i] + lk + 4 * r, uc = uo < nU ? uo : 0;
                            const int rq = uo < nU ? b_extra_q(uo) : -1, rx = uo < nU ? b_extra_v(uo) : -1, rqc = rq >= 0 ? rq : 0, rxc = rx >= 0 ? rx : 0;
                            const double *X = gam_[i] ? PB : Pm;
                            const int xs = gam_[i] ? nUp : ldx;
                            const double q0 = Bm[rqc * nU + uc], x0 = Bm[rxc * nU + uc];
                            bq[i][r] = rq >= 0 ? q0 : 0.0; bx[i][r] = rx >= 0 ? x0 : 0.0;
                            xq[i][r] = X[rqc * xs + col_[i]]; xx[i][r] = X[rxc * xs + col_[i]];
                        }
#pragma unroll
                    for (int i = 0; i < T2; i++)
#pragma unroll
                        for (int r = 0; r < 4; r++) ex[i][r] = fma(bx[i][r], xx[i][r], bq[i][r] * xq[i][r]);
                }
#pragma unroll
                for (int i = 0; i < T2; i++) {
                    if (!on_[i]) continue;
                    const bool is_gamma = gam_[i];
                    const int col = col_[i];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int uo = 16 * tu_[i] + lk + 4 * r;      // output row (input index u), column `col`
                        if (uo < nU) {
                            double v = (c0[i][r] + c1[i][r]) + ex[i][r];
                            if (is_gamma) {
                                if (col < nU) {
                                    double g = v + a.R_dev[(size_t)s * a.R_seed_stride + (size_t)k * a.R_step_stride + uo * nU + col];
                                    if (hz) g += hz[(size_t)(nxh + uo) * hzR + nxh + col];
                                    G[uo * ldw + col] = g;
                                } else if (bb_in_tile && col == nU) {       // r_k + B'b
                                    const double rw = v + rv[uo];
                                    rv[uo] = rw; G[uo * ldw + nU] = rw;
                                }
                            } else {
                                if (hz && col < nxh) v += hz[(size_t)col * hzR + nxh + uo];
                                G[uo * ldw + nU + 1 + col] = v;
                            }
                        }
                    }
                }
            }
            if (tid < nU && !bb_in_tile) { const double rw = affine ? wv[tid] + rv[tid] : 0.0; rv[tid] = rw; G[tid * ldw + nU] = rw; }   // r_k + B'b
            LQ_LDS_SYNC();
            LQ_STAMP(1);
            if (k > kb) fill_B(k - 1);      // B_k is dead from here on
        }
        {   // ---- phase 3: wave 0 factorises gamma while the others accumulate Q_k + A'(P A) on the upper-triangle tiles --------------
            LQ_DS_PHASE;
            if (wave == 0) lq_factor_call<NR>(G, ldw, nU, lane, fac, &s_sing);
            else if (wave != 4) {
                const double *Qk = a.Q_dev + (size_t)s * a.Q_seed_stride + (size_t)k * a.Q_step_stride;
                // the weights Q_k (+ curvature) come from global memory: requested here, added behind the chains (as the chains' starting
                // values they put a global-memory latency in front of the first matrix-core instruction)
                v4d qw[TSYM];
#pragma unroll
                for (int i = 0; i < TSYM; i++) {
                    v4d c = zero4;
                    const int col = 16 * tj_[i] + lr;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * ti_[i] + lk + 4 * r;
                        if (ok_[i] && row < nX && col < nX) {
                            c[r] = Qk[(size_t)row * nX + col];
                            if (hz && row < nxh && col < nxh) c[r] += hz[(size_t)row * hzR + col];
                        }
                    }
                    qw[i] = c; acc[i] = zero4;
                }
                // each tile's k-loop in two partial sums (2 x TSYM independent chains), the next step's operands requested ahead
                v4d acc2[TSYM];
#pragma unroll
                for (int i = 0; i < TSYM; i++) acc2[i] = zero4;
                auto ld = [&](int k0, double (&av)[TSYM], double (&bw)[TSYM]) {
                    const int g0 = grow(k0) + lk;
#pragma unroll
                    for (int i = 0; i < TSYM; i++) {
                        const double x = AD[(k0 + lk) * lda + (mm_[i] ? 16 * ti_[i] + lr : 0)];
                        av[i] = mm_[i] ? x : 0.0;                                           // A'[i][g] = A[g][i]; a tile of the v columns adds zero
                        bw[i] = Pm[g0 * ldx + 16 * tj_[i] + lr];                            // (P A)[g][j]
                    }
                };
                {
                    double a0[TSYM], b0[TSYM], a1[TSYM], b1[TSYM];
                    ld(0, a0, b0);
                    int k0 = 0;
                    for (; k0 + 8 <= KC; k0 += 8) {
                        ld(k0 + 4, a1, b1);
#pragma unroll
                        for (int i = 0; i < TSYM; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[i], acc[i], 0, 0, 0);
                        if (k0 + 8 < KC) ld(k0 + 8, a0, b0);
#pragma unroll
                        for (int i = 0; i < TSYM; i++) acc2[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[i], acc2[i], 0, 0, 0);
                    }
                    if (k0 < KC) {
#pragma unroll
                        for (int i = 0; i < TSYM; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[i], acc[i], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < TSYM; i++) acc[i] = (acc[i] + acc2[i]) + qw[i];
                {   // (A'(P A))[Qk_m][j] += A[v_m][Qk_m] (P A)[v_m][j] for the v rows outside the compact rows (branch-free, loads first)
                    double cx[TSYM][4], px[TSYM][4];
#pragma unroll
                    for (int i = 0; i < TSYM; i++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = 16 * ti_[i] + lk + 4 * r, rx = a_extra(row), col = 16 * tj_[i] + lr;
                            const double c = avs[rx >= 0 ? row - nd : 0];
                            cx[i][r] = (mm_[i] && rx >= 0) ? c : 0.0;
                            px[i][r] = Pm[(rx >= 0 ? rx : 0) * ldx + col];
                        }
#pragma unroll
                    for (int i = 0; i < TSYM; i++)
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[i][r] = fma(cx[i][r], px[i][r], acc[i][r]);
                }
            }
            LQ_STAMP(6);
            // the buffer that held P B becomes K: padding rows and the v columns (never solved: their right-hand sides are zero) are zero
            for (int o = tid; o < (nUp - nU) * ldx; o += LQM_T) Ks[nU * ldx + o] = 0.0;
            for (int o = tid; o < nU * (ldx - ldc); o += LQM_T) { const int u = o / (ldx - ldc); Ks[u * ldx + ldc + (o - u * (ldx - ldc))] = 0.0; }
            LQ_LDS_SYNC();
        }
        {   // ---- phase 4: every wave replays the factorisation on its slice of the right-hand sides --------------------------------------
            LQ_DS_PHASE;
            const int rhs_total = 1 + ldc, slice = (rhs_total + NW - 1) / NW;
            const int lo = wave * slice, nrhs = lo < rhs_total ? (rhs_total - lo < slice ? rhs_total - lo : slice) : 0;
            LQ_STAMP(7);      // (diagnostic build: up to here the wait for the factorisation / the tiles; from here the gains)
            if (*reinterpret_cast<const int *>(fac + 64)) {
                // [C | K] = gamma^-1 [r + B'b | Kpart] on the matrix cores: NUT x NTC tiles of K (five k-steps each), C as 18 dot products
                for (int t = wave; t < NUT * NTC; t += NW) {
                    const int tu = t / NTC, tj = t - tu * NTC, u = 16 * tu + lr, col = 16 * tj + lr;
                    v4d c = zero4;
                    for (int v0 = 0; v0 < nUp; v0 += 4) {
                        const double x = G[(u < nU ? u : 0) * ldw + (v0 + lk < nU ? v0 + lk : 0)];
                        const double av = (u < nU && v0 + lk < nU) ? x : 0.0;                  // gamma^-1 [u][v]
                        const double bw = G[(v0 + lk) * ldw + nU + 1 + col];                    // Kpart [v][col] (rows nU .. nUp-1 of G are zero)
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bw, c, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int uo = 16 * tu + lk + 4 * r; if (uo < nU) Ks[uo * ldx + col] = c[r]; }
                }
                if (tid < nU) {
                    double acc = 0.0;
                    for (int v = 0; v < nU; v++) acc = fma(G[tid * ldw + v], G[v * ldw + nU], acc);
                    wv[tid] = acc;
                }
            } else
            lq_apply_rows<NR, (16 * NT + 1 + NW - 1) / NW>(G, ldw, nU, lo, nrhs, Ks, ldx, wv, lane, fac);
            LQ_LDS_SYNC();
            LQ_STAMP(2);
        }
        {   // ---- outputs K_k, C_k; new P tiles -= Kpart' K; new b ---------------------------------------------------------------------
            LQ_DS_PHASE;
            double *Cs = wv;
            double *Ko = a.K_dev + (sN + k) * (size_t)nU * nX;
            for (int o = tid; o < nU * nX; o += LQM_T) Ko[o] = Ks[(o / nX) * ldx + o % nX];
            if (a.C_dev && tid < nU) a.C_dev[(sN + k) * nU + tid] = Cs[tid];
            for (int u0 = 0; u0 < nUp; u0 += 4) {       // acc -= Kpart' K  (Kpart: the right-hand-side block of G, untouched by the solve)
                double av[TSYM], bw[TSYM];
#pragma unroll
                for (int i = 0; i < TSYM; i++) {
                    const double x = G[(u0 + lk) * ldw + nU + 1 + 16 * ti_[i] + lr];      // (rows nU .. nUp-1 of G exist and are zero)
                    av[i] = mm_[i] ? -x : 0.0;
                    bw[i] = Ks[(u0 + lk) * ldx + 16 * tj_[i] + lr];
                }
#pragma unroll
                for (int i = 0; i < TSYM; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bw[i], acc[i], 0, 0, 0);
            }
            if (affine) for (int i = tid; i < nX; i += LQM_T) {     // new b = q_k + A'b - K'(r_k + B'b)
                const double qi = a.q_dev[(sN + s + k) * nX + i];      // (added last: the chains do not wait for it)
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
                if (i < ldc) {
                    for (int m = 0; m + 4 <= KC; m += 4) {
                        const int g0 = grow(m);
                        v0 = fma(AD[m * lda + i], bv[g0], v0); v1 = fma(AD[(m + 1) * lda + i], bv[g0 + 1], v1);
                        v2 = fma(AD[(m + 2) * lda + i], bv[g0 + 2], v2); v3 = fma(AD[(m + 3) * lda + i], bv[g0 + 3], v3);
                    }
                    const int rx = a_extra(i);
                    if (rx >= 0) v0 = fma(avs[i - nd], bv[rx], v0);
                }
                int u = 0;
                for (; u + 2 <= nU; u += 2) { v1 = fma(-Ks[u * ldx + i], rv[u], v1); v2 = fma(-Ks[(u + 1) * ldx + i], rv[u + 1], v2); }
                for (; u < nU; u++) v3 = fma(-Ks[u * ldx + i], rv[u], v3);
                bn[i] = ((v0 + v1) + (v2 + v3)) + qi;
            }
            LQ_LDS_SYNC();
            LQ_STAMP(3);
        }
        {   // ---- phase 5: P <- new upper tiles, b <- new b; phase 6: mirror the upper triangle (see k_tv_lq_mfma) ---------------------------
            LQ_DS_PHASE;
#pragma unroll
            for (int i = 0; i < TSYM; i++) {
                if (ok_[i]) {
#pragma unroll
                    for (int r = 0; r < 4; r++) Pm[(16 * ti_[i] + lk + 4 * r) * ldx + 16 * tj_[i] + lr] = acc[i][r];
                }
            }
            if (affine) for (int i = tid; i < nX; i += LQM_T) bv[i] = bn[i];
            if (k > kb && tid < nq - nd) lds[L.avs + (cur ^ 1) * 32 + tid] = pav;
            if (k > kb && affine && tid < nU) lds[L.rn + tid] = prn;
            LQ_LDS_SYNC();
            LQ_STAMP(4);
            for (int d = 1 + wave; d < nX; d += NW)
                for (int i = lane; i < nX - d; i += 64) {
                    const bool diag_tile = (i >> 4) == ((i + d) >> 4);
                    const double up = Pm[i * ldx + i + d];
                    const double v = diag_tile ? 0.5 * (up + Pm[(i + d) * ldx + i]) : up;
                    Pm[i * ldx + i + d] = v; Pm[(i + d) * ldx + i] = v;
                }
        }
        cur ^= 1;
        __syncthreads();           // also waits for this wave's global -> LDS loads of A_{k-1}, B_{k-1}
        LQ_STAMP(5);
    }
#if defined(TG_PROFILE)
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 8; i++) g_lq_prof[i] = lq_acc[i];
#endif
    {
        k = 0;
        LQ_DS_PHASE;
        if (a.P0_dev) for (int e = tid; e < nX * nX; e += LQM_T) a.P0_dev[(size_t)s * nX * nX + e] = Pm[(e / nX) * ldx + e % nX];
        if (a.b0_dev && affine) for (int i = tid; i < nX; i += LQM_T) a.b0_dev[(size_t)s * nX + i] = bv[i];
        if (a.status_dev && tid == 0) a.status_dev[s] = (s_sing || (a.Pt_dev && a.status_dev[s] != TG_OK)) ? TG_SINGULAR : TG_OK;      // (a later chunk of a chunked sweep keeps an earlier chunk's verdict)
    }
}

size_t lq_lds_bytes(int nX, int nU, int ts) {
    const int ldx = round_up(nX, ts), ldw = nU + 1 + ldx;
    return sizeof(double) * ((size_t)2 * ldx * ldx + (size_t)ldx * nU + 2 * (size_t)nU * ldx + (size_t)nU * ldw + 2 * ldx + 3 * nU);
}

// ------------------------------------------------------------------------------------------------------
// Matrix-vector sweeps along k (one workgroup per seed, A_k / B_k staged in LDS with a padded row stride).
// ------------------------------------------------------------------------------------------------------
constexpr int SW_T = 256;

// A_k (nX x nX, row stride lda in LDS), B_k and a third nU x nX block (K_k) travel global -> registers -> LDS: the
// loads of step k+1 are issued while step k computes (2-D mapping, 256-byte row segments, no integer division).
struct SweepStage {
    static constexpr int RI = 12, CI = 3, PB = 12;   // nX <= 96, nX * nU <= 12 * 256
    double a[RI * CI], b[PB], c[PB];
    __device__ void load(const double *Ak, const double *Bk, const double *Ck, int nX, int nU, int tid) {
        const int r0 = tid >> 5, c0 = tid & 31;
#pragma unroll
        for (int ri = 0; ri < RI; ri++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) {
                const int r = r0 + 8 * ri, cc = c0 + 32 * ci;
                if (r < nX && cc < nX) a[ri * CI + ci] = Ak[r * nX + cc];
            }
#pragma unroll
        for (int i = 0; i < PB; i++) { const int e = tid + i * SW_T; if (e < nX * nU) { b[i] = Bk[e]; c[i] = Ck[e]; } }
    }
    // A -> Am[r * lda + c]; B -> Bm (row-major nX x nU); C (nU x nX) -> Cm[(e / nX) * ldc + e % nX] when ldc != nX
    __device__ void store(double *Am, double *Bm, double *Cm, int nX, int nU, int lda, int ldc, int tid) const {
        const int r0 = tid >> 5, c0 = tid & 31;
#pragma unroll
        for (int ri = 0; ri < RI; ri++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) {
                const int r = r0 + 8 * ri, cc = c0 + 32 * ci;
                if (r < nX && cc < nX) Am[r * lda + cc] = a[ri * CI + ci];
            }
#pragma unroll
        for (int i = 0; i < PB; i++) {
            const int e = tid + i * SW_T;
            if (e < nX * nU) { Bm[e] = b[i]; Cm[ldc == nX ? e : (e / nX) * ldc + e % nX] = c[i]; }
        }
    }
};

// Backward adjoint of doptimizer.py:319-345: Z_k = z_{k+1} (the vector the second derivatives of step k are
// contracted with), z_k = q_k - K_k' r_k + (A_k - B_k K_k)' z_{k+1}, z_N = q_N.
__global__ __launch_bounds__(SW_T) void k_adjoint(int N, int nX, int nU, const int *sel, const double *A, const double *B,
                                                  const double *K, const double *q, const double *r, double *Z) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, s = sel ? sel[blockIdx.x] : blockIdx.x;
    const int lda = nX | 1;
    double *Am = lds, *Bm = Am + nX * lda, *Km = Bm + nX * nU, *z = Km + nU * nX, *zn = z + nX, *w = zn + nX;
    const size_t sN = (size_t)s * N;
    for (int i = tid; i < nX; i += SW_T) z[i] = q[(sN + s + N) * nX + i];
    SweepStage st;
    st.load(A + (sN + N - 1) * (size_t)nX * nX, B + (sN + N - 1) * (size_t)nX * nU, K + (sN + N - 1) * (size_t)nU * nX, nX, nU, tid);
    __syncthreads();
    for (int k = N - 1; k >= 0; k--) {
        st.store(Am, Bm, Km, nX, nU, lda, nX, tid);
        if (k > 0) st.load(A + (sN + k - 1) * (size_t)nX * nX, B + (sN + k - 1) * (size_t)nX * nU, K + (sN + k - 1) * (size_t)nU * nX, nX, nU, tid);
        for (int i = tid; i < nX; i += SW_T) Z[(sN + k) * nX + i] = z[i];
        __syncthreads();
        if (tid < nU) {  // w = r_k + B' z
            double v0 = r[(sN + k) * nU + tid], v1 = 0.0, v2 = 0.0, v3 = 0.0;
            int i = 0;
            for (; i + 3 < nX; i += 4) {
                v0 += Bm[i * nU + tid] * z[i]; v1 += Bm[(i + 1) * nU + tid] * z[i + 1];
                v2 += Bm[(i + 2) * nU + tid] * z[i + 2]; v3 += Bm[(i + 3) * nU + tid] * z[i + 3];
            }
            for (; i < nX; i++) v0 += Bm[i * nU + tid] * z[i];
            w[tid] = (v0 + v1) + (v2 + v3);
        }
        __syncthreads();
        for (int i = tid; i < nX; i += SW_T) {
            double v0 = q[(sN + s + k) * nX + i], v1 = 0.0, v2 = 0.0, v3 = 0.0;
            int m = 0;
            for (; m + 3 < nX; m += 4) {
                v0 += Am[m * lda + i] * z[m]; v1 += Am[(m + 1) * lda + i] * z[m + 1];
                v2 += Am[(m + 2) * lda + i] * z[m + 2]; v3 += Am[(m + 3) * lda + i] * z[m + 3];
            }
            for (; m < nX; m++) v0 += Am[m * lda + i] * z[m];
            double v = (v0 + v1) + (v2 + v3);
            for (int u = 0; u < nU; u++) v -= Km[u * nX + i] * w[u];
            zn[i] = v;
        }
        __syncthreads();
        for (int i = tid; i < nX; i += SW_T) z[i] = zn[i];
        __syncthreads();
    }
}

// Forward tangent rollout of doptimizer.py:391-402: dU_k = -K_k dX_k - C_k, dX_{k+1} = A_k dX_k + B_k dU_k,
// dX_0 = 0; also dcost = sum_k q_k.dX_k + r_k.dU_k (calc_dcost, :262-270).
__global__ __launch_bounds__(SW_T) void k_tangent(int N, int nX, int nU, const int *sel, const double *A, const double *B,
                                                  const double *K, const double *C, const double *q, const double *r,
                                                  double *dX, double *dU, double *dcost) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, s = sel ? sel[blockIdx.x] : blockIdx.x;
    const int lda = nX | 1;
    double *Am = lds, *Bm = Am + nX * lda, *Km = Bm + nX * nU, *x = Km + nU * lda, *xn = x + nX, *u = xn + nX, *red = u + nU;
    const size_t sN = (size_t)s * N;
    for (int i = tid; i < nX; i += SW_T) { x[i] = 0.0; dX[(sN + s) * nX + i] = 0.0; }
    double part = 0.0;
    SweepStage st;
    st.load(A + sN * (size_t)nX * nX, B + sN * (size_t)nX * nU, K + sN * (size_t)nU * nX, nX, nU, tid);
    __syncthreads();
    for (int k = 0; k < N; k++) {
        st.store(Am, Bm, Km, nX, nU, lda, lda, tid);
        if (k + 1 < N) st.load(A + (sN + k + 1) * (size_t)nX * nX, B + (sN + k + 1) * (size_t)nX * nU, K + (sN + k + 1) * (size_t)nU * nX, nX, nU, tid);
        __syncthreads();
        if (tid < nU) {
            double v0 = -C[(sN + k) * nU + tid], v1 = 0.0, v2 = 0.0, v3 = 0.0;
            int m = 0;
            for (; m + 3 < nX; m += 4) {
                v0 -= Km[tid * lda + m] * x[m]; v1 -= Km[tid * lda + m + 1] * x[m + 1];
                v2 -= Km[tid * lda + m + 2] * x[m + 2]; v3 -= Km[tid * lda + m + 3] * x[m + 3];
            }
            for (; m < nX; m++) v0 -= Km[tid * lda + m] * x[m];
            const double v = (v0 + v1) + (v2 + v3);
            u[tid] = v;
            dU[(sN + k) * nU + tid] = v;
            part += r[(sN + k) * nU + tid] * v;
        }
        for (int i = tid; i < nX; i += SW_T) part += q[(sN + s + k) * nX + i] * x[i];
        __syncthreads();
        for (int i = tid; i < nX; i += SW_T) {
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
            int m = 0;
            for (; m + 3 < nX; m += 4) {
                v0 += Am[i * lda + m] * x[m]; v1 += Am[i * lda + m + 1] * x[m + 1];
                v2 += Am[i * lda + m + 2] * x[m + 2]; v3 += Am[i * lda + m + 3] * x[m + 3];
            }
            for (; m < nX; m++) v0 += Am[i * lda + m] * x[m];
            double v = (v0 + v1) + (v2 + v3);
            for (int c = 0; c < nU; c++) v += Bm[i * nU + c] * u[c];
            xn[i] = v;
            dX[(sN + s + k + 1) * nX + i] = v;
        }
        __syncthreads();
        for (int i = tid; i < nX; i += SW_T) x[i] = xn[i];
        __syncthreads();
    }
    for (int i = tid; i < nX; i += SW_T) part += q[(sN + s + N) * nX + i] * x[i];
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double t_ = 0.0;
        for (int i = 0; i < SW_T; i++) t_ += red[i];
        dcost[s] = t_;
    }
}

// The same rollout with the matrices in REGISTERS and the dot products spread over lanes: thread (row i, part p = tid & 3) keeps a
// quarter of row i of A_k and of B_k, thread (input j, part tid & 7) an eighth of row j of K_k -- loaded straight from global memory
// (a wavefront reads 16 whole rows, contiguous), one step ahead -- and the partial sums meet through DPP (quad_perm / row_half_mirror).
// k_tangent stages A, B, K in LDS every step and has one lane walk a whole row, an LDS round trip per four entries: 8 us a step
// where the arithmetic is 0.2.  Needs 4 nX and 8 nU threads (nX <= 96, nU <= 32).
constexpr int TR_MAX_CA = 24, TR_MAX_CB = 8, TR_MAX_CK = 12;   // most columns of A / B / K a thread holds (k_tangent_rows<CA, CB, CK>: the sizes compiled in)
template <int CTRL> __device__ __forceinline__ double tr_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// Thread (row i, part p) holds columns p, p + 4, p + 8, ... of its row (a quad reads 32 contiguous bytes per load); loads are unconditional
// -- rows and columns past the end are clamped to the last one and meet the ZERO padding of dX / dU in LDS -- because a guarded global
// load becomes a basic block of its own (47 of them per step, each with its address arithmetic and exec-mask juggling).
// PAIR (even nX and nU): the thread holds PAIRS of neighbouring columns -- 2 p, 2 p + 1, then 8 (A, B) or 16 (K) columns on -- and loads each
// pair with one 16-byte instruction: half the vector-memory instructions per step.  The step is bound by them: without its loads the kernel
// takes 1.7 us a step, with the 47 8-byte loads per thread 4.6 (32-byte pieces of 640-byte rows: the address coalescer, not HBM).
template <int CA, int CB, int CK, bool PAIR = false> struct TangentRegs {
    static constexpr int TR_CA = CA, TR_CB = CB, TR_CK = CK;
    double a[TR_CA], b[TR_CB], k[TR_CK], cq, cr, cc;     // matrix slices; q_k[i], r_k[j], C_k[j] of the thread's row
    // column of slot c of a part-p thread when parts are `np` wide: the plain interleave p + np c, or pairs (2 p + (c & 1)) + 2 np (c >> 1)
    __device__ __forceinline__ static int col_of(int p, int c, int np) { return PAIR ? 2 * p + (c & 1) + 2 * np * (c >> 1) : p + np * c; }
    __device__ __forceinline__ void load(const double *Ak, const double *Bk, const double *Kk, const double *Ck, const double *qk, const double *rk,
                                         int nX, int nU, int row, int pa, int nca, int ncb, int jrow, int pk, int nck) {
        const double *ar = Ak + (size_t)(row < nX ? row : nX - 1) * nX, *br = Bk + (size_t)(row < nX ? row : nX - 1) * nU;
        const double *kr = Kk + (size_t)(jrow < nU ? jrow : nU - 1) * nX;
        if constexpr (PAIR) {
            typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int c = 0; c < TR_CA; c += 2) if (c < nca) { const int col = col_of(pa, c, 4); const d2 v = *(const d2 *)(ar + (col < nX ? col : nX - 2)); a[c] = v[0]; a[c + 1] = v[1]; }
#pragma unroll
            for (int c = 0; c < TR_CB; c += 2) if (c < ncb) { const int col = col_of(pa, c, 4); const d2 v = *(const d2 *)(br + (col < nU ? col : nU - 2)); b[c] = v[0]; b[c + 1] = v[1]; }
#pragma unroll
            for (int c = 0; c < TR_CK; c += 2) if (c < nck) { const int col = col_of(pk, c, 8); const d2 v = *(const d2 *)(kr + (col < nX ? col : nX - 2)); k[c] = v[0]; k[c + 1] = v[1]; }
        } else {
#pragma unroll
        for (int c = 0; c < TR_CA; c++) if (c < nca) { const int col = pa + 4 * c; a[c] = ar[col < nX ? col : nX - 1]; }
#pragma unroll
        for (int c = 0; c < TR_CB; c++) if (c < ncb) { const int col = pa + 4 * c; b[c] = br[col < nU ? col : nU - 1]; }
#pragma unroll
        for (int c = 0; c < TR_CK; c++) if (c < nck) { const int col = pk + 8 * c; k[c] = kr[col < nX ? col : nX - 1]; }
        }
        cq = qk[row < nX ? row : nX - 1];
        cr = rk[jrow < nU ? jrow : nU - 1];
        cc = Ck[jrow < nU ? jrow : nU - 1];
    }
};
template <int TR_CA, int TR_CB, int TR_CK, bool PAIR = false>
__global__ __launch_bounds__(384) void k_tangent_rows(int N, int nX, int nU, const int *sel, const double *A, const double *B,
                                                      const double *K, const double *C, const double *q, const double *r,
                                                      double *dX, double *dU, double *dcost) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, nt = blockDim.x, s = sel ? sel[blockIdx.x] : blockIdx.x;
    typedef TangentRegs<TR_CA, TR_CB, TR_CK, PAIR> Regs;
    const int row = tid >> 2, pa = tid & 3, nca = PAIR ? 2 * ((nX + 7) >> 3) : (nX + 3) >> 2, ncb = PAIR ? 2 * ((nU + 7) >> 3) : (nU + 3) >> 2;
    const int jrow = tid >> 3, pk = tid & 7, nck = PAIR ? 2 * ((nX + 15) >> 4) : (nX + 7) >> 3;
    const int px = 8 * nck, pu = 4 * ncb;                       // padded lengths of dX and dU in LDS (the padding stays zero)
    double *x0 = lds, *x1 = x0 + px, *u = x1 + px, *red = u + pu;
    const size_t sN = (size_t)s * N;
    for (int i = tid; i < 2 * px + pu; i += nt) lds[i] = 0.0;
    for (int i = tid; i < nX; i += nt) dX[(sN + s) * nX + i] = 0.0;
    double part = 0.0;
    Regs cur, nxt;
    cur.load(A + sN * (size_t)nX * nX, B + sN * (size_t)nX * nU, K + sN * (size_t)nU * nX, C + sN * nU, q + (sN + s) * nX, r + sN * nU, nX, nU, row, pa, nca, ncb, jrow, pk, nck);
    __syncthreads();
    double *x = x0, *xn = x1;
    for (int k = 0; k < N; k++) {
        const size_t kn = sN + (k + 1 < N ? k + 1 : k);
        nxt.load(A + kn * (size_t)nX * nX, B + kn * (size_t)nX * nU, K + kn * (size_t)nU * nX, C + kn * nU, q + (kn + s) * nX, r + kn * nU, nX, nU, row, pa, nca, ncb, jrow, pk, nck);
        // ---- dU = -K dX - C: eight lanes per input
        {
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int c = 0; c < TR_CK; c += 2) {
                if (c < nck) v0 = fma(cur.k[c], x[Regs::col_of(pk, c, 8)], v0);
                if (c + 1 < nck) v1 = fma(cur.k[c + 1], x[Regs::col_of(pk, c + 1, 8)], v1);
            }
            double v = v0 + v1;
            v += tr_dpp<0xB1>(v); v += tr_dpp<0x4E>(v); v += tr_dpp<0x141>(v);     // quad xor 1, xor 2, mirror of the half row
            if (jrow < nU && pk == 0) {
                const double uj = -cur.cc - v;
                u[jrow] = uj;
                dU[(sN + k) * nU + jrow] = uj;
                part = fma(cur.cr, uj, part);
            }
        }
        if (row < nX && pa == 0) part = fma(cur.cq, x[row], part);
        __syncthreads();
        // ---- dX' = A dX + B dU: four lanes per state
        {
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
#pragma unroll
            for (int c = 0; c < TR_CA; c += 4) {
                if (c < nca) v0 = fma(cur.a[c], x[Regs::col_of(pa, c, 4)], v0);
                if (c + 1 < nca) v1 = fma(cur.a[c + 1], x[Regs::col_of(pa, c + 1, 4)], v1);
                if (c + 2 < nca) v2 = fma(cur.a[c + 2], x[Regs::col_of(pa, c + 2, 4)], v2);
                if (c + 3 < nca) v3 = fma(cur.a[c + 3], x[Regs::col_of(pa, c + 3, 4)], v3);
            }
#pragma unroll
            for (int c = 0; c < TR_CB; c += 2) {
                if (c < ncb) v0 = fma(cur.b[c], u[Regs::col_of(pa, c, 4)], v0);
                if (c + 1 < ncb) v1 = fma(cur.b[c + 1], u[Regs::col_of(pa, c + 1, 4)], v1);
            }
            double v = (v0 + v1) + (v2 + v3);
            v += tr_dpp<0xB1>(v); v += tr_dpp<0x4E>(v);
            if (row < nX && pa == 0) { xn[row] = v; dX[(sN + s + k + 1) * nX + row] = v; }
        }
        __syncthreads();
        double *t_ = x; x = xn; xn = t_;
        cur = nxt;
    }
    if (tid < nX) part = fma(q[(sN + s + N) * nX + tid], x[tid], part);
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double t_ = 0.0;
        for (int i = 0; i < nt; i++) t_ += red[i];
        dcost[s] = t_;
    }
}

// ------------------------------------------------------------------------------------------------------
// Quadratic tracking cost (dcost.py:5-118): l = 1/2 (x-xd)'Q(x-xd) + 1/2 (u-ud)'R(u-ud), m = 1/2 (x-xd)'Qf(x-xd).
// Trajectory t of `group` candidates per seed compares against the reference of seed t / group.
// ------------------------------------------------------------------------------------------------------
constexpr int CT_T = 256;

__global__ __launch_bounds__(CT_T) void k_cost(int N, int nX, int nU, int group, const int *sel, const double *X, const double *U,
                                                const double *Xd, const double *Ud, const double *Q, const double *R,
                                                const double *Qf, double *cost) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t t = blockIdx.x, s = sel ? (size_t)sel[t / group] : t / group;
    double *Qt = lds, *Rt = Qt + nX * nX, *Qft = Rt + nU * nU, *dx = Qft + nX * nX + wave * (nX + nU), *du = dx + nX;
    double *red = Qft + nX * nX + 4 * (nX + nU);
    for (int e = tid; e < nX * nX; e += CT_T) { Qt[(e % nX) * nX + e / nX] = Q[e]; Qft[(e % nX) * nX + e / nX] = Qf[e]; }
    for (int e = tid; e < nU * nU; e += CT_T) Rt[(e % nU) * nU + e / nU] = R[e];
    __syncthreads();
    double part = 0.0;
    for (int k0 = 0; k0 <= N; k0 += 4) {  // each wavefront takes one of four consecutive time steps
        const int k = k0 + wave;
        if (k <= N) {
            const double *xk = X + (t * (N + 1) + k) * nX, *xd = Xd + (s * (N + 1) + k) * nX;
            for (int i = lane; i < nX; i += 64) dx[i] = xk[i] - xd[i];
            if (k < N) {
                const double *uk = U + (t * N + k) * nU, *ud = Ud + (s * N + k) * nU;
                for (int i = lane; i < nU; i += 64) du[i] = uk[i] - ud[i];
            }
        }
        __syncthreads();
        if (k <= N) {
            const double *W = k < N ? Qt : Qft;
            for (int i = lane; i < nX; i += 64) {
                double v = 0.0;
                for (int j = 0; j < nX; j++) v += W[j * nX + i] * dx[j];
                part += 0.5 * dx[i] * v;
            }
            if (k < N) for (int i = lane; i < nU; i += 64) {
                double v = 0.0;
                for (int j = 0; j < nU; j++) v += Rt[j * nU + i] * du[j];
                part += 0.5 * du[i] * v;
            }
        }
        __syncthreads();
    }
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double t_ = 0.0;
        for (int i = 0; i < CT_T; i++) t_ += red[i];
        cost[t] = t_;
    }
}

// The same cost on the matrix cores.  k_cost keeps Q', Qf' and R' in LDS (110 KB at the puppet's size: one workgroup per CU) and walks
// the horizon four steps at a time behind two barriers per trip -- 5.6 us per trip, 11 ms for the 2048 candidates of one Armijo
// round, 7 % of a discopt iteration.  Here a wavefront takes SIXTEEN consecutive steps as the rows of one v_mfma_f64_16x16x4 tile
// row: D = DX Q (DX [16][nX] straight from global memory into the A operand, Q row-major in LDS as the B operand, NT independent
// accumulator chains), then the cost of those steps is sum D .* DX with DX re-read in the accumulator layout; likewise DU R.  The
// terminal step (weight Qf) is one wavefront's dot products.  No barrier inside the sweep; only Q and R are staged (52 KB).
template <int NT>
__global__ __launch_bounds__(CT_T) void k_cost_mfma(int N, int nX, int nU, int group, const int *sel, const double *X, const double *U,
                                                     const double *Xd, const double *Ud, const double *Q, const double *R,
                                                     const double *Qf, double *cost) {
    extern __shared__ double lds[];
    constexpr int ldx = 16 * NT, NW = CT_T / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const size_t t = blockIdx.x, s = sel ? (size_t)sel[t / group] : t / group;
    const int ldu = round_up(nU, 16), NUT = ldu >> 4;
    double *Qm = lds, *Rm = Qm + ldx * ldx, *red = Rm + ldu * ldu;
    for (int e = tid; e < ldx * ldx; e += CT_T) { const int r = e / ldx, c = e % ldx; Qm[e] = (r < nX && c < nX) ? Q[r * nX + c] : 0.0; }
    for (int e = tid; e < ldu * ldu; e += CT_T) { const int r = e / ldu, c = e % ldu; Rm[e] = (r < nU && c < nU) ? R[r * nU + c] : 0.0; }
    __syncthreads();
    const double *Xt = X + t * (size_t)(N + 1) * nX, *Xs = Xd + s * (size_t)(N + 1) * nX;
    const double *Ut = U + t * (size_t)N * nU, *Us = Ud + s * (size_t)N * nU;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;
    const int n_row_tiles = (N + 15) >> 4;          // steps 0 .. N-1 carry Q and R; step N (Qf) is handled below
    for (int rt = wave; rt < n_row_tiles; rt += NW) {
        // ---- state part: rows = steps 16 rt .. 16 rt + 15 (rows >= N contribute nothing) ---------------------------------------
        {
            const int row = 16 * rt + lr;
            double av[ldx / 4];
#pragma unroll
            for (int ks = 0; ks < ldx / 4; ks++) {
                const int c = 4 * ks + lk;
                av[ks] = (row < N && c < nX) ? Xt[(size_t)row * nX + c] - Xs[(size_t)row * nX + c] : 0.0;
            }
            v4d acc[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = zero4;
#pragma unroll
            for (int ks = 0; ks < ldx / 4; ks++) {
#pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], Qm[(4 * ks + lk) * ldx + 16 * j + lr], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rr = 16 * rt + lk + 4 * r, c = 16 * j + lr;
                    if (rr < N && c < nX) part += 0.5 * acc[j][r] * (Xt[(size_t)rr * nX + c] - Xs[(size_t)rr * nX + c]);
                }
        }
        // ---- input part -----------------------------------------------------------------------------------------------------
        {
            const int row = 16 * rt + lr;
            for (int j = 0; j < NUT; j++) {
                v4d acc = zero4;
                for (int k0 = 0; k0 < ldu; k0 += 4) {
                    const int c = k0 + lk;
                    const double a_ = (row < N && c < nU) ? Ut[(size_t)row * nU + c] - Us[(size_t)row * nU + c] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, Rm[(k0 + lk) * ldu + 16 * j + lr], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rr = 16 * rt + lk + 4 * r, c = 16 * j + lr;
                    if (rr < N && c < nU) part += 0.5 * acc[r] * (Ut[(size_t)rr * nU + c] - Us[(size_t)rr * nU + c]);
                }
            }
        }
    }
    if (wave == NW - 1) {     // terminal cost 1/2 dx_N' Qf dx_N: one output column per lane
        const double *xk = Xt + (size_t)N * nX, *xd = Xs + (size_t)N * nX;
        for (int i = lane; i < nX; i += 64) {
            double v = 0.0;
            for (int j = 0; j < nX; j++) v += Qf[j * nX + i] * (xk[j] - xd[j]);
            part += 0.5 * (xk[i] - xd[i]) * v;
        }
    }
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {      // fixed summation order: the result does not depend on timing
        double t_ = 0.0;
        for (int i = 0; i < CT_T; i++) t_ += red[i];
        cost[t] = t_;
    }
}

// gradients q_k = (x_k - xd_k)'Q (k < N), q_N = (x_N - xd_N)'Qf, r_k = (u_k - ud_k)'R  (dcost.py:62-84)
__global__ __launch_bounds__(CT_T) void k_cost_grad(int N, int nX, int nU, const int *sel, const double *X, const double *U,
                                                     const double *Xd, const double *Ud, const double *Q, const double *R,
                                                     const double *Qf, double *q, double *r) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const size_t s = sel ? sel[blockIdx.y] : blockIdx.y;
    const int k = blockIdx.x;  // 0..N
    double *dx = lds, *du = dx + nX;
    const double *xk = X + (s * (N + 1) + k) * nX, *xd = Xd + (s * (N + 1) + k) * nX;
    for (int i = tid; i < nX; i += CT_T) dx[i] = xk[i] - xd[i];
    if (k < N) for (int i = tid; i < nU; i += CT_T) du[i] = U[(s * N + k) * nU + i] - Ud[(s * N + k) * nU + i];
    __syncthreads();
    const double *W = k < N ? Q : Qf;
    for (int j = tid; j < nX; j += CT_T) {
        double v = 0.0;
        for (int i = 0; i < nX; i++) v += dx[i] * W[(size_t)i * nX + j];
        q[(s * (N + 1) + k) * nX + j] = v;
    }
    if (k < N) for (int j = tid; j < nU; j += CT_T) {
        double v = 0.0;
        for (int i = 0; i < nU; i++) v += du[i] * R[(size_t)i * nU + j];
        r[(s * N + k) * nU + j] = v;
    }
}

// Armijo candidates (doptimizer.py:436-446): bX[s][m] = X[s] + lambda_m dX[s], bU likewise, for the seeds in `sel`.
// Candidate row c = blockIdx.y * M + m.
__global__ void k_candidates(int N, int nX, int nU, int M, const int *sel, const double *lambdas, const double *X,
                             const double *U, const double *dX, const double *dU, double *bX, double *bU) {
    const size_t row = blockIdx.y, s = sel ? sel[row / M] : row / M;
    const double lam = lambdas[row % M];
    const size_t nx = (size_t)(N + 1) * nX, nu = (size_t)N * nU;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nx; e += (size_t)gridDim.x * blockDim.x)
        bX[row * nx + e] = X[s * nx + e] + lam * dX[s * nx + e];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nu; e += (size_t)gridDim.x * blockDim.x)
        bU[row * nu + e] = U[s * nu + e] + lam * dU[s * nu + e];
}

// dst[dst_rows[i]] = src[src_rows[i]] for rows of `width` doubles (accepting Armijo candidates, gathering sub-batches).
__global__ void k_copy_rows(int n, size_t width, const int *dst_rows, const int *src_rows, const double *src, double *dst) {
    for (size_t i = blockIdx.y; i < (size_t)n; i += gridDim.y) {
        const size_t d = dst_rows ? dst_rows[i] : i, sr = src_rows ? src_rows[i] : i;
        for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < width; e += (size_t)gridDim.x * blockDim.x)
            dst[d * width + e] = src[sr * width + e];
    }
}

}  // namespace

namespace {
// Stream lanes of the discrete-optimisation kernels: every launch of this file goes to the calling thread's current lane --
// lane 0 is the device's default stream (what every call used before), lanes 1 and 2 are two ordinary (blocking) streams per
// device.  Work on a blocking stream orders itself against the default stream in both directions, so a caller forks by
// switching lanes (kernels on lanes 1 and 2 run side by side) and joins by going back to lane 0 -- no events to manage.
thread_local int g_lane = 0;
std::mutex g_lane_mutex;
constexpr int DOPT_LANES = 4;
std::map<int, std::array<hipStream_t, DOPT_LANES>> g_lane_streams;
hipStream_t dopt_lane_stream(int device, int lane) {
    if (lane <= 0 || lane > DOPT_LANES) return nullptr;
    std::lock_guard<std::mutex> lock(g_lane_mutex);
    auto it = g_lane_streams.find(device);
    if (it == g_lane_streams.end()) { std::array<hipStream_t, DOPT_LANES> none{}; it = g_lane_streams.emplace(device, none).first; }
    hipStream_t &st = it->second[lane - 1];
    if (!st && hipStreamCreate(&st) != hipSuccess) st = nullptr;     // (falls back to the default stream: still correct)
    return st;
}
hipStream_t dopt_stream(int device) { return dopt_lane_stream(device, g_lane); }
}  // namespace

extern "C" {

int tg_tv_lq(int32_t device, const tg_lq_problem *p) {
    if (!p || p->n_problems <= 0 || p->horizon <= 0 || p->nX <= 0 || p->nU <= 0) return fail(TG_ERR_INVALID, "bad LQ problem sizes");
    if (!p->A_dev || !p->B_dev || !p->Q_dev || !p->Qf_dev || !p->R_dev || !p->K_dev) return fail(TG_ERR_INVALID, "null LQ buffer");
    if ((p->q_dev == nullptr) != (p->r_dev == nullptr)) return fail(TG_ERR_INVALID, "q and r must be given together");
    if (p->hz_dev && (p->hz_R < p->hz_nx + p->nU || p->hz_nx > p->nX)) return fail(TG_ERR_INVALID, "bad curvature block sizes");
    if (p->nU > 64) return fail(TG_ERR_UNSUPPORTED, "more than 64 inputs");
    if (p->k_begin < 0 || p->k_end < 0 || p->k_end > p->horizon || (p->k_end > 0 && p->k_begin >= p->k_end) || (p->k_end == 0 && p->k_begin != 0))
        return fail(TG_ERR_INVALID, "bad step range of the sweep");
    if (p->k_end > 0 && p->k_end < p->horizon && !p->Pt_dev) return fail(TG_ERR_INVALID, "a sweep that ends before the horizon needs the terminal (P, b) of the steps behind it");
    if (p->ds_nd < 0 || p->ds_nk < 0 || p->ds_nu < 0 || (p->ds_nd > 0 && (2 * (p->ds_nd + p->ds_nk) != p->nX || p->ds_nu + p->ds_nk != p->nU)))
        return fail(TG_ERR_INVALID, "DSystem block structure does not match nX / nU");
    // size class: tile size TS with nX <= 16*TS (one tile per thread), prefetch registers RI*CI >= nX*ceil(nX/32)/8
    const int nX = p->nX, nXU = p->nX * p->nU;
    HIP_TRY(hipSetDevice(device));
    // matrix-core sweep (k_tv_lq_mfma) whenever it fits: nX <= 96, nU <= 32, LDS; TREPAMD_LQ_LEGACY=1 keeps the VALU kernel
    {
        const LqLayout L(p->nX, p->nU);
        const size_t lds2 = sizeof(double) * (size_t)L.total;
        const char *legacy = std::getenv("TREPAMD_LQ_LEGACY");
        if (!(legacy && legacy[0] == '1') && nX <= 96 && p->nU <= 32 && 1 + nX <= 8 * (64 - p->nU) && lds2 <= 160 * 1024 - 256) {
            // instantiated size classes: nX <= 16, 32, 48, 80, 96 (tiles per dimension 1, 2, 3, 5, 6) x nU <= 4, 8, 20, 32
            const int nt = nX <= 16 ? 1 : (nX <= 32 ? 2 : (nX <= 48 ? 3 : (nX <= 80 ? 5 : 6)));
            const int nr = p->nU <= 4 ? 4 : (p->nU <= 8 ? 8 : (p->nU <= 20 ? 20 : 32));
            // DSystem block structure: its own kernel (k_tv_lq_ds) when the padding of the dense-row blocks can be taken from the sparse rows
            const char *dense_env = std::getenv("TREPAMD_LQ_DENSE");
            if (p->ds_nd > 0 && !(dense_env && dense_env[0] == '1') && p->ds_nk >= round_up(p->ds_nd, 4) - p->ds_nd && p->ds_nk <= 31 && (nX % 2) == 0) {
                const LqDsLayout D(16 * nt, p->nU, p->ds_nd, p->ds_nd + p->ds_nk);
                const size_t ldsd = sizeof(double) * (size_t)D.total;
                if (ldsd <= 160 * 1024 - 256 && nt * (D.ldc / 16) + nt * ((p->nU + 15) / 16) <= 32) {
#define TG_LQD_LAUNCH(NT_, NR_)                                                                                                  \
                    do {                                                                                                         \
                        if (ldsd > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_tv_lq_ds<NT_, NR_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd)); \
                        hipLaunchKernelGGL((k_tv_lq_ds<NT_, NR_>), dim3(p->n_problems), dim3(LQM_T), ldsd, dopt_stream(device), *p);                 \
                    } while (0)
#define TG_LQD_NR(NT_)                                                                                                           \
                    switch (nr) {                                                                                                \
                    case 4: TG_LQD_LAUNCH(NT_, 4); break;                                                                        \
                    case 8: TG_LQD_LAUNCH(NT_, 8); break;                                                                        \
                    case 20: TG_LQD_LAUNCH(NT_, 20); break;                                                                      \
                    default: TG_LQD_LAUNCH(NT_, 32); break;                                                                      \
                    }
                    switch (nt) {
                    case 1: TG_LQD_NR(1) break;
                    case 2: TG_LQD_NR(2) break;
                    case 3: TG_LQD_NR(3) break;
                    case 5: TG_LQD_NR(5) break;
                    default: TG_LQD_NR(6) break;
                    }
#undef TG_LQD_NR
#undef TG_LQD_LAUNCH
                    HIP_TRY(hipGetLastError());
                    return TG_SUCCESS;
                }
            }
            const LqLayout L2(16 * nt, p->nU);   // the kernel pads nX to 16 * nt
            const size_t ldsk = sizeof(double) * (size_t)L2.total;
            if (ldsk <= 160 * 1024 - 256) {
#define TG_LQ_LAUNCH(NT_, NR_)                                                                                                   \
                do {                                                                                                             \
                    if (ldsk > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_tv_lq_mfma<NT_, NR_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsk)); \
                    hipLaunchKernelGGL((k_tv_lq_mfma<NT_, NR_>), dim3(p->n_problems), dim3(LQM_T), ldsk, dopt_stream(device), *p);                   \
                } while (0)
#define TG_LQ_NR(NT_)                                                                                                            \
                switch (nr) {                                                                                                    \
                case 4: TG_LQ_LAUNCH(NT_, 4); break;                                                                             \
                case 8: TG_LQ_LAUNCH(NT_, 8); break;                                                                             \
                case 20: TG_LQ_LAUNCH(NT_, 20); break;                                                                           \
                default: TG_LQ_LAUNCH(NT_, 32); break;                                                                           \
                }
                switch (nt) {
                case 1: TG_LQ_NR(1) break;
                case 2: TG_LQ_NR(2) break;
                case 3: TG_LQ_NR(3) break;
                case 5: TG_LQ_NR(5) break;
                default: TG_LQ_NR(6) break;
                }
#undef TG_LQ_NR
#undef TG_LQ_LAUNCH
                HIP_TRY(hipGetLastError());
                return TG_SUCCESS;
            }
        }
    }
    const int cls = (nX <= 32 && nXU <= 2 * LQ_T) ? 0 : ((nX <= 64 && nXU <= 6 * LQ_T) ? 1 : ((nX <= 80 && nXU <= 8 * LQ_T) ? 2 : ((nX <= 96 && nXU <= 12 * LQ_T) ? 3 : -1)));
    if (cls < 0) return fail(TG_ERR_UNSUPPORTED, "state dimension too large for the LDS-resident Riccati kernel");
    const int ts = cls == 0 ? 2 : (cls == 1 ? 4 : (cls == 2 ? 5 : 6));
    const size_t lds = lq_lds_bytes(p->nX, p->nU, ts);
    if (lds > 160 * 1024 - 64) return fail(TG_ERR_UNSUPPORTED, "state dimension too large for the LDS-resident Riccati kernel");
    const void *fn = cls == 0 ? (const void *)k_tv_lq<2, 4, 1, 2> : (cls == 1 ? (const void *)k_tv_lq<4, 8, 2, 6>
                     : (cls == 2 ? (const void *)k_tv_lq<5, 10, 3, 8> : (const void *)k_tv_lq<6, 12, 3, 12>));
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    switch (cls) {
    case 0: hipLaunchKernelGGL((k_tv_lq<2, 4, 1, 2>), dim3(p->n_problems), dim3(LQ_T), lds, dopt_stream(device), *p); break;
    case 1: hipLaunchKernelGGL((k_tv_lq<4, 8, 2, 6>), dim3(p->n_problems), dim3(LQ_T), lds, dopt_stream(device), *p); break;
    case 2: hipLaunchKernelGGL((k_tv_lq<5, 10, 3, 8>), dim3(p->n_problems), dim3(LQ_T), lds, dopt_stream(device), *p); break;
    default: hipLaunchKernelGGL((k_tv_lq<6, 12, 3, 12>), dim3(p->n_problems), dim3(LQ_T), lds, dopt_stream(device), *p); break;
    }
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

int tg_adjoint_sweep(int32_t device, int32_t n_problems, int32_t horizon, int32_t nX, int32_t nU, const int32_t *select_dev,
                     const double *A_dev, const double *B_dev, const double *K_dev, const double *q_dev, const double *r_dev,
                     double *Z_dev) {
    if (n_problems <= 0 || horizon <= 0 || !A_dev || !B_dev || !K_dev || !q_dev || !r_dev || !Z_dev) return fail(TG_ERR_INVALID, "bad arguments");
    const size_t lds = sizeof(double) * ((size_t)nX * (nX | 1) + (size_t)nX * nU + (size_t)nU * nX + 2 * nX + nU);
    if (lds > 160 * 1024 - 64 || nX > 96 || nX * nU > 12 * SW_T) return fail(TG_ERR_UNSUPPORTED, "state dimension too large");
    HIP_TRY(hipSetDevice(device));
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_adjoint, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_adjoint, dim3(n_problems), dim3(SW_T), lds, dopt_stream(device), horizon, nX, nU, select_dev, A_dev, B_dev, K_dev, q_dev, r_dev, Z_dev);
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

int tg_tangent_rollout(int32_t device, int32_t n_problems, int32_t horizon, int32_t nX, int32_t nU, const int32_t *select_dev,
                       const double *A_dev, const double *B_dev, const double *K_dev, const double *C_dev, const double *q_dev,
                       const double *r_dev, double *dX_dev, double *dU_dev, double *dcost_dev) {
    if (n_problems <= 0 || horizon <= 0 || !A_dev || !B_dev || !K_dev || !C_dev || !q_dev || !r_dev || !dX_dev || !dU_dev || !dcost_dev)
        return fail(TG_ERR_INVALID, "bad arguments");
    if (nX <= 4 * TR_MAX_CA && nU <= 4 * TR_MAX_CB && nX <= 8 * TR_MAX_CK && nX >= 1 && nU >= 1 && !std::getenv("TREPAMD_TANGENT_LDS")) {     // rows in registers (k_tangent_rows)
        int nt = 4 * nX > 8 * nU ? 4 * nX : 8 * nU;
        nt = (nt + 63) & ~63;
        if (nt <= 384) {
            HIP_TRY(hipSetDevice(device));
            // even sizes: pairs of columns per thread, 16-byte loads (rows of A, B, K then start on 16-byte boundaries)
            const bool pair = nX % 2 == 0 && nU % 2 == 0 && (((uintptr_t)A_dev | (uintptr_t)B_dev | (uintptr_t)K_dev) & 15) == 0 && !std::getenv("TREPAMD_TANGENT_NO_PAIRS");
            const int nca = pair ? 2 * ((nX + 7) / 8) : (nX + 3) / 4, ncb = pair ? 2 * ((nU + 7) / 8) : (nU + 3) / 4, nck = pair ? 2 * ((nX + 15) / 16) : (nX + 7) / 8;
            const size_t ldsr = sizeof(double) * (2 * (size_t)(8 * nck) + 4 * ncb + nt);
#define LAUNCH_TR(CA_, CB_, CK_, PAIR_)                                                                                                              \
            hipLaunchKernelGGL((k_tangent_rows<CA_, CB_, CK_, PAIR_>), dim3(n_problems), dim3(nt), ldsr, dopt_stream(device), horizon, nX, nU, select_dev, \
                               A_dev, B_dev, K_dev, C_dev, q_dev, r_dev, dX_dev, dU_dev, dcost_dev)
            if (pair) {
                if (nca <= 8 && ncb <= 4 && nck <= 4) LAUNCH_TR(8, 4, 4, true);
                else if (nca <= 20 && ncb <= 6 && nck <= 10) LAUNCH_TR(20, 6, 10, true);       // the puppet's nX = 80, nU = 18
                else LAUNCH_TR(24, 8, 12, true);
            }
            else if (nca <= 8 && ncb <= 4 && nck <= 4) LAUNCH_TR(8, 4, 4, false);
            else if (nca <= 20 && ncb <= 6 && nck <= 10) LAUNCH_TR(20, 6, 10, false);
            else LAUNCH_TR(24, 8, 12, false);
#undef LAUNCH_TR
            HIP_TRY(hipGetLastError());
            return TG_SUCCESS;
        }
    }
    const int lda = nX | 1;
    const size_t lds = sizeof(double) * ((size_t)nX * lda + (size_t)nX * nU + (size_t)nU * lda + 2 * nX + nU + SW_T);
    if (lds > 160 * 1024 - 64 || nX > 96 || nX * nU > 12 * SW_T) return fail(TG_ERR_UNSUPPORTED, "state dimension too large");
    HIP_TRY(hipSetDevice(device));
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_tangent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_tangent, dim3(n_problems), dim3(SW_T), lds, dopt_stream(device), horizon, nX, nU, select_dev, A_dev, B_dev, K_dev, C_dev, q_dev,
                       r_dev, dX_dev, dU_dev, dcost_dev);
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

int tg_quadratic_cost(int32_t device, int32_t n_trajectories, int32_t group, const int32_t *select_dev, int32_t horizon,
                      int32_t nX, int32_t nU, const double *X_dev, const double *U_dev, const double *Xd_dev, const double *Ud_dev, const double *Q_dev,
                      const double *R_dev, const double *Qf_dev, double *cost_dev) {
    if (n_trajectories <= 0 || group <= 0 || horizon <= 0 || !X_dev || !U_dev || !Xd_dev || !Ud_dev || !Q_dev || !R_dev || !Qf_dev || !cost_dev)
        return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    if (nX <= 96 && nU <= 96 && !std::getenv("TREPAMD_COST_LEGACY")) {     // matrix-core version (k_cost_mfma)
        const int nt = (nX + 15) / 16, ldx = 16 * nt, ldu = round_up(nU, 16);
        const size_t ldsm = sizeof(double) * ((size_t)ldx * ldx + (size_t)ldu * ldu + CT_T);
#define LAUNCH_COST(NT_)                                                                                                                  \
        case NT_:                                                                                                                         \
            if (ldsm > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_cost_mfma<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm)); \
            hipLaunchKernelGGL((k_cost_mfma<NT_>), dim3(n_trajectories), dim3(CT_T), ldsm, dopt_stream(device), horizon, nX, nU, group, select_dev, X_dev, U_dev,       \
                               Xd_dev, Ud_dev, Q_dev, R_dev, Qf_dev, cost_dev);                                                           \
            break;
        switch (nt) { LAUNCH_COST(1) LAUNCH_COST(2) LAUNCH_COST(3) LAUNCH_COST(4) LAUNCH_COST(5) LAUNCH_COST(6) default: break; }
#undef LAUNCH_COST
        HIP_TRY(hipGetLastError());
        return TG_SUCCESS;
    }
    const size_t lds = sizeof(double) * (2 * (size_t)nX * nX + (size_t)nU * nU + 4 * (size_t)(nX + nU) + CT_T);
    if (lds > 160 * 1024 - 64) return fail(TG_ERR_UNSUPPORTED, "state dimension too large");
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_cost, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_cost, dim3(n_trajectories), dim3(CT_T), lds, dopt_stream(device), horizon, nX, nU, group, select_dev, X_dev, U_dev, Xd_dev, Ud_dev, Q_dev, R_dev,
                       Qf_dev, cost_dev);
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

int tg_quadratic_cost_gradients(int32_t device, int32_t n_problems, int32_t horizon, int32_t nX, int32_t nU, const int32_t *select_dev,
                                const double *X_dev, const double *U_dev, const double *Xd_dev, const double *Ud_dev,
                                const double *Q_dev, const double *R_dev, const double *Qf_dev, double *q_dev, double *r_dev) {
    if (n_problems <= 0 || horizon <= 0 || !X_dev || !U_dev || !Xd_dev || !Ud_dev || !Q_dev || !R_dev || !Qf_dev || !q_dev || !r_dev)
        return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_cost_grad, dim3(horizon + 1, n_problems), dim3(CT_T), sizeof(double) * (nX + nU), dopt_stream(device), horizon, nX, nU, select_dev,
                       X_dev, U_dev, Xd_dev, Ud_dev, Q_dev, R_dev, Qf_dev, q_dev, r_dev);
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

int tg_armijo_candidates(int32_t device, int32_t n_problems, int32_t n_lambdas, int32_t horizon, int32_t nX, int32_t nU,
                         const int32_t *select_dev, const double *lambdas_dev, const double *X_dev, const double *U_dev,
                         const double *dX_dev, const double *dU_dev, double *bX_dev, double *bU_dev) {
    if (n_problems <= 0 || n_lambdas <= 0 || horizon <= 0 || !lambdas_dev || !X_dev || !U_dev || !dX_dev || !dU_dev || !bX_dev || !bU_dev)
        return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    const int bx = (int)(((size_t)(horizon + 1) * nX + 255) / 256);
    hipLaunchKernelGGL(k_candidates, dim3(bx > 64 ? 64 : bx, n_problems * n_lambdas), dim3(256), 0, dopt_stream(device), horizon, nX, nU, n_lambdas, select_dev,
                       lambdas_dev, X_dev, U_dev, dX_dev, dU_dev, bX_dev, bU_dev);
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

int tg_copy_rows(int32_t device, int32_t n_rows, uint64_t row_doubles, const int32_t *dst_rows_dev, const int32_t *src_rows_dev,
                 const double *src_dev, double *dst_dev) {
    if (n_rows <= 0 || !src_dev || !dst_dev) return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    const int bx = (int)((row_doubles + 255) / 256);
    hipLaunchKernelGGL(k_copy_rows, dim3(bx > 64 ? 64 : (bx < 1 ? 1 : bx), n_rows > 65535 ? 65535 : n_rows), dim3(256), 0, dopt_stream(device), n_rows, (size_t)row_doubles, dst_rows_dev,
                       src_rows_dev, src_dev, dst_dev);
    HIP_TRY(hipGetLastError());
    return TG_SUCCESS;
}

#if defined(TG_PROFILE)
int tg_lq_profile(int32_t device, int64_t out[8]) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lq_prof), 8 * sizeof(long long)));
    return TG_SUCCESS;
}
#endif

int tg_device_synchronize(int32_t device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return TG_SUCCESS;
}

int tg_dopt_use_stream(int32_t device, int32_t lane) {
    (void)device;
    if (lane < 0 || lane > DOPT_LANES) return fail(TG_ERR_INVALID, "stream lane must be 0 (default stream) .. 4");
    g_lane = lane;
    return TG_SUCCESS;
}

/* The HIP stream of lane 1 .. 4 (hipStream_t as void *; created on first use), e.g. for tg_batch_set_stream: a batch's kernels then
 * run in that lane. */
void *tg_dopt_lane_stream(int32_t device, int32_t lane) {
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    return (void *)dopt_lane_stream(device, lane);
}

/* Order lane `waiter` after everything enqueued so far in lane `signal` (an event; no host synchronisation).  Lanes 1 .. 4. */
int tg_dopt_lane_wait(int32_t device, int32_t waiter, int32_t signal) {
    if (waiter < 1 || waiter > DOPT_LANES || signal < 1 || signal > DOPT_LANES || waiter == signal) return fail(TG_ERR_INVALID, "bad stream lanes");
    HIP_TRY(hipSetDevice(device));
    hipStream_t sw = dopt_lane_stream(device, waiter), ss = dopt_lane_stream(device, signal);
    if (!sw || !ss) return fail(TG_ERR_HIP, "stream lane not available");
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ss);
    if (e == hipSuccess) e = hipStreamWaitEvent(sw, ev, 0);
    hipEventDestroy(ev);          // (released once the recorded work has completed)
    if (e != hipSuccess) return fail(TG_ERR_HIP, "event between stream lanes failed");
    return TG_SUCCESS;
}
}  // extern "C"
