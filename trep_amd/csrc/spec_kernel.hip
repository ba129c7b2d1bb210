// spec_kernel.hip -- the rollout kernel of ONE system with its schedule compiled in (trep_amd/specialize.py).
//
// The generic kernel k_run<TEAM, MODE_ROLLOUT> reads ~110 integers (sizes, counts, LDS offsets) and ~65 table pointers
// of the schedule at run time; they do not fit the scalar register file, so the compiler spills them into VGPR lanes and
// every use pays a v_readlane plus the address arithmetic around it.  Here the same source (mvi_core.hpp) is
// instantiated on `SpecProg` (generated header, -DTG_SPEC_HEADER=...): integers are immediates (LDS offsets fold into
// the ds_* offset fields, loop bounds are known), tables are constant arrays in the code object.  Nothing about the
// arithmetic changes -- it is the same template -- so results equal the generic kernel's up to the compiler's FMA
// contraction choices (<= 1e-12 relative, identical Newton iteration counts; tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <type_traits>

#define TG_GJ_PANEL_DEFAULT 1      // 17..31 unknowns, full-wave team: the Newton systems go through gj_panel (mvi_core.hpp)
#include "mvi_core.hpp"
#include TG_SPEC_HEADER

namespace {
// SPEC_ARGS_IN_MEMORY: the launch arguments are read from device memory through a constant-address-space reference and
// re-read at the head of every step, instead of ~50 scalar registers held (and spilled) for the whole rollout
#if defined(SPEC_ARGS_IN_MEMORY)
#define SPEC_KERNEL_ARGS const tg::RunArgs *__restrict__ Ag
#define SPEC_ARGS_REF tg::KArgs &A = *(tg::KArgs *)Ag
#else
#define SPEC_KERNEL_ARGS const tg::RunArgs A
#define SPEC_ARGS_REF ((void)0)
#endif

// PIVOT (rollouts): 0 single-precision pivot ranking, 1 the reference's exact pivot rule -- two kernels, so that neither carries
// the other's solver (registers, callee-saved spills) in its call graph
// wavefronts per workgroup: the derivative kernels of a full-wave team run with helper waves (mvi_core.hpp, TG_HELPER_WAVES)
template <int MODE> constexpr int spec_waves() { return ((MODE == tg::MODE_DERIV1 || MODE == tg::MODE_DERIV2Z) && SPEC_TEAM == 64) ? TG_NW : 1; }
static_assert(TG_NW >= 1 && TG_NW <= 2, "helper waves: the pair lists of program.hpp are split in exactly two parts (wave_part, wp_* / wt_* / wcp4)");
static_assert(TG_NW == 1 || SPEC_TEAM == 64, "helper waves are for full-wave teams (trep_amd/specialize.py passes TG_HELPER_WAVES only then)");

// waves per SIMD the rollout kernel is compiled for (its register budget: 2 -> 256, 3 -> 168): -DTG_ROLLOUT_WAVES=3 is an occupancy
// experiment (tools/ab_spec.sh), not a product setting -- the LDS slice of a trajectory allows 8 per CU = 2 per SIMD
#ifndef TG_DERIV_WAVES
#define TG_DERIV_WAVES 1      // wavefronts per SIMD the second-derivative kernel's register allocation leaves room for (its LDS slice allows one; the first-
                              // derivative kernel: two when its compact slice lets three workgroups of two waves share a CU)
#endif
#ifndef TG_ROLLOUT_WAVES
#define TG_ROLLOUT_WAVES 2
#endif
template <int MODE, int PIVOT = 0>
__global__ __launch_bounds__(64 * spec_waves<MODE>(), MODE == tg::MODE_DERIV1 ? (SpecProg::a_ok ? 2 : 1) : (MODE == tg::MODE_DERIV2Z ? TG_DERIV_WAVES : TG_ROLLOUT_WAVES)) void k_spec(SPEC_KERNEL_ARGS) {
    SPEC_ARGS_REF;
double *lds = tg_lds_base();
    const SpecProg P{};
    int wave = 0, team = threadIdx.x / SPEC_TEAM, lane = threadIdx.x % SPEC_TEAM;
    if constexpr (spec_waves<MODE>() > 1) {     // (only then: masking the thread index changes the one-wave kernels' register allocation)
        wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        team = 0; lane = threadIdx.x & 63;
    }
    const int block = MODE == tg::MODE_ROLLOUT ? tg_xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int traj = tg::tg_remap_trajectory(A, block * (64 / SPEC_TEAM) + team);
    constexpr int stride = MODE == tg::MODE_DERIV2Z ? SpecProg::e_lds_per_team : (MODE == tg::MODE_DERIV1 ? SpecProg::a_lds_per_team : SpecProg::lds_per_team);
    tg::run_trajectory<SPEC_TEAM, MODE, SPEC_SPRINGS, const SpecProg, std::remove_reference<decltype(A)>::type, PIVOT>(P, A, lds + (size_t)team * stride, lane, traj, wave, spec_waves<MODE>());
}

// Test hook: the Newton-system solve of this library's rollout kernel (default pivot rule) on caller-supplied matrices [nf][nf + 1],
// one workgroup per matrix, in the rollout kernel's own LDS layout: the structured solve along the compiled-in plan if the system
// has one (bbd.hpp), the pivoting solver if a pivot guard fails.  path: 1 structured, 2 pivoting solver, -1 singular.
__global__ __launch_bounds__(64, 2) void k_spec_debug_solve(const double *A_in, double *x_out, int *path_out, int skip_structured) {
#if defined(__HIP_DEVICE_COMPILE__)
    double *S = tg_lds_base();
    const SpecProg P{};
    constexpr int nf = SpecProg::nf, ld = SpecProg::df_ld, nb4 = (nf + 3) >> 2;
    const int lane = threadIdx.x;
    const double *src = A_in + (size_t)blockIdx.x * nf * (nf + 1);
    constexpr bool PKI = tg::tg_static_pk<SpecProg>::value;     // the rollout kernel solves the image in the plan's own order (bbd.hpp, BbdPacked)
    auto load_dense = [&]() {
        for (int e = lane; e < nf * ld; e += 64) S[P.o_Df + e] = 0.0;
        __syncthreads();
        for (int e = lane; e < nf * (nf + 1); e += 64) S[P.o_Df + (e / (nf + 1)) * ld + e % (nf + 1)] = src[e];
    };
    int path = 0;
    bool ok = false;
    if constexpr (SPEC_TEAM == 64 && SpecProg::bbd_ok != 0) {
        int *tab = (int *)(S + P.o_bbd);
        for (int e = lane; e < 128; e += 64) tab[e] = P.bbd_tab[e];
        if constexpr (PKI) {     // the caller's dense matrix scattered to the packed places (entries without a place must be structural zeros)
            for (int e = lane; e < SpecProg::bbd_pk_size; e += 64) S[P.o_Df + e] = 0.0;
            __syncthreads();
            for (int e = lane; e < nf * (nf + 1); e += 64) { const int at = P.bbd_map[e]; if (at >= 0) S[P.o_Df + at] = src[e]; }
            for (int e = lane; e < SpecProg::bbd_pk_nones; e += 64) S[P.o_Df + P.bbd_ones[e]] = 1.0;
        } else load_dense();
        __syncthreads();
        typedef typename std::conditional<PKI, tg::BbdPackedImage<SpecProg::bbd_pk_nr, SpecProg::bbd_pk_nc2, SpecProg::bbd_pk_tb, SpecProg::bbd_pk_tc2, SpecProg::bbd_pk_xs>, tg::BbdDenseImage>::type Img;
        if (!skip_structured && tg::gj_bbd<nf, ld, SpecProg::bbd_ng, SpecProg::bbd_nb, SpecProg::bbd_t, tg::BbdNoUpdate, Img>(S + P.o_Df, tg::bbd_rows<SpecProg::bbd_ng + SpecProg::bbd_nb>(tab, lane), PKI ? S + P.o_W + 12 * P.n_joints : S + P.o_J, lane, P.bbd_tvar)) { ok = true; path = 1; }
        __syncthreads();
        if (PKI && ok && lane < nf) x_out[(size_t)blockIdx.x * nf + lane] = S[P.o_Df + SpecProg::bbd_pk_xs + lane];
        if (PKI && !ok) load_dense();
    } else load_dense();
    __syncthreads();
    if (!ok) {
        if constexpr (SPEC_TEAM == 64 && nb4 >= 5 && nf <= 31 && 12 * SpecProg::n_items >= 128) ok = tg::Core<64>::gj_panel<4 * nb4>(true, S + P.o_Df, nf, ld, lane, S + P.o_J);
        else if constexpr (SPEC_TEAM == 64 && nb4 <= 8) ok = tg::Core<64>::gj_rows<4 * nb4>(true, S + P.o_Df, nf, ld, lane);
        path = ok ? 2 : -1;
    }
    __syncthreads();
    if (lane < nf && !(PKI && path == 1)) x_out[(size_t)blockIdx.x * nf + lane] = S[P.o_Df + lane * ld + nf];
    if (lane == 0) path_out[blockIdx.x] = path;
#endif
}

template <int MODE, int PIVOT = 0>
int launch_mode(const tg::RunArgs *A, tg::RunArgs *slot, int grid, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spec<MODE, PIVOT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
#if defined(SPEC_ARGS_IN_MEMORY)
    // `slot` is one of the batch's device-side argument blocks and `A` lives in the batch's pinned host ring (trepamd.hip,
    // launch()): an asynchronous copy ordered before the kernel on the batch's stream; the caller reuses a (slot, ring entry)
    // pair only after the event recorded behind this launch has completed
    if (!slot) return 3;
    if (hipMemcpyAsync(slot, A, sizeof(tg::RunArgs), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    hipLaunchKernelGGL((k_spec<MODE, PIVOT>), dim3(grid), dim3(64 * spec_waves<MODE>()), lds, stream, (const tg::RunArgs *)slot);
#else
    hipLaunchKernelGGL((k_spec<MODE, PIVOT>), dim3(grid), dim3(64 * spec_waves<MODE>()), lds, stream, *A);
#endif
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace

extern "C" {
const int *tg_spec_sizes(void) {
#if defined(TG_MOCK_REAL_LDS)      // timing mock with aliased LDS areas (tools/mock_third_wave.py): answer with the real schedule's slice
    constexpr int lds_doubles = TG_MOCK_REAL_LDS;
#else
    constexpr int lds_doubles = SpecProg::lds_per_team;
#endif
    static const int s[8] = {(int)sizeof(tg::DevProg), (int)sizeof(tg::RunArgs), SpecProg::nq, SpecProg::nd, SpecProg::nc, SpecProg::n_items,
                             SpecProg::n_pairs, lds_doubles};
    return s;
}
// hash of the generated header this library was compiled against (tg_system_spec_key; trep_amd/specialize.py passes it)
unsigned long long tg_spec_key(void) {
#if defined(TG_SPEC_KEY)
    return TG_SPEC_KEY;
#else
    return 0ull;
#endif
}
// wavefronts per trajectory in the derivative kernels of this library (helper waves, mvi_core.hpp)
int tg_spec_waves(void) { return spec_waves<tg::MODE_DERIV2Z>(); }
// bit m set: kernel mode m (tg::MODE_*) has a specialised instantiation in this library
int tg_spec_modes(void) {
    int m = 1 << tg::MODE_ROLLOUT;
#if defined(SPEC_DERIVATIVES)
    m |= (1 << tg::MODE_DERIV1) | (1 << tg::MODE_DERIV2Z);
#endif
    return m;
}
// (test hook) n_mats systems [nf][nf + 1] in, solutions [nf] and the path taken out; device pointers
int tg_spec_debug_solve(const double *A_dev, double *x_dev, int *path_dev, int n_mats, int skip_structured) {
    const size_t lds = sizeof(double) * (size_t)SpecProg::lds_per_team;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spec_debug_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_spec_debug_solve, dim3(n_mats), dim3(64), lds, 0, A_dev, x_dev, path_dev, skip_structured);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int tg_spec_launch(int mode, const tg::RunArgs *A, tg::RunArgs *device_slot, int grid, size_t lds, void *stream) {
#if defined(TG_MOCK_TIMING)     // the mock's header aliases LDS areas: its slice is smaller than the one the host computed from the real schedule
    if (mode == tg::MODE_ROLLOUT) lds = sizeof(double) * (size_t)SpecProg::lds_per_team * (64 / SPEC_TEAM);
#endif
#if defined(TG_MOCK_LDS_D1)   // timing mock: another LDS size (another number of resident workgroups) for the first-derivative kernel; wrong numbers
    if (mode == tg::MODE_DERIV1) lds = TG_MOCK_LDS_D1;
#endif
#if defined(TG_MOCK_LDS_D2)
    if (mode == tg::MODE_DERIV2Z) lds = TG_MOCK_LDS_D2;
#endif
    switch (mode) {
    case tg::MODE_ROLLOUT:
        return A->exact_pivot ? launch_mode<tg::MODE_ROLLOUT, 1>(A, device_slot, grid, lds, (hipStream_t)stream)
                              : launch_mode<tg::MODE_ROLLOUT, 0>(A, device_slot, grid, lds, (hipStream_t)stream);
#if defined(SPEC_DERIVATIVES)
    case tg::MODE_DERIV1: return launch_mode<tg::MODE_DERIV1>(A, device_slot, grid, lds, (hipStream_t)stream);
    case tg::MODE_DERIV2Z: return launch_mode<tg::MODE_DERIV2Z>(A, device_slot, grid, lds, (hipStream_t)stream);
#endif
    default: return 2;
    }
}
}
