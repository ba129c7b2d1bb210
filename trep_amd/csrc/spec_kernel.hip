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
static_assert(TG_NW == 1 || SPEC_TEAM == 64, "helper waves are for full-wave teams (trep_amd/specialize.py passes TG_HELPER_WAVES only then)");

template <int MODE, int PIVOT = 0>
__global__ __launch_bounds__(64 * spec_waves<MODE>(), (MODE == tg::MODE_DERIV1 || MODE == tg::MODE_DERIV2Z) ? 1 : 2) void k_spec(SPEC_KERNEL_ARGS) {
    SPEC_ARGS_REF;
double *lds = tg_lds_base();
    const SpecProg P{};
    int wave = 0, team = threadIdx.x / SPEC_TEAM, lane = threadIdx.x % SPEC_TEAM;
    if constexpr (spec_waves<MODE>() > 1) {     // (only then: masking the thread index changes the one-wave kernels' register allocation)
        wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        team = 0; lane = threadIdx.x & 63;
    }
    const int block = MODE == tg::MODE_ROLLOUT ? tg_xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int traj = block * (64 / SPEC_TEAM) + team;
    constexpr int stride = MODE == tg::MODE_DERIV2Z ? SpecProg::e_lds_per_team : (MODE == tg::MODE_DERIV1 ? SpecProg::d_lds_per_team : SpecProg::lds_per_team);
    tg::run_trajectory<SPEC_TEAM, MODE, SPEC_SPRINGS, const SpecProg, std::remove_reference<decltype(A)>::type, PIVOT>(P, A, lds + (size_t)team * stride, lane, traj, wave, spec_waves<MODE>());
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
    static const int s[8] = {(int)sizeof(tg::DevProg), (int)sizeof(tg::RunArgs), SpecProg::nq, SpecProg::nd, SpecProg::nc, SpecProg::n_items,
                             SpecProg::n_pairs, SpecProg::lds_per_team};
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
int tg_spec_launch(int mode, const tg::RunArgs *A, tg::RunArgs *device_slot, int grid, size_t lds, void *stream) {
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
