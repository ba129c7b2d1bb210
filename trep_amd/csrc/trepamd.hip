// trepamd.hip -- libtrepamd.so: gfx950 kernels + the C ABI of include/trep_amd.h.
//
// Launch geometry: one 64-thread workgroup (exactly one CDNA4 wavefront) holds 64/TEAM teams, one
// trajectory per team; the grid is ceil(batch / teams-per-block) workgroups, i.e. >> 256 CUs for
// the benchmark batches.  Single-wave workgroups make every __syncthreads() a wave-local
// s_waitcnt (no cross-wave barrier) and let the LDS slice of a trajectory be private to its wave.
// Trajectories are independent, so there is no inter-workgroup communication at all.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mvi_core.hpp"
#include "spec_emit.inc"
#include <dlfcn.h>

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(TG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The derivative modes keep 60-80 KB of LDS per trajectory, i.e. at most two wavefronts per CU: they may use the whole
// register file of a SIMD (no spills, deeper unrolling); the rollout modes run two wavefronts per SIMD.
template <int TEAM, int MODE, bool SPRINGS>
__global__ __launch_bounds__(64, (MODE == tg::MODE_DERIV1 || MODE == tg::MODE_DERIV2Z || MODE == tg::MODE_DYN_DERIV1) ? 1 : 2) void k_run(const tg::DevProg *__restrict__ Pg, const tg::RunArgs A) {
    double *lds = tg_lds_base();
    // The schedule sits in device memory and is read through a CONSTANT-address-space reference (mvi_core.hpp, CProg):
    // every field access is a scalar load that a phase issues when it needs it, instead of ~150 kernel-argument values
    // that the compiler would hoist, keep alive for the whole rollout and spill into VGPR lanes.
    tg::CProg &P = *(tg::CProg *)Pg;
    const int team = threadIdx.x / TEAM, lane = threadIdx.x % TEAM;
    const int block = MODE == tg::MODE_ROLLOUT ? tg_xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int traj = tg::tg_remap_trajectory(A, block * (64 / TEAM) + team);
    const int stride = MODE == tg::MODE_DERIV2Z ? P.e_lds_per_team : (MODE == tg::MODE_DERIV1 ? P.a_lds_per_team : (MODE == tg::MODE_DYN_DERIV1 ? P.g_lds_per_team : P.lds_per_team));
    tg::run_trajectory<TEAM, MODE, SPRINGS>(P, A, lds + (size_t)team * stride, lane, traj);
}

// Forward-mode kernels of the continuous dynamics (mvi_core.hpp, run_forward; dual.hpp): one wavefront per trajectory, the trajectory's
// LDS slice in units of Real.  Not a hot path (the reference's calc_dynamics_deriv2 is O(nq^4) per state): full-wave teams only.
template <int MODE, bool SPRINGS, class Real>
__global__ __launch_bounds__(64, 1) void k_forward(const tg::DevProg *__restrict__ Pg, const tg::RunArgs A) {
    Real *lds = (Real *)tg_lds_base();
    tg::CProg &P = *(tg::CProg *)Pg;
    tg::run_forward<64, MODE, SPRINGS>(P, A, lds, (int)threadIdx.x, (int)blockIdx.x);
}

}  // namespace

struct tg_system {
    tg::HostProgram H;
    int team = 64;
};

struct tg_batch {
    tg_system *sys = nullptr;
    int batch = 0, device = 0;
    tg::DevProg P{};           // device-pointer view
    tg::DevProg *d_prog = nullptr;   // the same view in device memory: what the kernels read (constant address space)
    // optional system-specialised rollout kernel (tg_batch_load_specialized): launcher exported by a generated library
    void *spec_lib = nullptr;
    int (*spec_launch)(int, const tg::RunArgs *, tg::RunArgs *, int, size_t, void *) = nullptr;
    // argument blocks of the specialised kernels: ARG_SLOTS device-side blocks fed from a pinned host ring (a truly asynchronous
    // hipMemcpyAsync; a slot is reused only after the launch that read it has finished: arg_done[i])
    static constexpr int ARG_SLOTS = 4;
    tg::RunArgs *d_args = nullptr;   // [ARG_SLOTS] device
    tg::RunArgs *h_args = nullptr;   // [ARG_SLOTS] pinned host
    hipEvent_t arg_done[ARG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool arg_used[ARG_SLOTS] = {false, false, false, false};
    int arg_next = 0;
    // small batches (the B = 1 drop-in path): tg_batch_step stages its inputs through ONE pinned block, and a pack kernel + ONE
    // copy bring (q2, p2, lambda1, iterations, status) back into a pinned host mirror that later tg_batch_get / tg_batch_status
    // calls answer from, until anything else touches the batch (launch(), tg_batch_set, restore ...)
    double *io_host = nullptr, *io_dev = nullptr;
    size_t io_in = 0, io_out = 0;      // doubles in the input / output part
    bool mirror_valid = false;
    int spec_modes = 0, spec_waves = 1;
    unsigned int spec_launched_modes = 0, generic_launched_modes = 0;   // bit m: a kernel of mode m went through that path (tg_batch_info)
    long long spec_launches = 0, generic_launches = 0;
    std::string spec_path;
    int *d_ints = nullptr;
    double *d_dbls = nullptr;
    double *q1 = nullptr, *q2 = nullptr, *p1 = nullptr, *p2 = nullptr, *lam = nullptr, *u1 = nullptr;
    double *stage_u = nullptr, *stage_k = nullptr, *stage_qh = nullptr, *stage_lh = nullptr, *f_out = nullptr;
    int *iters = nullptr, *status = nullptr, *fallbacks = nullptr;
    double *z_dev = nullptr, *hz_dev = nullptr, *zl_dev = nullptr;
    double *dyn = nullptr;     // staging of the host-facing continuous-dynamics call: q, dq, u, ddq_k, ddq, lambda
    double *dyn_d1 = nullptr;  // ... and of its eight first-derivative arrays
    double *energy = nullptr;  // [batch][2] output of tg_batch_energy
    double *lag = nullptr;     // outputs of tg_batch_lagrangian
    int *dyn_ints = nullptr;   // its status / iteration words (the integrator's own stay untouched)
    int *seeds = nullptr;      // [2][batch] direction variables of the forward-mode calls
    double *d1[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool have_d1 = false;
    long long *prof = nullptr; // diagnostic build only (TG_PROFILE)
    double *snap = nullptr;    // snapshot of (q1,q2,p1,p2,lam,u1)
    double snap_t1 = 0.0, snap_t2 = 0.0;
    long long total_iters = 0;
    double t1 = 0.0, t2 = 0.0, tolerance = 1.0e-10;
    int predictor = 0;
    double *dt_dev = nullptr;  // optional non-uniform time base (tg_batch_set_step_sizes)
    std::vector<double> dt_host;
    int dt_by_trajectory = 0;
    int exact_pivot = 0;       // 1: Newton systems solved with the reference's exact pivot rule (gj_rows_exact)
    hipStream_t stream = nullptr;
    bool own_stream = true;
    static constexpr size_t TIMING_CAP = 4096;
    bool timing = false;       // HIP-event timing of the launches: off until tg_batch_timing is called once
    double folded_ms = 0.0;    // launches recycled past TIMING_CAP
    long long folded_n = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> pool;
};

namespace {

int widths(const tg_batch *b, int field) {
    const tg::DevProg &P = b->P;
    switch (field) {
    case TG_F_Q1: case TG_F_Q2: return P.nq;
    case TG_F_P1: case TG_F_P2: return P.nd;
    case TG_F_U1: return P.nu;
    case TG_F_LAMBDA1: return P.nc;
    default: break;
    }
    if (field >= TG_F_Q2_DQ1 && field <= TG_F_L1_DK2) {
        const int k = field - TG_F_Q2_DQ1, var = k % 4, out = k / 4;
        const int rows = var == 0 ? P.nq : (var == 1 ? P.nd : (var == 2 ? P.nu : P.nk));
        return rows * (out == 2 ? P.nc : P.nd);
    }
    return -1;
}
double *field_ptr(tg_batch *b, int field) {
    switch (field) {
    case TG_F_Q1: return b->q1;
    case TG_F_Q2: return b->q2;
    case TG_F_P1: return b->p1;
    case TG_F_P2: return b->p2;
    case TG_F_U1: return b->u1;
    case TG_F_LAMBDA1: return b->lam;
    default: break;
    }
    if (field >= TG_F_Q2_DQ1 && field <= TG_F_L1_DK2) return b->d1[field - TG_F_Q2_DQ1];
    return nullptr;
}

int pick_team(const tg::HostProgram &H) {
    if (const char *env = std::getenv("TREPAMD_TEAM")) {
        int t = std::atoi(env);
        if (t == 1 || t == 4 || t == 16 || t == 64) return t;
    }
    const tg::DevProg &P = H.p;
    int width = std::max(P.n_items, (P.nf * (P.nf + 1)) / 4);
    int team = width > 32 ? 64 : (width > 8 ? 16 : (width > 2 ? 4 : 1));
    // the block's LDS (all teams) must fit the 64 KiB a workgroup may use without opting in
    while (team < 64 && (size_t)(64 / team) * P.lds_per_team * sizeof(double) > 64 * 1024) team *= 4;
    return team;
}

template <typename T>
void append(std::vector<T> &pool, const std::vector<T> &v, size_t &off) {
    off = pool.size();
    pool.insert(pool.end(), v.begin(), v.end());
    while (pool.size() % 2) pool.push_back(T());
}

template <int TEAM, int MODE, bool SPRINGS>
int launch_variant(tg_batch *b, const tg::RunArgs &A, int grid, size_t lds) {
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_run<TEAM, MODE, SPRINGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_run<TEAM, MODE, SPRINGS>), dim3(grid), dim3(64), lds, b->stream, b->d_prog, A);
    return TG_SUCCESS;
}

// systems with spring potentials run their own instantiation of every kernel (mvi_core.hpp, Core<TEAM, SPRINGS>)
template <int TEAM, int MODE>
int launch_one(tg_batch *b, const tg::RunArgs &A, int grid, size_t lds) {
#if defined(TG_PROFILE)   // the diagnostic build only instruments the plain kernels of full-wave teams (fewer instantiations)
    if (TEAM != 64 || b->P.has_cs || b->P.n_springs || b->P.has_plane || b->P.n_wrenches)
        return fail(TG_ERR_UNSUPPORTED, "the profiling build covers full-wave teams without spring / plane / wrench features");
    return launch_variant<64, MODE, false>(b, A, grid, lds);
#else
    return (b->P.has_cs || b->P.n_springs || b->P.has_plane || b->P.n_wrenches) ? launch_variant<TEAM, MODE, true>(b, A, grid, lds) : launch_variant<TEAM, MODE, false>(b, A, grid, lds);
#endif
}

template <int TEAM>
int launch_team(tg_batch *b, const tg::RunArgs &A, int grid, size_t lds) {
    switch (A.mode) {
    case tg::MODE_ROLLOUT: return launch_one<TEAM, tg::MODE_ROLLOUT>(b, A, grid, lds);
    case tg::MODE_CALC_P2: return launch_one<TEAM, tg::MODE_CALC_P2>(b, A, grid, lds);
    case tg::MODE_CALC_F: return launch_one<TEAM, tg::MODE_CALC_F>(b, A, grid, lds);
    case tg::MODE_DERIV1: return launch_one<TEAM, tg::MODE_DERIV1>(b, A, grid, lds);
    case tg::MODE_DERIV2Z: return launch_one<TEAM, tg::MODE_DERIV2Z>(b, A, grid, lds);
#if !defined(TG_PROFILE)   // the continuous-dynamics modes are not instrumented (and not instantiated) in the diagnostic build
    case tg::MODE_DYNAMICS: return launch_one<TEAM, tg::MODE_DYNAMICS>(b, A, grid, lds);
    case tg::MODE_DYN_DERIV1: return launch_one<TEAM, tg::MODE_DYN_DERIV1>(b, A, grid, lds);
    case tg::MODE_ENERGY: return launch_one<TEAM, tg::MODE_ENERGY>(b, A, grid, lds);
    case tg::MODE_LAGRANGIAN: return launch_one<TEAM, tg::MODE_LAGRANGIAN>(b, A, grid, lds);
#endif
    default: return fail(TG_ERR_INVALID, "unknown kernel mode");
    }
}

int launch(tg_batch *b, tg::RunArgs &A) {
    b->mirror_valid = false;
    const int team = b->sys->team, per_block = 64 / team;
    const int grid = ((A.remap_len > 0 ? A.remap_count : A.batch) + per_block - 1) / per_block;
    const int per_team = A.mode == tg::MODE_DERIV2Z ? b->P.e_lds_per_team : (A.mode == tg::MODE_DERIV1 ? b->P.a_lds_per_team : (A.mode == tg::MODE_DYN_DERIV1 ? b->P.g_lds_per_team : b->P.lds_per_team));
    const size_t lds = (size_t)per_block * per_team * sizeof(double);
    if (lds > 160 * 1024) return fail(TG_ERR_UNSUPPORTED, "system too large for the LDS-resident kernel");
    // HIP-event timing is opt-in (the first tg_batch_timing call switches it on): a plain MidpointVI.step() loop creates
    // no events.  When on, at most TIMING_CAP launches are kept; older pairs are recycled (their time is folded into
    // the running totals), and every error path hands the pair back to the pool.
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (b->timing) {
        if (b->events.size() >= tg_batch::TIMING_CAP) {
            auto old = b->events.front();
            float t = 0.f;
            if (hipEventSynchronize(old.second) == hipSuccess && hipEventElapsedTime(&t, old.first, old.second) == hipSuccess) { b->folded_ms += t; b->folded_n++; }
            b->events.erase(b->events.begin());
            e0 = old.first; e1 = old.second;
        } else if (b->pool.size() >= 2) { e0 = b->pool.back(); b->pool.pop_back(); e1 = b->pool.back(); b->pool.pop_back(); }
        else {
            HIP_TRY(hipEventCreate(&e0));
            if (hipEventCreate(&e1) != hipSuccess) { b->pool.push_back(e0); return fail(TG_ERR_HIP, "hipEventCreate failed"); }
        }
        if (hipEventRecord(e0, b->stream) != hipSuccess) { b->pool.push_back(e0); b->pool.push_back(e1); return fail(TG_ERR_HIP, "hipEventRecord failed"); }
    }
    int rc;
    if (b->spec_launch && ((b->spec_modes >> A.mode) & 1)) {
        const int i = b->arg_next;
        b->arg_next = (i + 1) % tg_batch::ARG_SLOTS;
        rc = TG_SUCCESS;
        if (b->arg_used[i] && hipEventSynchronize(b->arg_done[i]) != hipSuccess) rc = fail(TG_ERR_HIP, "hipEventSynchronize failed");
        if (rc == TG_SUCCESS) {
            b->h_args[i] = A;
            rc = b->spec_launch(A.mode, &b->h_args[i], b->d_args + i, grid, lds, (void *)b->stream) == 0 ? TG_SUCCESS : fail(TG_ERR_HIP, "specialised kernel launch failed");
            b->arg_used[i] = hipEventRecord(b->arg_done[i], b->stream) == hipSuccess;
            if (!b->arg_used[i]) hipStreamSynchronize(b->stream);
            b->spec_launched_modes |= 1u << A.mode; b->spec_launches++;
        }
    } else {
    b->generic_launched_modes |= 1u << A.mode; b->generic_launches++;
    rc = team == 64 ? launch_team<64>(b, A, grid, lds) : (team == 16 ? launch_team<16>(b, A, grid, lds)
             : (team == 4 ? launch_team<4>(b, A, grid, lds) : launch_team<1>(b, A, grid, lds)));
    }
    if (rc == TG_SUCCESS && hipGetLastError() != hipSuccess) rc = fail(TG_ERR_HIP, "kernel launch failed");
    if (b->timing) {
        if (rc != TG_SUCCESS || hipEventRecord(e1, b->stream) != hipSuccess) {
            b->pool.push_back(e0); b->pool.push_back(e1);
            return rc != TG_SUCCESS ? rc : fail(TG_ERR_HIP, "hipEventRecord failed");
        }
        b->events.emplace_back(e0, e1);
    }
    return rc;
}

#if !defined(TG_PROFILE)
template <int MODE, bool SPRINGS, class Real>
int launch_forward_variant(tg_batch *b, const tg::RunArgs &A, size_t lds) {
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_forward<MODE, SPRINGS, Real>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_forward<MODE, SPRINGS, Real>), dim3(A.batch), dim3(64), lds, b->stream, b->d_prog, A);
    return TG_SUCCESS;
}
template <int MODE, class Real>
int launch_forward_mode(tg_batch *b, const tg::RunArgs &A) {
    const int per_team = std::max(b->P.lds_per_team, MODE == tg::MODE_DYN_DERIV1 ? b->P.g_lds_per_team : 0);
    const size_t lds = (size_t)per_team * sizeof(Real);
    if (lds > 160 * 1024) return fail(TG_ERR_UNSUPPORTED, "system too large for the LDS-resident forward-mode kernel");
    const bool springs = b->P.has_cs || b->P.n_springs || b->P.has_plane || b->P.n_wrenches;
    int rc = springs ? launch_forward_variant<MODE, true, Real>(b, A, lds) : launch_forward_variant<MODE, false, Real>(b, A, lds);
    if (rc == TG_SUCCESS && hipGetLastError() != hipSuccess) rc = fail(TG_ERR_HIP, "kernel launch failed");
    return rc;
}
#endif
// A continuous-dynamics mode on dual numbers: A.seed1 (and A.seed2 for order 2) name the direction variable(s) of every trajectory.
int launch_forward(tg_batch *b, const tg::RunArgs &A, int order) {
#if defined(TG_PROFILE)
    (void)b; (void)A; (void)order;
    return fail(TG_ERR_UNSUPPORTED, "the profiling build has no forward-mode kernels");
#else
    typedef tgdual::Dual<double> D1;
    typedef tgdual::Dual<D1> D2;
    b->mirror_valid = false;
    if (A.mode == tg::MODE_DYN_DERIV1 && order == 1) return launch_forward_mode<tg::MODE_DYN_DERIV1, D1>(b, A);
    if (A.mode == tg::MODE_LAGRANGIAN && order == 1) return launch_forward_mode<tg::MODE_LAGRANGIAN, D1>(b, A);
    if (A.mode == tg::MODE_LAGRANGIAN && order == 2) return launch_forward_mode<tg::MODE_LAGRANGIAN, D2>(b, A);
    return fail(TG_ERR_INVALID, "forward mode: first derivatives of the dynamics (order 1) or the Lagrangian (order 1, 2)");
#endif
}
// the direction variables of a forward-mode call, checked and copied to the device; null seed2: first order
int stage_seeds(tg_batch *b, const int32_t *seed1_host, const int32_t *seed2_host) {
    const size_t B = (size_t)b->batch;
    const int nvar = 2 * b->P.nq + b->P.nk + b->P.nu;
    for (size_t i = 0; i < B; i++) {
        if (seed1_host[i] < -1 || seed1_host[i] >= nvar) return fail(TG_ERR_INVALID, "direction variable out of range (q | dq | ddq_k | u)");
        if (seed2_host && (seed2_host[i] < -1 || seed2_host[i] >= nvar)) return fail(TG_ERR_INVALID, "direction variable out of range (q | dq | ddq_k | u)");
    }
    if (!b->seeds) HIP_TRY(hipMalloc(&b->seeds, 2 * B * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(b->seeds, seed1_host, B * sizeof(int), hipMemcpyHostToDevice, b->stream));
    if (seed2_host) HIP_TRY(hipMemcpyAsync(b->seeds + B, seed2_host, B * sizeof(int), hipMemcpyHostToDevice, b->stream));
    return TG_SUCCESS;
}

// Host-facing derivative outputs: the twelve first-derivative arrays and the contraction buffers.  Allocated on
// first use so that batches that only roll out, or that write A/B and HZ into caller-provided device buffers,
// do not reserve 80+ kB per trajectory.
int ensure_deriv_buffers(tg_batch *b, bool first, bool second) {
    const tg::DevProg &P = b->P;
    const size_t B = (size_t)b->batch;
    auto dalloc = [&](double **p, size_t n) -> bool {
        if (*p) return true;
        return hipMalloc(p, (n ? n : 1) * sizeof(double)) == hipSuccess && hipMemset(*p, 0, (n ? n : 1) * sizeof(double)) == hipSuccess;
    };
    bool ok = true;
    if (first) for (int k = 0; k < 12 && ok; k++) {
        const int var = k % 4, out = k / 4;
        const size_t rows = var == 0 ? P.nq : (var == 1 ? P.nd : (var == 2 ? P.nu : P.nk));
        ok = dalloc(&b->d1[k], B * rows * (out == 2 ? P.nc : P.nd));
    }
    if (second && ok) ok = dalloc(&b->z_dev, B * P.nX) && dalloc(&b->zl_dev, B * P.nc) && dalloc(&b->hz_dev, B * (size_t)P.d_nrhs * P.d_nrhs);
    return ok ? TG_SUCCESS : fail(TG_ERR_HIP, "device allocation failed");
}

tg::RunArgs base_args(tg_batch *b, int mode) {
    tg::RunArgs A{};
    A.batch = b->batch; A.mode = mode; A.max_iterations = 200;
    A.t1 = b->t1; A.t2 = b->t2; A.tolerance = b->tolerance; A.predictor = b->predictor; A.exact_pivot = b->exact_pivot;
    A.dt_steps = b->dt_host.empty() ? nullptr : b->dt_dev;
    A.dt_period = (b->dt_by_trajectory && !b->dt_host.empty()) ? (int)b->dt_host.size() : 0;
    A.q1 = b->q1; A.q2 = b->q2; A.p1 = b->p1; A.p2 = b->p2; A.lam = b->lam; A.u1 = b->u1;
    A.iters = b->iters; A.status = b->status; A.f_out = b->f_out; A.fallbacks = b->fallbacks;
    A.prof_out = b->prof;
    for (int i = 0; i < 12; i++) A.d1[i] = b->d1[i];
    A.z = b->z_dev; A.hz = b->hz_dev;
    A.group_size = 1;
    return A;
}

// DSystem.set(X[s][k], U[s][k], k, xk_hint = X[s][k+1]) for trajectory t = s*horizon + k (dsystem.py:229-251):
// state (q2,p2) <- X[s][k] (the step kernel shifts it into slot 1), lambda <- 0, inputs and hint staged.
__global__ void k_set_from_trajectories(const tg::DevProg P, int seeds, int horizon, const double *X, const double *U,
                                        double *q1, double *q2, double *p1, double *p2, double *lam, double *su,
                                        double *sk, double *sqh) {
    const size_t t = blockIdx.x;
    const size_t s = t / horizon, k = t % horizon;
    const double *x0 = X + (s * (horizon + 1) + k) * P.nX, *x1 = x0 + P.nX, *u = U + (s * horizon + k) * (size_t)(P.nu + P.nk);
    for (int i = threadIdx.x; i < P.nq; i += blockDim.x) { q1[t * P.nq + i] = x0[i]; q2[t * P.nq + i] = x0[i]; }
    for (int i = threadIdx.x; i < P.nd; i += blockDim.x) {
        p1[t * P.nd + i] = x0[P.nq + i]; p2[t * P.nd + i] = x0[P.nq + i];
        sqh[t * P.nd + i] = x1[i];
    }
    for (int i = threadIdx.x; i < P.nc; i += blockDim.x) lam[t * P.nc + i] = 0.0;
    for (int i = threadIdx.x; i < P.nu; i += blockDim.x) su[t * P.nu + i] = u[i];
    for (int i = threadIdx.x; i < P.nk; i += blockDim.x) sk[t * P.nk + i] = u[P.nu + i];
}

// initialize_from_state(t, Q, p) with (Q, p) taken from the head of a DSystem state vector X = [Q; p; v]
__global__ void k_init_from_X(const tg::DevProg P, const double *X, size_t stride, double *q1, double *q2, double *p1,
                              double *p2, double *lam) {
    const size_t t = blockIdx.x;
    const double *x = X + t * stride;
    for (int i = threadIdx.x; i < P.nq; i += blockDim.x) { q1[t * P.nq + i] = x[i]; q2[t * P.nq + i] = x[i]; }
    for (int i = threadIdx.x; i < P.nd; i += blockDim.x) { p1[t * P.nd + i] = x[P.nq + i]; p2[t * P.nd + i] = x[P.nq + i]; }
    for (int i = threadIdx.x; i < P.nc; i += blockDim.x) lam[t * P.nc + i] = 0.0;
}


// Test hook: the Newton-system solver of the rollout kernels (gj_rows) on a caller-supplied matrix, with its pivot order.
__global__ void k_debug_solve(int n, int ld, int exact, const double *A_in, double *x_out, int *piv_out, int *status_out) {
#if defined(__HIP_DEVICE_COMPILE__)   // gj_rows exists in the device pass only
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    int *trace = (int *)(lds + n * ld);
    for (int e = lane; e < n * (n + 1); e += 64) lds[(e / (n + 1)) * ld + e % (n + 1)] = A_in[e];
    if (lane < 32) trace[lane] = -1;
    __syncthreads();
    bool ok = true;
    if (exact == 1) switch ((n + 3) >> 2) {
    case 1: ok = tg::Core<64>::gj_rows_exact<4, true>(true, lds, n, ld, lane, trace); break;
    case 2: ok = tg::Core<64>::gj_rows_exact<8, true>(true, lds, n, ld, lane, trace); break;
    case 3: ok = tg::Core<64>::gj_rows_exact<12, true>(true, lds, n, ld, lane, trace); break;
    case 4: ok = tg::Core<64>::gj_rows_exact<16, true>(true, lds, n, ld, lane, trace); break;
    case 5: ok = tg::Core<64>::gj_rows_exact<20, true>(true, lds, n, ld, lane, trace); break;
    case 6: ok = tg::Core<64>::gj_rows_exact<24, true>(true, lds, n, ld, lane, trace); break;
    case 7: ok = tg::Core<64>::gj_rows_exact<28, true>(true, lds, n, ld, lane, trace); break;
    default: ok = tg::Core<64>::gj_rows_exact<32, true>(true, lds, n, ld, lane, trace); break;
    }
    else if (exact == 2) {     // the full-wave panel solver (default pivot rule), 16 < n < 32; scratch behind the trace words
        double *scr = lds + n * ld + 16;
        switch ((n + 3) >> 2) {
        case 5: ok = tg::Core<64>::gj_panel<20, true>(true, lds, n, ld, lane, scr, trace); break;
        case 6: ok = tg::Core<64>::gj_panel<24, true>(true, lds, n, ld, lane, scr, trace); break;
        case 7: ok = tg::Core<64>::gj_panel<28, true>(true, lds, n, ld, lane, scr, trace); break;
        default: ok = tg::Core<64>::gj_panel<32, true>(true, lds, n, ld, lane, scr, trace); break;
        }
    }
    else switch ((n + 3) >> 2) {
    case 1: ok = tg::Core<64>::gj_rows<4, true>(true, lds, n, ld, lane, trace); break;
    case 2: ok = tg::Core<64>::gj_rows<8, true>(true, lds, n, ld, lane, trace); break;
    case 3: ok = tg::Core<64>::gj_rows<12, true>(true, lds, n, ld, lane, trace); break;
    case 4: ok = tg::Core<64>::gj_rows<16, true>(true, lds, n, ld, lane, trace); break;
    case 5: ok = tg::Core<64>::gj_rows<20, true>(true, lds, n, ld, lane, trace); break;
    case 6: ok = tg::Core<64>::gj_rows<24, true>(true, lds, n, ld, lane, trace); break;
    case 7: ok = tg::Core<64>::gj_rows<28, true>(true, lds, n, ld, lane, trace); break;
    default: ok = tg::Core<64>::gj_rows<32, true>(true, lds, n, ld, lane, trace); break;
    }
    __syncthreads();
    if (lane < n) { x_out[lane] = lds[lane * ld + n]; piv_out[lane] = trace[lane]; }
    if (lane == 0) *status_out = ok ? TG_OK : TG_SINGULAR;
#endif
}

}  // namespace

namespace tg_detail {
int fail(int code, const std::string &msg) { return ::fail(code, msg); }
}

extern "C" {

const char *tg_version(void) { return "trep_amd 0.1 (gfx950)"; }
const char *tg_last_error(void) { return g_error.c_str(); }

int tg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* out[0..3] = compute units, LDS bytes a workgroup may use (opt-in maximum), wavefront size, 0 */
int tg_device_info(int32_t device, int32_t out[4]) {
    if (!out) return fail(TG_ERR_INVALID, "null argument");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    out[0] = prop.multiProcessorCount; out[1] = (int32_t)prop.sharedMemPerBlockOptin; out[2] = prop.warpSize; out[3] = 0;
    return TG_SUCCESS;
}

tg_system *tg_system_create(const tg_system_desc *desc) {
    if (!desc) { fail(TG_ERR_INVALID, "null descriptor"); return nullptr; }
    try {
        tg_system *s = new tg_system();
        s->H = tg::build_program(desc);
        s->team = pick_team(s->H);
        return s;
    } catch (const std::exception &e) {
        fail(TG_ERR_INVALID, e.what());
        return nullptr;
    }
}
void tg_system_destroy(tg_system *sys) { delete sys; }

int tg_system_sizes(const tg_system *sys, int32_t out[6]) {
    if (!sys) return fail(TG_ERR_INVALID, "null system");
    const tg::DevProg &P = sys->H.p;
    out[0] = P.nq; out[1] = P.nd; out[2] = P.nk; out[3] = P.nu; out[4] = P.nc; out[5] = P.nX;
    return TG_SUCCESS;
}

/* Introspection used by bench.py / DESIGN.md: team size, LDS bytes per trajectory, schedule sizes. */
int tg_system_info(const tg_system *sys, int32_t out[8]) {
    if (!sys) return fail(TG_ERR_INVALID, "null system");
    const tg::DevProg &P = sys->H.p;
    out[0] = sys->team; out[1] = (int32_t)(P.lds_per_team * sizeof(double)); out[2] = P.n_joints; out[3] = P.n_levels;
    out[4] = P.n_bodies; out[5] = P.n_items; out[6] = P.n_pairs; out[7] = P.n_dh;
    return TG_SUCCESS;
}

tg_batch *tg_batch_create(tg_system *sys, int32_t batch, int32_t device) {
    if (!sys || batch <= 0) { fail(TG_ERR_INVALID, "bad arguments"); return nullptr; }
    int ndev = tg_device_count();
    if (ndev <= 0) { fail(TG_ERR_HIP, "no HIP device visible: libtrepamd has no CPU path"); return nullptr; }
    if (device < 0 || device >= ndev) { fail(TG_ERR_INVALID, "device index out of range"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(TG_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    tg_batch *b = new tg_batch();
    b->sys = sys; b->batch = batch; b->device = device;
    const tg::HostProgram &H = sys->H;
    b->P = H.p;
    // all index / constant tables live in two device buffers; bind() points the DevProg into them
    const std::vector<int> &ints = H.ipool;
    const std::vector<double> &dbls = H.dpool;
    bool ok = hipMalloc(&b->d_ints, ints.size() * sizeof(int)) == hipSuccess &&
              hipMalloc(&b->d_dbls, dbls.size() * sizeof(double)) == hipSuccess &&
              hipMemcpy(b->d_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(b->d_dbls, dbls.data(), dbls.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    tg::DevProg &P = b->P;
    H.bind(P, b->d_ints, b->d_dbls);
    if (ok) ok = hipMalloc(&b->d_prog, sizeof(tg::DevProg)) == hipSuccess &&
                 hipMemcpy(b->d_prog, &P, sizeof(tg::DevProg), hipMemcpyHostToDevice) == hipSuccess;
    auto dalloc = [&](double **p, size_t n) {
        if (!ok) return;
        ok = hipMalloc(p, (n ? n : 1) * sizeof(double)) == hipSuccess && hipMemset(*p, 0, (n ? n : 1) * sizeof(double)) == hipSuccess;
    };
    const size_t B = (size_t)batch;
    dalloc(&b->q1, B * P.nq); dalloc(&b->q2, B * P.nq); dalloc(&b->p1, B * P.nd); dalloc(&b->p2, B * P.nd);
    dalloc(&b->lam, B * P.nc); dalloc(&b->u1, B * P.nu);
    dalloc(&b->stage_u, B * P.nu); dalloc(&b->stage_k, B * P.nk); dalloc(&b->stage_qh, B * P.nd); dalloc(&b->stage_lh, B * P.nc);
    dalloc(&b->f_out, B * P.nf);
    // derivative outputs (d1[12], z, hz) are allocated on first use: ensure_deriv_buffers()
    dalloc(&b->snap, B * (2 * (size_t)P.nq + 2 * (size_t)P.nd + P.nc + P.nu));
    if (ok) ok = hipMalloc(&b->iters, B * sizeof(int)) == hipSuccess && hipMalloc(&b->status, B * sizeof(int)) == hipSuccess &&
                 hipMemset(b->iters, 0, B * sizeof(int)) == hipSuccess && hipMemset(b->status, 0, B * sizeof(int)) == hipSuccess;
    if (ok) ok = hipMalloc(&b->fallbacks, B * sizeof(int)) == hipSuccess && hipMemset(b->fallbacks, 0, B * sizeof(int)) == hipSuccess;
#if defined(TG_PROFILE)
    if (ok) ok = hipMalloc(&b->prof, 16 * sizeof(long long)) == hipSuccess && hipMemset(b->prof, 0, 16 * sizeof(long long)) == hipSuccess;
#endif
    if (ok) ok = hipStreamCreate(&b->stream) == hipSuccess;
    if (!ok) { fail(TG_ERR_HIP, "device allocation failed"); tg_batch_destroy(b); return nullptr; }
    return b;
}

void tg_batch_destroy(tg_batch *b) {
    if (!b) return;
    hipSetDevice(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    for (auto &e : b->events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : b->pool) hipEventDestroy(e);
    void *ptrs[] = {b->d_args, b->dt_dev, b->d_prog, b->d_ints, b->d_dbls, b->q1, b->q2, b->p1, b->p2, b->lam, b->u1, b->stage_u, b->stage_k,
                    b->stage_qh, b->stage_lh, b->f_out, b->iters, b->status, b->fallbacks, b->snap, b->z_dev, b->hz_dev, b->zl_dev, b->dyn, b->dyn_ints, b->seeds, b->dyn_d1, b->energy, b->lag,
                    b->d1[0], b->d1[1], b->d1[2], b->d1[3], b->d1[4], b->d1[5], b->d1[6], b->d1[7], b->d1[8], b->d1[9], b->d1[10], b->d1[11]};
    for (void *p : ptrs) if (p) hipFree(p);
    if (b->h_args) hipHostFree(b->h_args);
    if (b->io_host) hipHostFree(b->io_host);
    for (auto &e : b->arg_done) if (e) hipEventDestroy(e);
    if (b->stream && b->own_stream) hipStreamDestroy(b->stream);
    if (b->spec_lib) dlclose(b->spec_lib);
    delete b;
}

int tg_batch_set_tolerance(tg_batch *b, double tolerance) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    b->tolerance = tolerance;
    return TG_SUCCESS;
}
int tg_batch_set_times(tg_batch *b, double t1, double t2) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    b->t1 = t1; b->t2 = t2;
    return TG_SUCCESS;
}
int tg_batch_get_times(const tg_batch *b, double *t1, double *t2) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    *t1 = b->t1; *t2 = b->t2;
    return TG_SUCCESS;
}
int tg_batch_field_width(const tg_batch *b, int32_t field) { return b ? widths(b, field) : -1; }

int tg_batch_set(tg_batch *b, int32_t field, const double *host) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    int w = widths(b, field);
    double *dst = field_ptr(b, field);
    if (w < 0 || !dst) return fail(TG_ERR_INVALID, "unknown or read-only field");
    if (w == 0) return TG_SUCCESS;
    b->mirror_valid = false;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(dst, host, (size_t)b->batch * w * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}
int tg_batch_get(tg_batch *b, int32_t field, double *host) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    int w = widths(b, field);
    double *src = field_ptr(b, field);
    if (w < 0 || !src) return fail(TG_ERR_INVALID, "unknown field");
    if (w == 0) return TG_SUCCESS;
    if (b->mirror_valid && (field == TG_F_Q2 || field == TG_F_P2 || field == TG_F_LAMBDA1)) {   // answered from the last step's mirror
        const tg::DevProg &P = b->P;
        const size_t B = (size_t)b->batch;
        const double *m = b->io_host + b->io_in + (field == TG_F_Q2 ? 0 : (field == TG_F_P2 ? B * P.nq : B * (P.nq + P.nd)));
        std::memcpy(host, m, B * w * sizeof(double));
        return TG_SUCCESS;
    }
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(host, src, (size_t)b->batch * w * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_calc_p2(tg_batch *b) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "calc_p2 needs t2 != t1");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_CALC_P2);
    int rc = launch(b, A);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_calc_f(tg_batch *b, double *f_host) {
    if (!b || !f_host) return fail(TG_ERR_INVALID, "null argument");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "calc_f needs t2 != t1");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_CALC_F);
    int rc = launch(b, A);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(f_host, b->f_out, (size_t)b->batch * b->P.nf * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

/* Per trajectory: how many Newton systems of the last rollout / step launch the structured solve (bbd.hpp) handed to the pivoting
 * solver because a pivot guard failed.  Zero for kernels without a structured solve.  Results are correct either way; a batch that
 * reports fallbacks on most systems (very small time steps, very heavy bodies) runs slower than with the pivoting solver alone. */
int tg_batch_solver_fallbacks(tg_batch *b, int32_t *fallbacks_out) {
    if (!b || !fallbacks_out) return fail(TG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(fallbacks_out, b->fallbacks, (size_t)b->batch * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_status(tg_batch *b, int32_t *iterations_out, int32_t *status_out) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    if (b->mirror_valid) {
        const tg::DevProg &P = b->P;
        const size_t B = (size_t)b->batch;
        const int32_t *m = reinterpret_cast<const int32_t *>(b->io_host + b->io_in + B * (P.nq + P.nd + P.nc));
        if (iterations_out) std::memcpy(iterations_out, m, B * sizeof(int32_t));
        if (status_out) std::memcpy(status_out, m + B, B * sizeof(int32_t));
        return TG_SUCCESS;
    }
    HIP_TRY(hipSetDevice(b->device));
    if (iterations_out) HIP_TRY(hipMemcpyAsync(iterations_out, b->iters, (size_t)b->batch * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    if (status_out) HIP_TRY(hipMemcpyAsync(status_out, b->status, (size_t)b->batch * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_step(tg_batch *b, double t2_new, const double *u1_host, const double *k2_host, const double *q2_hint_host,
                  const double *lambda_hint_host, int32_t max_iterations, int32_t *iterations_out, int32_t *status_out) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    const tg::DevProg &P = b->P;
    if ((P.nu && !u1_host) || (P.nk && !k2_host)) return fail(TG_ERR_INVALID, "u1 / k2 required");
    if (t2_new == b->t2) return fail(TG_ERR_STATE, "step needs t2_new != t2");
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch;
    const bool want_lh = lambda_hint_host && P.nc;
    // Small batches: the call is host-latency bound (a MidpointVI.step() loop; tools/step_latency.py).  Inputs and results live in one
    // pinned, device-visible host block: the kernel reads (u1, k2, hints) from it at the head of the step and writes (q2, p2, lambda1,
    // iterations, status) into it beside the device state -- one launch on the stream, no copy engine, no packing kernel.
    const size_t n_in = B * ((size_t)P.nu + P.nk + P.nd + P.nc), n_out = B * ((size_t)P.nq + P.nd + P.nc) + B;   // 2 B ints = B doubles
    // (gated on the trajectory count: the path was measured at B = 1 .. 64 only; larger batches take the copy engine below)
    if (B <= 64 && (n_in + n_out) * sizeof(double) <= (1u << 20)) {
        if (!b->io_host) {
            if (hipHostMalloc(&b->io_host, (n_in + n_out) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                hipHostGetDevicePointer(reinterpret_cast<void **>(&b->io_dev), b->io_host, 0) != hipSuccess)
                return fail(TG_ERR_HIP, "allocation of the step staging block failed");
            b->io_in = n_in; b->io_out = n_out;
        }
        double *hu = b->io_host, *hk = hu + B * P.nu, *hq = hk + B * P.nk, *hl = hq + B * P.nd;
        double *du = b->io_dev, *dk = du + B * P.nu, *dq = dk + B * P.nk, *dl = dq + B * P.nd;    // the same block as the device sees it
        if (P.nu) std::memcpy(hu, u1_host, B * P.nu * sizeof(double));
        if (P.nk) std::memcpy(hk, k2_host, B * P.nk * sizeof(double));
        if (q2_hint_host) std::memcpy(hq, q2_hint_host, B * P.nd * sizeof(double));
        if (want_lh) std::memcpy(hl, lambda_hint_host, B * P.nc * sizeof(double));
        tg::RunArgs A = base_args(b, tg::MODE_ROLLOUT);
        A.n_steps = 1; A.dt = t2_new - b->t2; A.max_iterations = max_iterations;
        A.U = du; A.K = dk;
        A.q2_hint = q2_hint_host ? dq : nullptr;
        A.lam_hint = want_lh ? dl : nullptr;
        A.mirror = b->io_dev + n_in;
        int rc = launch(b, A);
        if (rc) return rc;
        b->t1 = b->t2; b->t2 = t2_new;
        HIP_TRY(hipStreamSynchronize(b->stream));
        b->mirror_valid = true;
        return tg_batch_status(b, iterations_out, status_out);
    }
    if (P.nu) HIP_TRY(hipMemcpyAsync(b->stage_u, u1_host, B * P.nu * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (P.nk) HIP_TRY(hipMemcpyAsync(b->stage_k, k2_host, B * P.nk * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (q2_hint_host) HIP_TRY(hipMemcpyAsync(b->stage_qh, q2_hint_host, B * P.nd * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (want_lh) HIP_TRY(hipMemcpyAsync(b->stage_lh, lambda_hint_host, B * P.nc * sizeof(double), hipMemcpyHostToDevice, b->stream));
    tg::RunArgs A = base_args(b, tg::MODE_ROLLOUT);
    A.n_steps = 1; A.dt = t2_new - b->t2; A.max_iterations = max_iterations;
    A.U = b->stage_u; A.K = b->stage_k;
    A.q2_hint = q2_hint_host ? b->stage_qh : nullptr;
    A.lam_hint = want_lh ? b->stage_lh : nullptr;
    int rc = launch(b, A);
    if (rc) return rc;
    b->t1 = b->t2; b->t2 = t2_new;
    return tg_batch_status(b, iterations_out, status_out);
}

// t1, t2 after n_steps steps from t2 with the uniform step dt or the batch's step-size list
static int advance_times(tg_batch *b, int n_steps, double dt) {
    if (!b->dt_host.empty() && !b->dt_by_trajectory) {
        if ((size_t)n_steps > b->dt_host.size()) return fail(TG_ERR_INVALID, "rollout longer than the step-size list of tg_batch_set_step_sizes");
        double t = b->t2, tp = b->t2;
        for (int k = 0; k < n_steps; k++) { tp = t; t += b->dt_host[k]; }
        b->t1 = tp; b->t2 = t;
    } else {
        b->t1 = b->t2 + (n_steps - 1) * dt;
        b->t2 = b->t2 + n_steps * dt;
    }
    return TG_SUCCESS;
}

int tg_batch_set_step_sizes(tg_batch *b, int32_t count, const double *dt_host, int32_t by_trajectory) {
    if (!b || count < 0 || (count > 0 && !dt_host)) return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));          // a launch in flight may still read the old list
    if (b->dt_dev) { HIP_TRY(hipFree(b->dt_dev)); b->dt_dev = nullptr; }
    b->dt_host.clear();
    b->dt_by_trajectory = 0;
    if (count == 0) return TG_SUCCESS;
    for (int i = 0; i < count; i++) if (dt_host[i] == 0.0) return fail(TG_ERR_INVALID, "zero step size");
    HIP_TRY(hipMalloc(&b->dt_dev, sizeof(double) * (size_t)count));
    HIP_TRY(hipMemcpy(b->dt_dev, dt_host, sizeof(double) * (size_t)count, hipMemcpyHostToDevice));
    b->dt_host.assign(dt_host, dt_host + count);
    b->dt_by_trajectory = by_trajectory ? 1 : 0;
    return TG_SUCCESS;
}

int tg_batch_rollout(tg_batch *b, int32_t n_steps, double dt, const double *U_dev, const double *K_dev, double *X_dev,
                     int32_t max_iterations) {
    if (!b || n_steps <= 0 || dt == 0.0) return fail(TG_ERR_INVALID, "bad arguments");
    if (!b->dt_host.empty() && !b->dt_by_trajectory && (size_t)n_steps > b->dt_host.size()) return fail(TG_ERR_INVALID, "rollout longer than the step-size list");
    const tg::DevProg &P = b->P;
    if ((P.nu && !U_dev) || (P.nk && !K_dev)) return fail(TG_ERR_INVALID, "U / K device buffers required");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_ROLLOUT);
    A.n_steps = n_steps; A.dt = dt; A.max_iterations = max_iterations;
    A.U = U_dev; A.K = K_dev; A.X = X_dev;
    int rc = launch(b, A);
    if (rc) return rc;
    return advance_times(b, n_steps, dt);
}

int tg_batch_rollout_closed_loop(tg_batch *b, int32_t n_steps, double dt, const double *Kproj_dev, int32_t group_size,
                                 const double *bX_dev, const double *bU_dev, double *X_dev, double *U_dev,
                                 int32_t max_iterations) {
    if (!b || n_steps <= 0 || dt == 0.0 || !Kproj_dev || !bX_dev || !bU_dev || group_size <= 0)
        return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_ROLLOUT);
    A.n_steps = n_steps; A.dt = dt; A.max_iterations = max_iterations;
    A.Kproj = Kproj_dev; A.bX = bX_dev; A.bU = bU_dev; A.Uout = U_dev; A.group_size = group_size; A.X = X_dev;
    int rc = launch(b, A);
    if (rc) return rc;
    return advance_times(b, n_steps, dt);
}

int tg_batch_rollout_closed_loop_subset(tg_batch *b, int32_t n_trajectories, int32_t n_steps, double dt, const double *Kproj_dev,
                                        int32_t group_size, const int32_t *group_select_dev, const double *bX_dev,
                                        const double *bU_dev, double *X_dev, double *U_dev, int32_t max_iterations) {
    if (!b || n_steps <= 0 || dt == 0.0 || !Kproj_dev || !bX_dev || !bU_dev || group_size <= 0 || n_trajectories <= 0 ||
        n_trajectories > b->batch)
        return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_ROLLOUT);
    A.batch = n_trajectories;
    A.n_steps = n_steps; A.dt = dt; A.max_iterations = max_iterations;
    A.Kproj = Kproj_dev; A.bX = bX_dev; A.bU = bU_dev; A.Uout = U_dev; A.group_size = group_size; A.X = X_dev;
    A.group_map = group_select_dev;
    int rc = launch(b, A);
    if (rc) return rc;
    return advance_times(b, n_steps, dt);
}

int tg_batch_rollout_stats(tg_batch *b, int64_t *total_iterations, int32_t *n_failed) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    std::vector<int> it(b->batch), st(b->batch);
    int rc = tg_batch_status(b, it.data(), st.data());
    if (rc) return rc;
    int64_t tot = 0; int nf = 0;
    for (int i = 0; i < b->batch; i++) { tot += it[i]; nf += (st[i] != TG_OK); }
    if (total_iterations) *total_iterations = tot;
    if (n_failed) *n_failed = nf;
    return TG_SUCCESS;
}

static int snapshot_copy(tg_batch *b, bool save) {
    const tg::DevProg &P = b->P;
    const size_t B = (size_t)b->batch;
    double *fields[6] = {b->q1, b->q2, b->p1, b->p2, b->lam, b->u1};
    const size_t w[6] = {(size_t)P.nq, (size_t)P.nq, (size_t)P.nd, (size_t)P.nd, (size_t)P.nc, (size_t)P.nu};
    size_t off = 0;
    for (int i = 0; i < 6; i++) {
        if (w[i]) {
            double *dst = save ? b->snap + off : fields[i];
            const double *src = save ? fields[i] : b->snap + off;
            HIP_TRY(hipMemcpyAsync(dst, src, B * w[i] * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        }
        off += B * w[i];
    }
    return TG_SUCCESS;
}

int tg_batch_snapshot(tg_batch *b) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    b->snap_t1 = b->t1; b->snap_t2 = b->t2;
    return snapshot_copy(b, true);
}

int tg_batch_restore(tg_batch *b) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    b->mirror_valid = false;
    b->t1 = b->snap_t1; b->t2 = b->snap_t2;
    return snapshot_copy(b, false);
}

/* Diagnostic build (make prof): per-phase cycle counters of trajectory 0 of the last launch. */
int tg_batch_profile(tg_batch *b, int64_t out[16]) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    if (!b->prof) return fail(TG_ERR_UNSUPPORTED, "library was not built with TG_PROFILE");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipMemcpy(out, b->prof, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return TG_SUCCESS;
}

int tg_batch_deriv1(tg_batch *b) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "Integrator has not solved the next time step yet.");
    HIP_TRY(hipSetDevice(b->device));
    int rc = ensure_deriv_buffers(b, true, false);
    if (rc) return rc;
    tg::RunArgs A = base_args(b, tg::MODE_DERIV1);
    rc = launch(b, A);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->have_d1 = true;
    return TG_SUCCESS;
}

int tg_batch_deriv2_contract(tg_batch *b, const double *z_host, double *hz_host) {
    if (!b || !z_host || !hz_host) return fail(TG_ERR_INVALID, "null argument");
    if (b->P.n_true_springs) return fail(TG_ERR_UNSUPPORTED, "V_dqdqdq() is undefined for LinearSpring (as in the reference): no second derivatives");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "Integrator has not solved the next time step yet.");
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch, R = (size_t)b->P.d_nrhs;
    if (int rc0 = ensure_deriv_buffers(b, false, true)) return rc0;
    HIP_TRY(hipMemcpyAsync(b->z_dev, z_host, B * b->P.nX * sizeof(double), hipMemcpyHostToDevice, b->stream));
    tg::RunArgs A = base_args(b, tg::MODE_DERIV2Z);
    int rc = launch(b, A);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(hz_host, b->hz_dev, B * R * R * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_deriv2_contract_lambda(tg_batch *b, const double *z_host, const double *zlambda_host, double *hz_host) {
    if (!b || !hz_host || (!z_host && !zlambda_host)) return fail(TG_ERR_INVALID, "null argument");
    if (b->P.n_true_springs) return fail(TG_ERR_UNSUPPORTED, "V_dqdqdq() is undefined for LinearSpring (as in the reference): no second derivatives");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "Integrator has not solved the next time step yet.");
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch, R = (size_t)b->P.d_nrhs;
    if (int rc0 = ensure_deriv_buffers(b, false, true)) return rc0;
    if (z_host) HIP_TRY(hipMemcpyAsync(b->z_dev, z_host, B * b->P.nX * sizeof(double), hipMemcpyHostToDevice, b->stream));
    else HIP_TRY(hipMemsetAsync(b->z_dev, 0, B * b->P.nX * sizeof(double), b->stream));
    if (zlambda_host && b->P.nc) HIP_TRY(hipMemcpyAsync(b->zl_dev, zlambda_host, B * b->P.nc * sizeof(double), hipMemcpyHostToDevice, b->stream));
    tg::RunArgs A = base_args(b, tg::MODE_DERIV2Z);
    A.zl = (zlambda_host && b->P.nc) ? b->zl_dev : nullptr;
    int rc = launch(b, A);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(hz_host, b->hz_dev, B * R * R * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_dynamics_device(tg_batch *b, const double *q_dev, const double *dq_dev, const double *u_dev, const double *ddqk_dev,
                             double *ddq_dev, double *lambda_dev, int32_t *status_dev) {
    if (!b || !q_dev || !dq_dev || !ddq_dev) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = b->P;
    if ((P.nu && !u_dev) || (P.nk && !ddqk_dev) || (P.nc && !lambda_dev)) return fail(TG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    if (!b->dyn_ints) HIP_TRY(hipMalloc(&b->dyn_ints, 2 * (size_t)b->batch * sizeof(int)));
    tg::RunArgs A = base_args(b, tg::MODE_DYNAMICS);
    // the state is an argument of this call: nothing of the integrator (q1, q2, p, lambda1, status) is touched
    A.q1 = A.q2 = const_cast<double *>(q_dev);
    A.u1 = const_cast<double *>(u_dev ? u_dev : b->u1);
    A.dq_in = dq_dev; A.ddqk_in = ddqk_dev; A.ddq_out = ddq_dev; A.lam_out = lambda_dev;
    A.iters = b->dyn_ints; A.status = status_dev ? status_dev : b->dyn_ints + b->batch;
    return launch(b, A);
}

static int lagrangian_host(tg_batch *b, const double *q_host, const double *dq_host, const int32_t *seed1_host, const int32_t *seed2_host, double *first_host, double *second_host) {
    if (!b || !q_host || !dq_host || !first_host || !second_host) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = b->P;
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch, nq = P.nq;
    const size_t in_total = B * (2 * nq + P.nu + P.nk + P.nd + P.nc), out_total = B * nq * (2 + 3 * nq);
    if (!b->dyn) HIP_TRY(hipMalloc(&b->dyn, (in_total ? in_total : 1) * sizeof(double)));
    if (!b->dyn_ints) HIP_TRY(hipMalloc(&b->dyn_ints, 2 * B * sizeof(int)));
    if (!b->lag) HIP_TRY(hipMalloc(&b->lag, (out_total ? out_total : 1) * sizeof(double)));
    double *q = b->dyn, *dq = q + B * nq, *o1 = b->lag, *o2 = o1 + 2 * B * nq;
    HIP_TRY(hipMemcpyAsync(q, q_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(dq, dq_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemsetAsync(b->lag, 0, out_total * sizeof(double), b->stream));     // the kernel accumulates
    tg::RunArgs A = base_args(b, tg::MODE_LAGRANGIAN);
    A.q1 = A.q2 = q; A.dq_in = dq; A.lag1_out = o1; A.lag2_out = o2;
    A.iters = b->dyn_ints; A.status = b->dyn_ints + b->batch;
    if (seed1_host) {
        if (int rc = stage_seeds(b, seed1_host, seed2_host)) return rc;
        A.seed1 = b->seeds; A.seed2 = seed2_host ? b->seeds + B : nullptr;
        if (int rc = launch_forward(b, A, seed2_host ? 2 : 1)) return rc;
    } else if (int rc = launch(b, A)) return rc;
    HIP_TRY(hipMemcpyAsync(first_host, o1, 2 * B * nq * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(second_host, o2, 3 * B * nq * nq * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_lagrangian(tg_batch *b, const double *q_host, const double *dq_host, double *first_host, double *second_host) {
    return lagrangian_host(b, q_host, dq_host, nullptr, nullptr, first_host, second_host);
}
int tg_batch_lagrangian_forward(tg_batch *b, const double *q_host, const double *dq_host, const int32_t *seed1_host, const int32_t *seed2_host,
                                double *first_host, double *second_host) {
    if (!seed1_host) return fail(TG_ERR_INVALID, "null argument");
    return lagrangian_host(b, q_host, dq_host, seed1_host, seed2_host, first_host, second_host);
}

int tg_batch_set_predictor(tg_batch *b, int32_t mode) {
    if (!b || mode < 0 || mode > 1) return fail(TG_ERR_INVALID, "predictor mode must be 0 or 1");
    b->predictor = mode;
    return TG_SUCCESS;
}

int tg_batch_energy(tg_batch *b, const double *q_host, const double *dq_host, double *energy_host) {
    if (!b || !q_host || !dq_host || !energy_host) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = b->P;
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch, nq = P.nq;
    const size_t in_total = B * (2 * nq + P.nu + P.nk + P.nd + P.nc);   // the staging block of the dynamics calls (>= 2 B nq + 2 B)
    if (!b->dyn) HIP_TRY(hipMalloc(&b->dyn, (in_total ? in_total : 1) * sizeof(double)));
    if (!b->dyn_ints) HIP_TRY(hipMalloc(&b->dyn_ints, 2 * B * sizeof(int)));
    if (!b->energy) HIP_TRY(hipMalloc(&b->energy, 2 * B * sizeof(double)));
    double *q = b->dyn, *dq = q + B * nq;
    HIP_TRY(hipMemcpyAsync(q, q_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(dq, dq_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    tg::RunArgs A = base_args(b, tg::MODE_ENERGY);
    A.q1 = A.q2 = q; A.dq_in = dq; A.energy_out = b->energy;
    A.iters = b->dyn_ints; A.status = b->dyn_ints + b->batch;
    if (int rc = launch(b, A)) return rc;
    HIP_TRY(hipMemcpyAsync(energy_host, b->energy, 2 * B * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

static int dyn_deriv1_device(tg_batch *b, const double *q_dev, const double *dq_dev, const double *u_dev, const double *ddqk_dev, const int *seed_dev,
                             double *const out_dev[8], int32_t *status_dev) {
    if (!b || !q_dev || !dq_dev || !out_dev) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = b->P;
    if ((P.nu && !u_dev) || (P.nk && !ddqk_dev)) return fail(TG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    if (!b->dyn_ints) HIP_TRY(hipMalloc(&b->dyn_ints, 2 * (size_t)b->batch * sizeof(int)));
    tg::RunArgs A = base_args(b, tg::MODE_DYN_DERIV1);
    A.q1 = A.q2 = const_cast<double *>(q_dev);
    A.u1 = const_cast<double *>(u_dev ? u_dev : b->u1);
    A.dq_in = dq_dev; A.ddqk_in = ddqk_dev; A.ddq_out = nullptr; A.lam_out = nullptr;
    for (int g = 0; g < 8; g++) A.g1[g] = out_dev[g];
    A.iters = b->dyn_ints; A.status = status_dev ? status_dev : b->dyn_ints + b->batch;
    if (seed_dev) { A.seed1 = seed_dev; return launch_forward(b, A, 1); }
    return launch(b, A);
}
int tg_batch_dynamics_deriv1_device(tg_batch *b, const double *q_dev, const double *dq_dev, const double *u_dev, const double *ddqk_dev,
                                    double *const out_dev[8], int32_t *status_dev) {
    return dyn_deriv1_device(b, q_dev, dq_dev, u_dev, ddqk_dev, nullptr, out_dev, status_dev);
}

static int dyn_deriv1_host(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host, const double *ddqk_host, const int32_t *seed_host,
                           double *f_dq, double *f_ddq, double *f_dddk, double *f_du,
                           double *lambda_dq, double *lambda_ddq, double *lambda_dddk, double *lambda_du, int32_t *status_host) {
    if (!b || !q_host || !dq_host) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = b->P;
    if ((P.nu && !u_host) || (P.nk && !ddqk_host)) return fail(TG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch, nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc;
    const size_t in_total = B * (2 * nq + nu + nk + nd + nc);
    if (!b->dyn) HIP_TRY(hipMalloc(&b->dyn, (in_total ? in_total : 1) * sizeof(double)));
    const size_t rows[4] = {nq, nq, nk, nu};
    size_t out_total = 0;
    for (int g = 0; g < 8; g++) out_total += B * rows[g & 3] * (g < 4 ? nd : nc);
    if (!b->dyn_d1) HIP_TRY(hipMalloc(&b->dyn_d1, (out_total ? out_total : 1) * sizeof(double)));
    double *q = b->dyn, *dq = q + B * nq, *u = dq + B * nq, *ddk = u + B * nu;
    HIP_TRY(hipMemcpyAsync(q, q_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(dq, dq_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (nu) HIP_TRY(hipMemcpyAsync(u, u_host, B * nu * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (nk) HIP_TRY(hipMemcpyAsync(ddk, ddqk_host, B * nk * sizeof(double), hipMemcpyHostToDevice, b->stream));
    double *host[8] = {f_dq, f_ddq, f_dddk, f_du, lambda_dq, lambda_ddq, lambda_dddk, lambda_du};
    double *dev[8];
    size_t off = 0, cnt[8];
    for (int g = 0; g < 8; g++) {
        cnt[g] = B * rows[g & 3] * (g < 4 ? nd : nc);
        dev[g] = (host[g] && cnt[g]) ? b->dyn_d1 + off : nullptr;
        off += cnt[g];
    }
    if (seed_host) { if (int rc = stage_seeds(b, seed_host, nullptr)) return rc; }
    if (int rc = dyn_deriv1_device(b, q, dq, nu ? u : nullptr, nk ? ddk : nullptr, seed_host ? b->seeds : nullptr, dev, nullptr)) return rc;
    for (int g = 0; g < 8; g++)
        if (dev[g]) HIP_TRY(hipMemcpyAsync(host[g], dev[g], cnt[g] * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (status_host) HIP_TRY(hipMemcpyAsync(status_host, b->dyn_ints + B, B * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}
int tg_batch_dynamics_deriv1(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host, const double *ddqk_host,
                             double *f_dq, double *f_ddq, double *f_dddk, double *f_du,
                             double *lambda_dq, double *lambda_ddq, double *lambda_dddk, double *lambda_du, int32_t *status_host) {
    return dyn_deriv1_host(b, q_host, dq_host, u_host, ddqk_host, nullptr, f_dq, f_ddq, f_dddk, f_du, lambda_dq, lambda_ddq, lambda_dddk, lambda_du, status_host);
}
int tg_batch_dynamics_deriv1_forward(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host, const double *ddqk_host,
                                     const int32_t *seed_host, double *f_dq, double *f_ddq, double *f_dddk, double *f_du,
                                     double *lambda_dq, double *lambda_ddq, double *lambda_dddk, double *lambda_du, int32_t *status_host) {
    if (!seed_host) return fail(TG_ERR_INVALID, "null argument");
    return dyn_deriv1_host(b, q_host, dq_host, u_host, ddqk_host, seed_host, f_dq, f_ddq, f_dddk, f_du, lambda_dq, lambda_ddq, lambda_dddk, lambda_du, status_host);
}

int tg_batch_dynamics(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host, const double *ddqk_host,
                      double *ddq_host, double *lambda_host, int32_t *status_host) {
    if (!b || !q_host || !dq_host || !ddq_host) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = b->P;
    if ((P.nu && !u_host) || (P.nk && !ddqk_host) || (P.nc && !lambda_host)) return fail(TG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    const size_t B = (size_t)b->batch, nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc;
    const size_t total = B * (2 * nq + nu + nk + nd + nc);
    if (!b->dyn) HIP_TRY(hipMalloc(&b->dyn, (total ? total : 1) * sizeof(double)));
    double *q = b->dyn, *dq = q + B * nq, *u = dq + B * nq, *ddk = u + B * nu, *ddq = ddk + B * nk, *lam = ddq + B * nd;
    HIP_TRY(hipMemcpyAsync(q, q_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(dq, dq_host, B * nq * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (nu) HIP_TRY(hipMemcpyAsync(u, u_host, B * nu * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (nk) HIP_TRY(hipMemcpyAsync(ddk, ddqk_host, B * nk * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (int rc = tg_batch_dynamics_device(b, q, dq, nu ? u : nullptr, nk ? ddk : nullptr, ddq, lam, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(ddq_host, ddq, B * nd * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (nc) HIP_TRY(hipMemcpyAsync(lambda_host, lam, B * nc * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (status_host) HIP_TRY(hipMemcpyAsync(status_host, b->dyn_ints + B, B * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_deriv2_contract_device(tg_batch *b, const double *z_dev, double *hz_dev) {
    if (!b || !z_dev || !hz_dev) return fail(TG_ERR_INVALID, "null argument");
    if (b->P.n_true_springs) return fail(TG_ERR_UNSUPPORTED, "V_dqdqdq() is undefined for LinearSpring (as in the reference): no second derivatives");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "Integrator has not solved the next time step yet.");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_DERIV2Z);
    A.z = z_dev; A.hz = hz_dev;
    return launch(b, A);
}

int tg_batch_deriv2_contract_device_range(tg_batch *b, const double *z_dev, double *hz_dev, int32_t horizon, int32_t k_begin, int32_t k_end) {
    if (!b || !z_dev || !hz_dev) return fail(TG_ERR_INVALID, "null argument");
    if (horizon <= 0 || b->batch % horizon != 0 || k_begin < 0 || k_end > horizon || k_begin >= k_end) return fail(TG_ERR_INVALID, "bad step range");
    if (b->P.n_true_springs) return fail(TG_ERR_UNSUPPORTED, "V_dqdqdq() is undefined for LinearSpring (as in the reference): no second derivatives");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "Integrator has not solved the next time step yet.");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_DERIV2Z);
    A.z = z_dev; A.hz = hz_dev;
    A.remap_len = k_end - k_begin; A.remap_stride = horizon; A.remap_off = k_begin; A.remap_count = (b->batch / horizon) * (k_end - k_begin);
    return launch(b, A);
}

int tg_batch_set_from_trajectories(tg_batch *b, int32_t seeds, int32_t horizon, double t0, double dt, const double *X_dev,
                                   const double *U_dev, int32_t max_iterations) {
    if (!b || !X_dev || !U_dev || seeds <= 0 || horizon <= 0 || dt == 0.0) return fail(TG_ERR_INVALID, "bad arguments");
    if ((int64_t)seeds * horizon != b->batch) return fail(TG_ERR_INVALID, "batch size must be seeds * horizon");
    HIP_TRY(hipSetDevice(b->device));
    const tg::DevProg &P = b->P;
    b->mirror_valid = false;
    hipLaunchKernelGGL(k_set_from_trajectories, dim3(b->batch), dim3(64), 0, b->stream, P, seeds, horizon, X_dev, U_dev,
                       b->q1, b->q2, b->p1, b->p2, b->lam, b->stage_u, b->stage_k, b->stage_qh);
    HIP_TRY(hipGetLastError());
    b->t1 = t0; b->t2 = t0;
    tg::RunArgs A = base_args(b, tg::MODE_ROLLOUT);
    A.n_steps = 1; A.dt = dt; A.max_iterations = max_iterations;
    A.U = b->stage_u; A.K = b->stage_k; A.q2_hint = b->stage_qh;
    int rc = launch(b, A);
    if (rc) return rc;
    b->t1 = t0; b->t2 = t0 + dt;
    return TG_SUCCESS;
}

int tg_batch_initialize_from_state_device(tg_batch *b, double t, const double *X_dev, uint64_t row_stride_doubles) {
    if (!b || !X_dev || row_stride_doubles < (uint64_t)(b->P.nq + b->P.nd)) return fail(TG_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    b->mirror_valid = false;
    hipLaunchKernelGGL(k_init_from_X, dim3(b->batch), dim3(64), 0, b->stream, b->P, X_dev, (size_t)row_stride_doubles, b->q1, b->q2,
                       b->p1, b->p2, b->lam);
    HIP_TRY(hipGetLastError());
    b->t1 = t; b->t2 = t;
    return TG_SUCCESS;
}

int tg_batch_linearize(tg_batch *b, double *A_dev, double *B_dev) {
    if (!b || !A_dev || !B_dev) return fail(TG_ERR_INVALID, "null argument");
    if (b->t2 == b->t1) return fail(TG_ERR_STATE, "Integrator has not solved the next time step yet.");
    HIP_TRY(hipSetDevice(b->device));
    tg::RunArgs A = base_args(b, tg::MODE_DERIV1);
    A.A_out = A_dev; A.B_out = B_dev;
    return launch(b, A);
}

void *tg_device_alloc(int32_t device, uint64_t bytes) {
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { fail(TG_ERR_HIP, "hipMalloc failed"); return nullptr; }
    return p;
}
int tg_device_free(int32_t device, void *ptr) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(ptr));
    return TG_SUCCESS;
}
int tg_memcpy_h2d(int32_t device, void *dst_dev, const void *src_host, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return TG_SUCCESS;
}
int tg_memcpy_d2h(int32_t device, void *dst_host, const void *src_dev, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return TG_SUCCESS;
}

int tg_batch_synchronize(tg_batch *b) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TG_SUCCESS;
}

int tg_batch_set_stream(tg_batch *b, void *hip_stream) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->own_stream && b->stream) HIP_TRY(hipStreamDestroy(b->stream));
    if (hip_stream) { b->stream = (hipStream_t)hip_stream; b->own_stream = false; }
    else { HIP_TRY(hipStreamCreate(&b->stream)); b->own_stream = true; }
    return TG_SUCCESS;
}

/* Test hook (tests/test_gpu_parity.py): solves the n x n system [A | b] (row-major [n][n+1], n <= 32) with the register
 * Gauss-Jordan of the rollout kernels and reports which original row was the pivot of each column: the reference's
 * LU_decomp (math-code.c:337-432, implicit scaling, strict `>` scan) must pick the same rows, ties included. */
int tg_debug_solve(int32_t device, int32_t n, int32_t exact, const double *A_aug_host, double *x_host, int32_t *pivot_rows_host, int32_t *status_host) {
    if (n <= 0 || n > 32 || !A_aug_host || !x_host || !pivot_rows_host || !status_host) return fail(TG_ERR_INVALID, "bad arguments");
    if (exact == 2 && (n <= 16 || n >= 32)) return fail(TG_ERR_INVALID, "the panel solver takes 16 < n < 32");
    HIP_TRY(hipSetDevice(device));
    double *dA = nullptr, *dx = nullptr; int *dp = nullptr, *ds = nullptr;
    const int ld = (n + 1) | 1;
    HIP_TRY(hipMalloc(&dA, sizeof(double) * n * (n + 1))); HIP_TRY(hipMalloc(&dx, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dp, sizeof(int) * n)); HIP_TRY(hipMalloc(&ds, sizeof(int)));
    HIP_TRY(hipMemcpy(dA, A_aug_host, sizeof(double) * n * (n + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_solve, dim3(1), dim3(64), sizeof(double) * (n * ld + 16 + 192), 0, n, ld, (int)exact, dA, dx, dp, ds);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(x_host, dx, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pivot_rows_host, dp, sizeof(int) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status_host, ds, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(dp); hipFree(ds);
    return TG_SUCCESS;
}

int tg_batch_set_pivot_rule(tg_batch *b, int32_t exact) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    b->exact_pivot = exact ? 1 : 0;
    return TG_SUCCESS;
}

}  // extern "C"
namespace {
std::string spec_header_text(const tg_system *sys) {
    std::string out;
    char line[160];
    std::snprintf(line, sizeof(line), "#define SPEC_TEAM %d\n#define SPEC_SPRINGS %s\n", sys->team,
                  (sys->H.p.has_cs || sys->H.p.n_springs || sys->H.p.has_plane || sys->H.p.n_wrenches) ? "true" : "false");
    out += line;
    emit_spec_header(sys->H, out);
    return out;
}
uint64_t fnv1a64(const std::string &t) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; }
    return h;
}
}  // namespace
extern "C" {

/* Text of the specialisation header of a system (spec_emit.inc); returns the length needed (incl. the terminator). */
int64_t tg_system_spec_header(const tg_system *sys, char *buf, uint64_t capacity) {
    if (!sys) { fail(TG_ERR_INVALID, "null system"); return -1; }
    const std::string out = spec_header_text(sys);
    if (buf && capacity) {
        const size_t n = std::min((size_t)capacity - 1, out.size());
        std::memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

/* FNV-1a (64 bit) of that text: what a specialised library carries as tg_spec_key() (-DTG_SPEC_KEY=...) and what
 * tg_batch_load_specialized compares, so a library built for another system -- or another parameter set of the same
 * topology -- is refused even when every size agrees. */
uint64_t tg_system_spec_key(const tg_system *sys) {
    if (!sys) { fail(TG_ERR_INVALID, "null system"); return 0; }
    return fnv1a64(spec_header_text(sys));
}

/* Rollouts of this batch use the kernel of a library built by trep_amd/specialize.py for exactly this system. */
int tg_batch_load_specialized(tg_batch *b, const char *library_path) {
    if (!b || !library_path) return fail(TG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    void *h = dlopen(library_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(TG_ERR_INVALID, std::string("cannot load ") + library_path + ": " + (dlerror() ? dlerror() : "?"));
    auto launch_fn = reinterpret_cast<int (*)(int, const tg::RunArgs *, tg::RunArgs *, int, size_t, void *)>(dlsym(h, "tg_spec_launch"));
    auto sizes_fn = reinterpret_cast<const int *(*)(void)>(dlsym(h, "tg_spec_sizes"));
    auto modes_fn = reinterpret_cast<int (*)(void)>(dlsym(h, "tg_spec_modes"));
    auto key_fn = reinterpret_cast<uint64_t (*)(void)>(dlsym(h, "tg_spec_key"));
    if (!launch_fn || !sizes_fn || !modes_fn || !key_fn) { dlclose(h); return fail(TG_ERR_INVALID, "not a specialised trep_amd kernel library"); }
    const tg::DevProg &P = b->P;
    const int want[8] = {(int)sizeof(tg::DevProg), (int)sizeof(tg::RunArgs), P.nq, P.nd, P.nc, P.n_items, P.n_pairs, P.lds_per_team};
    const int *got = sizes_fn();
    for (int i = 0; i < 8; i++) if (got[i] != want[i]) { dlclose(h); return fail(TG_ERR_INVALID, "specialised kernel was built for a different system or library version"); }
    if (key_fn() != tg_system_spec_key(b->sys)) { dlclose(h); return fail(TG_ERR_INVALID, "specialised kernel was built from a different schedule (header hash mismatch)"); }
    if (!b->d_args) {
        bool ok = hipMalloc(&b->d_args, sizeof(tg::RunArgs) * tg_batch::ARG_SLOTS) == hipSuccess &&
                  hipHostMalloc(&b->h_args, sizeof(tg::RunArgs) * tg_batch::ARG_SLOTS, hipHostMallocDefault) == hipSuccess;
        int made = 0;
        for (; made < tg_batch::ARG_SLOTS && ok; made++) ok = hipEventCreateWithFlags(&b->arg_done[made], hipEventDisableTiming) == hipSuccess;
        if (!ok) {      // leave nothing half-made behind: a retry must find the batch as it was
            for (int i = 0; i + 1 < made; i++) { hipEventDestroy(b->arg_done[i]); b->arg_done[i] = nullptr; }
            if (b->h_args) { hipHostFree(b->h_args); b->h_args = nullptr; }
            if (b->d_args) { hipFree(b->d_args); b->d_args = nullptr; }
            dlclose(h);
            return fail(TG_ERR_HIP, "allocation of the argument blocks failed");
        }
    }
    if (b->spec_lib) { hipStreamSynchronize(b->stream); dlclose(b->spec_lib); }
    b->spec_lib = h; b->spec_launch = launch_fn; b->spec_modes = modes_fn(); b->spec_path = library_path;
    auto waves_fn = reinterpret_cast<int (*)(void)>(dlsym(h, "tg_spec_waves"));
    b->spec_waves = waves_fn ? waves_fn() : 1;
    return TG_SUCCESS;
}

/* The structured-solve plan of a system (bbd.hpp; host only): out[0..7] = plan found, groups, largest own block, largest border
 * list, trailing size, nf, nd, 0; pattern (optional, [nf * nf]) the structural non-zeros of the Newton matrix the plan was derived
 * from (symmetrised; all zero if the system is outside the plan's range); tab (optional, [128]) the packed plan tables. */
int tg_system_newton_plan(const tg_system *sys, int32_t out[8], uint8_t *pattern, int32_t *tab) {
    if (!sys || !out) return fail(TG_ERR_INVALID, "null argument");
    const tg::DevProg &P = sys->H.p;
    out[0] = P.bbd_ok; out[1] = P.bbd_g; out[2] = P.bbd_ng; out[3] = P.bbd_nb; out[4] = P.bbd_t; out[5] = P.nf; out[6] = P.nd; out[7] = 0;
    if (pattern) {
        std::memset(pattern, 0, (size_t)P.nf * P.nf);
        if (sys->H.newton_pattern.size() == (size_t)P.nf * P.nf) std::memcpy(pattern, sys->H.newton_pattern.data(), sys->H.newton_pattern.size());
    }
    if (tab) for (int i = 0; i < 128; i++) tab[i] = i < (int)sys->H.bbd_tab.size() ? sys->H.bbd_tab[i] : 0;
    return TG_SUCCESS;
}

/* Test hook (tests/test_gpu_parity.py): the Newton-system solve of the batch's SPECIALISED rollout kernel on caller-supplied
 * systems [n_mats][nf][nf + 1] (nf = the system's unknowns): the structured solve along the system's plan if it has one (bbd.hpp),
 * the pivoting solver when a pivot guard fails or skip_structured is set.  path[m]: 1 structured, 2 pivoting, -1 singular. */
int tg_batch_debug_newton_solve(tg_batch *b, int32_t n_mats, int32_t skip_structured, const double *A_aug_host, double *x_host, int32_t *path_host) {
    if (!b || n_mats <= 0 || !A_aug_host || !x_host || !path_host) return fail(TG_ERR_INVALID, "bad arguments");
    if (!b->spec_lib) return fail(TG_ERR_INVALID, "no specialised kernel library loaded");
    auto fn = reinterpret_cast<int (*)(const double *, double *, int *, int, int)>(dlsym(b->spec_lib, "tg_spec_debug_solve"));
    if (!fn) return fail(TG_ERR_INVALID, "the specialised library has no solve hook");
    const int nf = b->P.nf;
    HIP_TRY(hipSetDevice(b->device));
    double *dA = nullptr, *dx = nullptr; int *dp = nullptr;
    const size_t nA = sizeof(double) * (size_t)n_mats * nf * (nf + 1), nx = sizeof(double) * (size_t)n_mats * nf, np = sizeof(int) * (size_t)n_mats;
    const char *what = nullptr;         /* one exit: whatever was allocated is freed on every path */
    int rc = 0;
    if (hipMalloc(&dA, nA) != hipSuccess || hipMalloc(&dx, nx) != hipSuccess || hipMalloc(&dp, np) != hipSuccess) what = "solve hook: device allocation failed";
    if (!what && hipMemcpy(dA, A_aug_host, nA, hipMemcpyHostToDevice) != hipSuccess) what = "solve hook: upload failed";
    if (!what) {
        rc = fn(dA, dx, dp, n_mats, skip_structured);
        if (hipDeviceSynchronize() != hipSuccess || rc != 0) what = "solve hook launch failed";
    }
    if (!what && (hipMemcpy(x_host, dx, nx, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(path_host, dp, np, hipMemcpyDeviceToHost) != hipSuccess))
        what = "solve hook: download failed";
    if (dA) hipFree(dA);
    if (dx) hipFree(dx);
    if (dp) hipFree(dp);
    if (what) { (void)hipGetLastError(); return fail(TG_ERR_HIP, what); }
    return TG_SUCCESS;
}

/* Which kernels this batch runs: out[0] bit m = mode m (tg::MODE_*) has a specialised kernel loaded; out[1] / out[2] bit m = a
 * mode-m launch has gone through a specialised / a generic kernel since the batch was created; out[3] / out[4] the number of
 * such launches; out[5] pivot rule; out[6] team size; out[7] 0. */
int tg_batch_info(const tg_batch *b, int32_t out[8]) {
    if (!b || !out) return fail(TG_ERR_INVALID, "null argument");
    out[0] = b->spec_launch ? b->spec_modes : 0;
    out[1] = (int32_t)b->spec_launched_modes; out[2] = (int32_t)b->generic_launched_modes;
    out[3] = (int32_t)std::min<long long>(b->spec_launches, 0x7fffffff); out[4] = (int32_t)std::min<long long>(b->generic_launches, 0x7fffffff);
    out[5] = b->exact_pivot; out[6] = b->sys->team; out[7] = b->spec_launch ? b->spec_waves : 1;
    return TG_SUCCESS;
}

/* The HIP stream the batch launches on (hipStream_t as void *): for ordering foreign work after it (tg_comm_wait_stream). */
void *tg_batch_stream(tg_batch *b) { return b ? (void *)b->stream : nullptr; }

int tg_batch_timing(tg_batch *b, int32_t reset, int32_t *n_launches, double *total_ms) {
    if (!b) return fail(TG_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->timing = true;
    double ms = b->folded_ms;
    for (auto &e : b->events) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e.first, e.second));
        ms += t;
    }
    if (n_launches) *n_launches = (int32_t)(b->events.size() + b->folded_n);
    if (total_ms) *total_ms = ms;
    if (reset) {
        for (auto &e : b->events) { b->pool.push_back(e.first); b->pool.push_back(e.second); }
        b->events.clear();
        b->folded_ms = 0.0; b->folded_n = 0;
    }
    return TG_SUCCESS;
}

}  // extern "C"
