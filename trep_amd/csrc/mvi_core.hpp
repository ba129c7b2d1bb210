// mvi_core.hpp -- the batched MidpointVI DEL step, written once for the device.
//
// One *team* of TEAM lanes (TEAM = 64: one wavefront per trajectory; smaller teams for small
// systems) advances one trajectory; all of a trajectory's working data lives in the team's LDS
// slice (layout: DevProg::o_*).  Work is organised as phases: inside a phase every lane processes
// independent items `for (i = lane; i < n; i += TEAM)`, phases are separated by TG_SYNC().
//
// Math (equal to the reference's cached-table algorithm, verified to 1e-15 against
// the oracle): world poses G_j of the variable frames by a level-ordered sweep; body Jacobian
// columns J_{F,k} = Ad_{g_F}^{-1} s_k; v_F = sum_k J_k dq_k; with P_j = sum_{k<j} J_k dq_k and
// [a,b] = ad_a b:   dv/dq_j = W_j = [P_j, J_j],   dJ_i/dq_j = [J_i, J_j] (i<j, else 0),
//                   d2v/dq_i dq_j = [W_i, J_j] (i<=j).
// These give L_dq, L_ddq, L_dqdq, L_ddqdq, L_ddqddq (reference system.c:129-557) and with them the
// DEL residual and Newton matrix of midpointvi.c:533-670.  Constraint values/Jacobians use
// dp_E/dq_k = w_k x (p_E - p_k) (rotary joint) or the joint axis (prismatic).
//
// The same source is compiled (a) by hipcc for gfx950 (trepamd.hip) and (b) by g++ with TEAM=1 as a
// host emulation used ONLY by the CPU test-suite to check the kernel logic where no GPU exists
// (tests/emu).  The product library contains no CPU path.
#pragma once
#include <cmath>

#include "program.hpp"
#include "dual.hpp"

#if defined(__HIPCC__)
#define TG_HD __host__ __device__ __forceinline__
// Phase boundary.  Lanes of a team exchange data through LDS only, so the fence is restricted to the LDS address
// space: a plain __syncthreads() also waits for every outstanding GLOBAL store (s_waitcnt vmcnt(0)), which puts
// the HBM write latency of the result rows on the critical path of the next phase.
// Helper waves (-DTG_HELPER_WAVES=n, system-specialised builds of full-wave teams): the second-derivative kernel runs n wavefronts per
// trajectory.  Wave 0 owns every wave-scoped phase (sweeps, register solvers, DPP searches); the flat pair / tile loops -- which
// only read LDS tables and accumulate with LDS atomics -- are shared by all n waves (TG_FORW) between workgroup barriers (TG_WSYNC).
// A phase boundary INSIDE wave 0's part must then not be a workgroup barrier: TG_SYNC becomes a wave-local fence (in a one-wave
// workgroup that is all s_barrier ever was).
#if defined(__HIP_DEVICE_COMPILE__)
#define TG_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#if defined(TG_HELPER_WAVES)
#define TG_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#else
#define TG_SYNC() TG_WSYNC()
#endif
#else
#define TG_SYNC() ((void)0)
#define TG_WSYNC() ((void)0)
#endif
#else
#define TG_HD inline
#define TG_SYNC() ((void)0)
#define TG_WSYNC() ((void)0)
#endif
#if defined(TG_HELPER_WAVES)
constexpr int TG_NW = TG_HELPER_WAVES;
#else
constexpr int TG_NW = 1;
#endif

#if defined(__HIPCC__)
// The LDS slice of a workgroup.  The kernels of this library have no static LDS, so their dynamic LDS starts at LDS address 0.
// Saying so -- instead of going through the `extern __shared__` symbol, whose address is only known at link time -- makes every
// "LDS base + offset" a plain number: with the schedule compiled in (spec_kernel.hip) ~50 loop-invariant addresses fold into the
// offset fields of the ds_* instructions and stop occupying scalar registers (-7 % rollout time), in the generic kernels one add
// per address goes away.  A literal 0 would be the null pointer, which is -1 in this address space: hence 8 - 1 element.  The
// symbol's real address is checked once per kernel (a workgroup-uniform compare) and the kernel traps if it is not 0.
__device__ __forceinline__ double *tg_lds_base() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) double lds_double;
    extern __shared__ double tg_dynamic_lds[];
    if ((__UINTPTR_TYPE__)(lds_double *)tg_dynamic_lds != 0) __builtin_trap();
    return (double *)((lds_double *)(__UINTPTR_TYPE__)8 - 1);
#else
    return nullptr;
#endif
}
#endif

// Which trajectory a workgroup of a rollout launch takes.  The dispatcher deals workgroups round-robin to the eight XCDs (workgroup b runs on
// XCD b % 8) and, inside an XCD, round-robin to its 32 CUs.  With the identity mapping and eight Armijo candidates per seed, candidate j of
// EVERY seed therefore ran on XCD j, and inside an XCD candidate j sat on the same few CUs: the large steps fail after a few dozen DEL steps,
// the small ones run the whole horizon, so three XCDs (then: 12 of 32 CUs) did all the full-length projections at full occupancy while the
// rest of the chip idled.  tg_xcd_block transposes twice: XCD x takes the x-th CONTIGUOUS eighth of the blocks, and the i-th workgroup of an
// XCD (CU i % 32, its (i / 32)-th resident workgroup) the (i / 32)-th block of that CU's contiguous share -- the eight workgroups resident on
// a CU are eight NEIGHBOURING trajectories (all candidates of one seed: they also share its gain schedule and reference in that CU's L1 and
// that XCD's L2).  A bijection on [0, G) for every G.
TG_HD int tg_xcd_block(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7, i = b >> 3;
    const int n = q + (x < r ? 1 : 0);                      // blocks of this XCD
    const int q2 = n >> 5, r2 = n & 31, u = i & 31, s_ = i >> 5;
    return x * q + (x < r ? x : r) + u * q2 + (u < r2 ? u : r2) + s_;
}

// The lane index is laundered through an empty asm at the head of every phase loop: the optimiser then
// cannot hoist lane-derived addresses and table look-ups of ALL phases out of the Newton / step loops
// (which made it keep hundreds of loop-invariant values alive and spill).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int tg_opaque(int x) { asm volatile("" : "+v"(x)); return x; }
#else
inline int tg_opaque(int x) { return x; }
#endif
#ifndef TG_CHAIN_PRIO
#define TG_CHAIN_PRIO 2     // wave priority inside the two longest serial chains (structured solve, chain rounds): the wave that is on one wins
                            // the SIMD's issue slots against a neighbour in a wide phase (-0.8 %; 3 and a third raised phase measured: no better)
#endif
#ifndef TG_PREFIX_GROUP
#define TG_PREFIX_GROUP 5
#endif
#ifndef TG_LT_TRIPS
#define TG_LT_TRIPS 4
#endif
#define TG_FOR(idx, n) for (int idx = tg_opaque(lane); idx < (n); idx += TEAM)
// the same over all the waves of a trajectory (helper-wave kernels; `wave` is 0 and `nw` 1 everywhere else)
#define TG_FORW(idx, n) for (int idx = tg_opaque(lane + TEAM * wave); idx < (n); idx += TEAM * nw)

#if defined(__HIPCC__)
// ROCm device-library wavefront reduction (DPP based); declared in hip/amd_detail only behind an opt-in macro
extern "C" __device__ __attribute__((const)) unsigned long long __ockl_wfred_max_u64(unsigned long long);
extern "C" __device__ __attribute__((const)) unsigned int __ockl_wfred_max_u32(unsigned int);
#endif

// Diagnostic build only (-DTG_PROFILE, `make prof`): per-phase cycle accumulation with s_memtime.
// Never enabled in the product library; the stamps never feed an output value.
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define TG_STAMP(id) do { long long t_ = (long long)__builtin_amdgcn_s_memtime(); prof[id] += t_ - prof_last; prof_last = t_; } while (0)
#ifndef TG_PROF_TRAJ
#define TG_PROF_TRAJ 0      // which trajectory of a rollout launch reports its counters
#endif
#else
#define TG_STAMP(id) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// 1/p to full precision: hardware seed (4.6e-8 relative, tools/micro/rcp_f64_accuracy.hip) and ONE cubic refinement
// r (1 + e + e^2), e = 1 - p r: three dependent fp64 operations instead of the four of two Newton steps (a dependent fp64
// operation costs ~30 cycles on this part); the error is e^3 ~ 1e-22 plus rounding.
__device__ __forceinline__ double tg_rcp(double p) {
    const double r = __builtin_amdgcn_rcp(p);
    const double e = fma(-p, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
// max over lanes 0..31 of a wavefront (the register solvers hold at most 32 rows): four row-shift steps leave each 16-lane
// row's maximum in its last lane; the two row maxima are combined on the scalar unit.  Two DPP steps shorter than the
// library's full-wave reduction, and this sits on the critical path of every pivot step.
__device__ __forceinline__ unsigned int tg_max_u32_lanes32(unsigned int v) {
    unsigned int t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); v = t > v ? t : v;   // row_shr:1
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); v = t > v ? t : v;   // row_shr:2
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); v = t > v ? t : v;   // row_shr:4
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); v = t > v ? t : v;   // row_shr:8
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 15), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 31);
    return a > b ? a : b;
}
#endif
#include <type_traits>
#include "bbd_solve.hpp"

namespace tg {
// the structured Newton solve needs its plan as compile-time constants: system-specialised schedules (SpecProg: static members) only
template <class P, class = void> struct tg_static_bbd { static constexpr bool value = false; };
template <class P> struct tg_static_bbd<P, typename std::enable_if<(P::bbd_ok >= 0)>::type> { static constexpr bool value = P::bbd_ok != 0; };
// number of (own + border) columns of the plan (1 where there is no plan: the type of an unused variable)
template <bool USE, class P, class = void> struct tg_static_bbd_cols { static constexpr int value = 1; };
template <class P> struct tg_static_bbd_cols<true, P, typename std::enable_if<(P::bbd_ok > 0)>::type> { static constexpr int value = P::bbd_ng + P::bbd_nb; };
// lane K of every quad (four neighbouring lanes) to the whole quad: two 32-bit DPP moves (quad_perm has no 64-bit form)
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL> __device__ __forceinline__ double tg_dpp_f64(double x) {      // any DPP control on a double (lanes without a source read 0)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tg_readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <int K> __device__ __forceinline__ double tg_quad_bcast(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, K * 0x55, 0xF, 0xF, true);     // (bound_ctrl: no tied `old` operand, hence no copy ahead of the move)
    hi = __builtin_amdgcn_update_dpp(0, hi, K * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#endif
// any system-specialised schedule (sizes are static members)
template <class P, class = void> struct tg_is_spec { static constexpr bool value = false; };
template <class P> struct tg_is_spec<P, typename std::enable_if<(P::nq >= 0)>::type> { static constexpr bool value = true; };
// ... whose item areas J and W start on an even double (16-byte LDS accesses of an item's six doubles)
template <class P, class = void> struct tg_items_aligned { static constexpr bool value = false; };
template <class P> struct tg_items_aligned<P, typename std::enable_if<(P::nq >= 0)>::type> { static constexpr bool value = (P::o_J % 2 == 0) && (P::o_W % 2 == 0); };
// ... whose chain schedule has a quad-lane sweep plan (program.hpp, sw_*)
template <class P, class = void> struct tg_static_sweep { static constexpr bool value = false; };
template <class P> struct tg_static_sweep<P, typename std::enable_if<(P::sw_ok > 0)>::type> { static constexpr bool value = true; };
// ... and so does the composite assembly of the Newton matrix (its group sums are unrolled over compile-time membership masks)
template <class P, class = void> struct tg_static_cmp { static constexpr bool value = false; };
template <class P> struct tg_static_cmp<P, typename std::enable_if<(P::cmp_ok >= 0)>::type> { static constexpr bool value = P::cmp_ok != 0 && P::tab_ok != 0; };

// ... and the world-frame evaluation of the rollout's residual (program.hpp, wev_*; -DTG_NO_WEV keeps the (body, config) item phases)
template <class P, class = void> struct tg_static_wev { static constexpr bool value = false; };
#if !defined(TG_NO_WEV) && !defined(TG_NO_CMP)
template <class P> struct tg_static_wev<P, typename std::enable_if<(P::wev_ok >= 0)>::type> { static constexpr bool value = P::wev_ok != 0 && P::cmp_ok != 0 && P::tab_ok != 0; };
#endif

// ... with the Newton image in the structured solve's own order (program.hpp, bbd_pk_*): OPT-IN, -DTG_BBD_PACKED.  Built as the round-4
// verdict asked (rows as 16-byte runs, no gathers, no selects on the loads: the solver shrinks from 901 to 703 instructions) and measured
// against the dense image on one box: 30.33 against 30.07 ms at a 12-double row stride, 30.44 against 30.15 at the conflict-free 14 --
// one percent SLOWER either way, so the dense image stays the default (tests/test_gpu_parity.py builds and checks the packed variant).
template <class P, class = void> struct tg_static_pk { static constexpr bool value = false; };
#if defined(TG_BBD_PACKED) && !defined(TG_NO_BBD)
template <class P> struct tg_static_pk<P, typename std::enable_if<(P::bbd_pk_ok >= 0)>::type> { static constexpr bool value = P::bbd_pk_ok != 0 && tg_static_wev<P>::value && tg_static_bbd<P>::value; };
#endif

enum { MODE_ROLLOUT = 0, MODE_CALC_P2 = 1, MODE_CALC_F = 2, MODE_DERIV1 = 3, MODE_DERIV2Z = 4, MODE_DYNAMICS = 5, MODE_DYN_DERIV1 = 6, MODE_ENERGY = 7, MODE_LAGRANGIAN = 8 };

struct RunArgs {
    int batch, n_steps, max_iterations, mode;
    int predictor;                         // rollout: 0 = the reference's initial guess q2 <- previous q2 (midpointvi.py:188-197), 1 = q2 + (q2 - q1)
    double dt, t1, t2, tolerance;
    double *q1, *q2, *p1, *p2, *lam, *u1;  // batch state, row-major [batch][width]
    const double *U, *K;                   // [batch][n_steps][nu], [batch][n_steps][nk]
    const double *q2_hint, *lam_hint;      // [batch][nd], [batch][nc] or null
    // closed-loop rollout (projection operator, reference dsystem.py:426-451, doptimizer.py:405-428):
    //   U_k = bU_k - Kproj_k (X_k - bX_k);  Kproj [groups][n_steps][nU][nX], one group per `group_size`
    //   consecutive trajectories; bX [batch][n_steps+1][nX]; bU [batch][n_steps][nU]; Uout like bU or null
    const double *Kproj, *bX, *bU;
    double *Uout;
    int group_size;
    const int *group_map;                  // optional [groups]: gain schedule of group g is Kproj[group_map[g]]
    double *X;                             // [batch][n_steps+1][nX] or null
    double *f_out;                         // MODE_CALC_F: [batch][nf]
    double *d1[12];                        // MODE_DERIV1 outputs q2_d{q1,p1,u1,k2}, p2_d*, l1_d*: [batch][var][out]
    double *A_out, *B_out;                 // MODE_DERIV1, if A_out != null: write DSystem.fdx / fdu instead
                                           // (dsystem.py:284-317): A [batch][nX][nX], B [batch][nX][nU]
    const double *z;                       // MODE_DERIV2Z: [batch][nX] contraction vector
    double *hz;                            // MODE_DERIV2Z: [batch][R][R], R = nq+nd+nu+nk
    int *iters, *status;                   // [batch]
    long long *prof_out;                   // diagnostic build: [16] cycle counters of trajectory 0
    const double *dt_steps;                // optional non-uniform time base: step sizes; rollout: dt of step k = dt_steps[k]; with
    int dt_period;                         // dt_period > 0 (one step per trajectory, batch = seeds x horizon): dt of trajectory t = dt_steps[t % dt_period]
    int exact_pivot;                       // rollout / step: 1 = the reference's pivot sequence bit for bit (gj_rows_exact), 0 = single-precision ranking
    const double *zl;                      // MODE_DERIV2Z, optional: [batch][nc] weights of the lambda1 second derivatives
    const double *dq_in, *ddqk_in;         // MODE_DYNAMICS: rates [batch][nq] and kinematic accelerations [batch][nk] (q in q1 = q2, u in u1)
    double *ddq_out, *lam_out;             // MODE_DYNAMICS: accelerations of the dynamic configs [batch][nd], constraint forces [batch][nc]
    double *g1[8];                         // MODE_DYN_DERIV1: f_dq, f_ddq [batch][nq][nd], f_dk [batch][nk][nd], f_du [batch][nu][nd], then the
                                           // same four for lambda ([..][nc]); derivative variable first, like the reference's arrays
    double *energy_out;                    // MODE_ENERGY: [batch][2] kinetic and potential energy at (q, dq_in)
    double *lag1_out, *lag2_out;           // MODE_LAGRANGIAN: [batch][2][nq] (L_dq, L_ddq) and [batch][3][nq][nq] (L_dqdq, L_ddqdq, L_ddqddq), zeroed by the caller
    double *mirror;                        // rollout, optional: host-visible copy of the final state, (q2 [batch][nq] | p2 [batch][nd] | lambda1 [batch][nc] |
                                           // int32 iterations [batch] | int32 status [batch]) -- pinned host memory written by the kernel (tg_batch_step)
    // launch over a SUBSET of the batch's trajectories (the k-chunks of the pipelined discopt Newton step): workgroup-trajectory i < remap_count
    // is trajectory (i / remap_len) * remap_stride + remap_off + i % remap_len of the batch; remap_len = 0: the identity
    int remap_len, remap_stride, remap_off, remap_count;
    int *fallbacks;                        // rollout / step, optional: [batch] how many Newton systems of this launch the structured solve handed to the pivoting solver
                                           // (a failed pivot guard: correct, but slower than either solver alone)
    // forward-mode kernels (run_forward): per trajectory the input variable each direction follows, numbered q [nq] | dq [nq] | ddq_k [nk] | u [nu]
    // (-1: none); seed2 null for first-order kernels.  The outputs (g1 / ddq_out / lam_out / lag1_out / lag2_out / energy_out) then receive the
    // derivative of the quantity along seed1 (and seed2) instead of the quantity.
    const int *seed1, *seed2;
};
// which trajectory of the batch a launch's i-th trajectory is (RunArgs::remap_*); A.batch for an index past the subset (an idle team)
template <class ARGS> TG_HD int tg_remap_trajectory(const ARGS &A, int i) {
    if (A.remap_len <= 0) return i;
    return i < A.remap_count ? (i / A.remap_len) * A.remap_stride + A.remap_off + i % A.remap_len : A.batch;
}

// The kernels read the schedule (DevProg) and the launch arguments (RunArgs) through CONSTANT-address-space references:
// every field access is then a scalar load from the kernel-argument segment.  Left alone the optimiser hoists all of those
// loop-invariant loads to the kernel prologue and keeps ~150 values alive in SGPRs for the whole rollout -- far more than
// the 100 or so there are, so it spills them into VGPR lanes and re-reads them with v_readlane at every use (1 267
// v_readlane + 372 v_writelane in the round-1 rollout kernel, next to 820 fp64 instructions).  tg_fresh() launders the
// struct's address through an empty asm at the head of every phase: the loads cannot move above it, a phase loads the few
// fields it needs when it starts (K$-resident, issued back to back) and nothing stays alive across phases.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) DevProg CProg;
typedef const RunArgs CArgs;   // by-value kernel argument (few fields are used inside the loops)
template <class T> __device__ __forceinline__ T &tg_fresh_always(T &r) { T *p = &r; asm volatile("" : "+s"(p)); return *p; }
#if defined(TG_FRESH_PHASES)
template <class T> __device__ __forceinline__ T &tg_fresh(T &r) { return tg_fresh_always(r); }
#else
template <class T> __device__ __forceinline__ T &tg_fresh(T &r) { return r; }
#endif
#if defined(TG_FRESH_STEP)
template <class T> __device__ __forceinline__ T &tg_fresh_step(T &r) { return tg_fresh_always(r); }
#else
template <class T> __device__ __forceinline__ T &tg_fresh_step(T &r) { return r; }
#endif
// launch arguments read through a constant-address-space reference (specialised kernel: RunArgs in device memory)
typedef const __attribute__((address_space(4))) RunArgs KArgs;
__device__ __forceinline__ KArgs &tg_fresh_args(KArgs &r) { return tg_fresh_always(r); }
template <class T> __device__ __forceinline__ T &tg_fresh_args(T &r) { return r; }
#else
typedef const DevProg CProg;
typedef const RunArgs CArgs;
typedef const RunArgs KArgs;
template <class T> inline T &tg_fresh_args(T &r) { return r; }
template <class T> inline T &tg_fresh(T &r) { return r; }
template <class T> inline T &tg_fresh_step(T &r) { return r; }
#endif

// sin and cos together for joint angles.  |x| < 2^17: three-constant Cody-Waite reduction to [-pi/4, pi/4]
// (k * pi/2 split so that the first two products are exact for k < 2^19) and the minimax polynomials of the
// classic fdlibm kernels (error < 1 ulp); larger arguments take the library routine.  About a fifth of the
// instructions of the library sincos, which matters here: one evaluation per rotary joint per pose sweep.
TG_HD void tg_sincos(double x, double *s, double *c) {
    if (!(fabs(x) < 131072.0)) {
#if defined(__HIP_DEVICE_COMPILE__)
        sincos(x, s, c);
#else
        *s = std::sin(x);
        *c = std::cos(x);
#endif
        return;
    }
    const double kf = rint(x * 6.36619772367581382433e-01);          // x * 2/pi
    double r = fma(-kf, 1.57079632673412561417e+00, x);               // pi/2, high 33 bits
    r = fma(-kf, 6.07710050630396597660e-11, r);                      // next 33 bits
    const double t = fma(-kf, 2.02226624871116645580e-21, r);         // third 33 bits
    const double z = t * t;
    // sin(t), cos(t) on [-pi/4, pi/4]
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                      2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double sn = fma(t * z, ps, t);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                      -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)kf & 3;
    const double ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

template <class T> TG_HD void tg_sincos(const tgdual::Dual<T> &x, tgdual::Dual<T> *s, tgdual::Dual<T> *c) {
    T sv, cv;
    tg_sincos(x.v, &sv, &cv);
    *s = tgdual::Dual<T>(sv, cv * x.d);
    *c = tgdual::Dual<T>(cv, -(sv * x.d));
}
// [a,b] = ad_a b for twists stored (v, w)
template <class RA, class RB, class RR> TG_HD void bracket(const RA *a, const RB *b, RR *r) {
    r[0] = a[4] * b[2] - a[5] * b[1] + a[1] * b[5] - a[2] * b[4];
    r[1] = a[5] * b[0] - a[3] * b[2] + a[2] * b[3] - a[0] * b[5];
    r[2] = a[3] * b[1] - a[4] * b[0] + a[0] * b[4] - a[1] * b[3];
    r[3] = a[4] * b[5] - a[5] * b[4];
    r[4] = a[5] * b[3] - a[3] * b[5];
    r[5] = a[3] * b[4] - a[4] * b[3];
}
// accumulate into LDS from several lanes at once
template <class T, class V> TG_HD void lds_add(tgdual::Dual<T> *p, const V &v);
TG_HD void lds_add(double *p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p += v;
#endif
}
// accumulate into a global output from several lanes / bodies
template <class T, class V> TG_HD void lds_add(tgdual::Dual<T> *p, const V &v) {   // (forward-mode scalars: part by part)
    const tgdual::Dual<T> x(v);
    lds_add(&p->v, x.v); lds_add(&p->d, x.d);
}
TG_HD void gl_add(double *p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
template <class RI, class RA, class RB> TG_HD auto inner6(const RI *I, const RA *a, const RB *b) -> decltype(I[0] * (a[0] * b[0])) {
    return I[0] * (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) + I[1] * (a[3] * b[3]) + I[2] * (a[4] * b[4]) +
           I[3] * (a[5] * b[5]);
}

// smallest c with 2^c >= cols, capped at log2(TEAM)
template <int TEAM>
TG_HD int tile_log2(int cols) {
    int c = 0;
    while ((1 << c) < cols && (1 << c) < TEAM) c++;
    return c;
}

// arg-max over the team; ties resolve to the smaller index (first maximum, as a serial scan finds)
template <int TEAM>
TG_HD void team_argmax(double &v, int &i) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int m = TEAM / 2; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(v, m, TEAM);
        const int oi = __shfl_xor(i, m, TEAM);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
#else
    (void)v; (void)i;
#endif
}

// SPRINGS: the spring potentials (ConfigSpring, LinearSpring) and the plane constraints are compiled in only for systems that have them, so
// that the spring-free kernels keep their instruction stream and register allocation.
template <int V> struct IntTag { static constexpr int value = V; };

// Real: the scalar the trajectory's LDS slice holds -- double everywhere except the forward-mode kernels of the continuous dynamics (dual.hpp),
// which run dynamics / dyn_deriv1 / lagrangian and what those call on Dual numbers; every other member is only ever instantiated for double.
template <int TEAM, bool SPRINGS = false, class PROG = CProg, class Real = double>
struct Core {
    TG_HD bool has_cs() const { return SPRINGS && P.has_cs; }
    // Potentials on a single config: d1 = V_dq, d2 = V_dqdq, d3 = V_dqdqdq of config i at the value q.  ConfigSpring
    // (configspring.c:22-45): k (q - q0), k, 0, summed per config on the host.  NonlinearConfigSpring
    // (nonlinear_config_spring.c:24-61): -y(x), -y'(x) m, +y''(x) m^2 with x = m q + b and y the piecewise quintic of
    // trep/spline.py (table rows: left knot, a..f; piece = number of later pieces whose left knot is <= x, spline.c:8-22).
    // The sign of the third derivative is the reference's (-ddy * -m * m, :48-58), not the derivative of the second.
    TG_HD void cs_eval(int i, Real q, Real &d1, Real &d2, Real &d3) const {
        d1 = P.cs_k[i] * q - P.cs_kq0[i]; d2 = P.cs_k[i]; d3 = 0.0;
        for (int sp = 0; sp < P.n_ncs; sp++) {
            if (P.ncs_i[3 * sp] != i) continue;
            const Real m = P.ncs_mb[2 * sp], x = m * q + P.ncs_mb[2 * sp + 1];
            const double *tab = P.ncs_tab + 7 * (size_t)P.ncs_i[3 * sp + 1];
            const int pieces = P.ncs_i[3 * sp + 2];
            int seg = 0;
            for (int j = 1; j < pieces; j++) seg += x >= tab[7 * j] ? 1 : 0;
            const double *c = tab + 7 * seg;
            const Real t = x - c[0];
            const Real y = c[1] * t * t * t * t * t + c[2] * t * t * t * t + c[3] * t * t * t + c[4] * t * t + c[5] * t + c[6];
            const Real dy = 5 * c[1] * t * t * t * t + 4 * c[2] * t * t * t + 3 * c[3] * t * t + 2 * c[4] * t + 1 * c[5];
            const Real ddy = 20 * c[1] * t * t * t + 12 * c[2] * t * t + 6 * c[3] * t + 2 * c[4];
            d1 -= y; d2 -= dy * m; d3 += ddy * m * m;
        }
    }
    TG_HD Real cs_d1(int i, Real q) const { Real a, b, c; cs_eval(i, q, a, b, c); return a; }
    TG_HD Real cs_d2(int i, Real q) const { Real a, b, c; cs_eval(i, q, a, b, c); return b; }
    TG_HD Real cs_d3(int i, Real q) const { Real a, b, c; cs_eval(i, q, a, b, c); return c; }
    TG_HD int n_springs() const { return SPRINGS ? P.n_springs : 0; }
    TG_HD int n_spair() const { return SPRINGS ? P.n_spair : 0; }
    TG_HD int n_sdh() const { return SPRINGS ? P.n_sdh : 0; }
    TG_HD bool has_plane() const { return SPRINGS && P.has_plane; }   // plane constraints ride on the same switch
    TG_HD int n_wrenches() const { return SPRINGS ? P.n_wrenches : 0; } // ... and the point forces
    TG_HD bool has_damper() const { return SPRINGS && P.has_damper; }     // ... and the linear dampers
    TG_HD int n_wdh() const { return SPRINGS ? P.n_wdh : 0; }
    TG_HD int n_wpair() const { return SPRINGS ? P.n_wpair : 0; }
    const double *d2w = nullptr;   // adjoint weights while the second-derivative kernel evaluates the midpoint, else null
    PROG &P;
    Real *S;
    int lane;
    int seed1 = -1, seed2 = -1;   // forward-mode kernels: the input variables (0 .. 2 nq + nk + nu: q, dq, ddq_k, u) the two directions follow
    TG_HD Real seeded(double v, int var) const { return tgdual::Seed<Real>::make(v, var == seed1, var == seed2); }
    double dt;
    int oGc;   // LDS offset of the joint poses that the end-point / constraint evaluation reads (P.o_G, or the second pose set of a dual sweep)
    int dsA = 0, dsB = 2;   // which configurations the two pose sets of pose_sweep_dual belong to (qval selectors: 0 midpoint, 1 q1, 2 q2)
    // table rows this lane needs in every Newton iteration of a rollout, read once per kernel (P.tab_ok, eval_both_tab):
    // joint (config | kind << 16) of the two sin/cos trips, the body's item range of the velocity prefix sums, and the
    // constant Newton-matrix entries (damping of this row, (constraint, config) of the lane's two Dh items, first pair record)
    int jck[2] = {0, 0}, bio[2] = {0, 0}, tck[2][2] = {{0, 0}, {0, 0}}, tpair[4] = {0, 0, 0, 0};
    int sc_rot2 = 0;   // number of (pose set, rotary joint) items of the dual sin/cos pass: they come first in the lane order
    int sj_n = 0;      // number of (pose set, joint) items of that pass (init_sweep_schedule)
    bool sj_quads = false;   // the quad-lane chain rounds follow the rollout's instance plan (sw_inst): rollout kernels only
    // quad-lane chain rounds, constants of the lane for the whole kernel (init_sweep_schedule): the instance it carries in pass p of round r
    // (swc[4 r + p]: chain slot | pose set << 8; slot 15 = none) and its place in the instance (entry (swr, swc_) of the 3 x 4 pose)
    int swc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, swr = 0, swcol = 0;
    bool swact = false;
    double tdamp = 0.0;
    // world-frame evaluation (eval_world): the lane's row of P.wev_lane and its config-pair records, constants of the lane for the whole
    // kernel; w_k = [V_k^-, s_k] of the lane's config between the evaluation and the Newton matrix
    int wvl[4] = {0, 0, 0, 0}, wpair[4] = {0, 0, 0, 0};
    int wdhx[2] = {0, 0}, wrhs = 0, wone = -1;      // packed Newton image: the lane's two Dh items (n | two addresses), its right-hand-side entry, its identity entry
    double wev_w[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    bool pk_image = false;    // the Newton matrix goes to the packed image (the kernel solves it with the structured solve: default pivot rule only)
    bool wev_on = false;      // the last evaluation was eval_world (rollout kernels): the Newton matrix continues from its LDS / register state
    long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool rates_ready = false;      // o_dq already holds (q2 - q1) / dt when eval_both_tab starts
    double res_f2 = 0.0;           // eval_both_tab: square of the residual entry this lane formed (0 past nd) ...
    bool res_hoff = false;         // ... and whether the constraint this lane evaluated is outside its tolerance: solved_fused()
    long long prof_last = 0;

    bool d1_compact = false;   // first-derivative kernel: the compact slice (DevProg::a_*; run_trajectory sets it for MODE_DERIV1 when the schedule allows)
    int wave = 0, nw = 1;   // helper-wave kernels: index of this wavefront within the trajectory's workgroup, number of waves (uniform)
    // this wave's part [lo, hi) of a two-part pair list (program.hpp, wp_* / wt_* / wcp4): the parts never meet at a table entry
    TG_HD void wave_part(int first, int split, int last, int &lo, int &hi) const {
        lo = first; hi = last;
        if (nw > 1) { lo = wave ? split : first; hi = wave ? last : split; }
    }
    TG_HD Core(PROG &p, Real *s, int l, double dt_) : P(p), S(s), lane(l), dt(dt_), oGc(p.o_G), inv_dt(1.0 / dt_) {}
    // x / dt with the step's reciprocal (inv_dt follows dt): the quotient estimate and one residual correction -- three
    // dependent operations instead of the ten of the division sequence; the correctly rounded quotient (Markstein's final step)
    double inv_dt;
    TG_HD double over_dt(double x) const { const double q = x * inv_dt; return fma(fma(-q, dt, x), inv_dt, q); }

    // Phase loop over n independent items, two per lane and trip: compute(i) only READS and returns its results,
    // store(i, r) writes them.  Both items' loads are issued before either item's stores -- the compiler cannot
    // reorder an LDS load over an earlier LDS store, so a plain two-trip loop waits out the full latency twice.
    template <class Compute, class Store>
    TG_HD void for_pairs(int n, Compute compute, Store store) {
        for (int i0 = tg_opaque(lane); i0 < n; i0 += 2 * TEAM) {
            const int i1 = i0 + TEAM;
            const bool two = i1 < n;
            const auto r0 = compute(i0);
            const auto r1 = compute(two ? i1 : i0);
            store(i0, r0);
            if (two) store(i1, r1);
        }
    }

    // configuration value at the evaluation point: 0 midpoint, 1 q1, 2 q2 (midpointvi.c:401-457)
    TG_HD Real qval(int sel, int c) const {
        Real a = S[P.o_q1 + c], b = S[P.o_q2 + c];
        return sel == 0 ? 0.5 * (b + a) : (sel == 1 ? a : b);
    }

    // ---- world poses of all joints ---------------------------------------------------------------------
    // (1) sin/cos (or the displacement) per joint; (2) every joint's LOCAL transform pre_j * lg(q_j) from
    // host-made coefficient rows (branch-free: entry = A + B*s1 + C*s0), stored where the world pose will be;
    // (3) world poses G_j = G_parent(j) * local_j in place:
    //   * device, full-wave teams: CHAIN sweep.  Joints are numbered chain by chain (program.hpp); a group of four lanes
    //     owns a chain, lane r < 3 carries ROW r of the running pose in registers (row r of a product only needs
    //     row r of the left factor), so the recurrence along a chain is a register-only FMA chain: no cross-lane
    //     traffic, no barrier; the local transforms are LDS reads that do not depend on it.  One barrier per round
    //     of chains (puppet: 2) instead of one per tree level (11).
    //   * otherwise (host emulation, small systems with several trajectories per wave): level by level, one lane
    //     per (joint, column).
    // Chain schedule of the sweep in LDS (written once per kernel): for round r and slot s < 16 two words,
    // (12 * first joint | chain length << 16) and 12 * parent joint (or -1: the world), length 0 for an empty slot.
    // Keeps global-memory look-ups (and their latency) out of the sweep.
    TG_HD void init_sweep_schedule(bool rollout = false) {
        PROG &P = tg_fresh(this->P);
        TG_FOR(i, 4 * P.n_bodies) S[P.o_I + i] = P.b_inertia[i];   // body inertias: LDS copy for the whole kernel
        TG_FOR(c, P.nc) S[P.o_ctol + c] = P.c_tol[c];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_BBD)
        if constexpr (TEAM == 64 && tg_static_bbd<typename std::remove_cv<PROG>::type>::value) {
            // plan tables of the structured Newton solve (bbd.hpp): staged once per rollout kernel behind the base region
            if (rollout) { int *tab = (int *)(S + P.o_bbd); TG_FOR(i, 128) tab[i] = P.bbd_tab[i]; }
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_CMP)
        if constexpr (TEAM == 64 && !SPRINGS && tg_static_cmp<typename std::remove_cv<PROG>::type>::value) {
            // composite Newton matrix: representative item | its body << 12 | subtree group << 20 of every dynamic config
            if (rollout) { int *tab = (int *)(S + P.o_cmpt); TG_FOR(i, P.nd) { const int it = P.cmp_rep[i]; tab[i] = it | (P.it_pack[4 * (size_t)it] << 12) | (P.cmp_grp[i] << 20); } }
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        if (TEAM == 64 && P.tab_ok) {
            // Lane order of the 2 x n_joints (pose set, joint) items of the sin/cos pass: the rotary joints of both pose sets first,
            // then the prismatic ones -- so that the second trip of the wavefront (puppet: 68 items, 38 of them rotary) has no
            // sin/cos to evaluate and skips that code.  jck: config | kind << 12 | joint << 16 | pose set << 28.
            // The (pose set, joint) items of the dual sin/cos + local-transform pass, rotary ones first (so that a trip without rotary joints
            // skips that code): host-made lists (program.hpp) -- the rollout's holds only the poses it reads (sj_list), the derivative
            // kernels' every joint in both sets (sj_full).  jck: config | kind << 12 | joint << 16 | pose set << 28.
            sj_n = rollout ? P.n_sj : 2 * P.n_joints;
            if (rollout) sc_rot2 = P.n_sj_rot;
            else { int n_rot = 0; for (int j = 0; j < P.n_joints; j++) n_rot += P.j_kind[j] >= TG_RX ? 1 : 0; sc_rot2 = 2 * n_rot; }
            sj_quads = rollout;
            if constexpr (tg_static_sweep<typename std::remove_cv<PROG>::type>::value) {
                typedef typename std::remove_cv<PROG>::type SP;
                if (rollout) {
                    const int l = tg_opaque(lane);
                    swact = l < 60;
                    const int q = swact ? l / 12 : 4, rc = swact ? l - 12 * q : 0;
                    swr = rc >> 2; swcol = rc & 3;
#pragma unroll
                    for (int i = 0; i < 16; i++) swc[i] = (i & 3) < SP::sw_np[i >> 2] ? sw_code<SP>(i * 5, q) : 15;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int pos = lane + u * TEAM;
                jck[u] = pos < sj_n ? (rollout ? P.sj_list[pos] : P.sj_full[pos]) : 0;
                // dh items of the dynamic configs (program.hpp, dhr_pack): constraint | item index << 8, config
                const int n = lane + u * TEAM < P.n_dhr ? lane + u * TEAM : 0;
                if (P.n_dhr) { tck[u][0] = P.dhr_pack[8 * (size_t)n] | (P.dhr_pack[8 * (size_t)n + 7] << 8); tck[u][1] = P.dhr_pack[8 * (size_t)n + 1]; }
            }
            const int b = lane < 6 * P.n_bodies ? lane / 6 : 0;
            bio[0] = P.b_item_off[b]; bio[1] = P.b_item_off[b + 1];
            tdamp = P.damp[lane < P.nd ? lane : 0];
            if constexpr (tg_static_wev<typename std::remove_cv<PROG>::type>::value) {
                typedef typename std::remove_cv<PROG>::type SP;
                static_assert((SP::n_cmpairs + TEAM - 1) / TEAM <= 4, "eval_world: at most four trips of config pairs");
                if (rollout) {
#pragma unroll
                    for (int i = 0; i < 4; i++) wvl[i] = P.wev_lane[4 * lane + i];
#pragma unroll
                    for (int u = 0; u < 4; u++) wpair[u] = P.cmp_pair[lane + u * TEAM < SP::n_cmpairs ? lane + u * TEAM : 0];
                    if constexpr (tg_static_pk<SP>::value) {
                        // the config pairs with their two places in the packed image: a | b << 6 | address (a, b) << 12 | address (b, a) << 22
#pragma unroll
                        for (int u = 0; u < 4; u++) wpair[u] = P.wev_pairx[lane + u * TEAM < SP::n_cmpairs ? lane + u * TEAM : 0];
#pragma unroll
                        for (int u = 0; u < 2; u++) wdhx[u] = P.wev_dhx[lane + u * TEAM < SP::n_dhr ? lane + u * TEAM : 0];
                        wrhs = P.bbd_map[(lane < SP::nf ? lane : 0) * (SP::nf + 1) + SP::nf];
                        wone = lane < SP::bbd_pk_nones ? P.bbd_ones[lane < SP::bbd_pk_nones ? lane : 0] : -1;
                    }
                }
            }
            if (P.n_npairs) { const int *p0 = P.pair4 + 4 * (size_t)(lane < P.n_npairs ? lane : 0); tpair[0] = p0[0]; tpair[1] = p0[1]; tpair[2] = p0[2]; tpair[3] = p0[3]; }
        }
#endif
        if (!P.sched_ok || TEAM != 64) { TG_SYNC(); return; }
        int *sched = (int *)(S + P.o_sched);
        TG_FOR(idx, 16 * P.n_rounds) {
            const int r = idx >> 4, c = P.round_off[r] + (idx & 15);
            int w0 = 0, w1 = -1;
            if (c < P.round_off[r + 1]) {
                w0 = (12 * P.ch_first[c]) | (P.ch_len[c] << 16);
                w1 = P.ch_parent[c] >= 0 ? 12 * P.ch_parent[c] : -1;
            }
            sched[2 * idx] = w0; sched[2 * idx + 1] = w1;
        }
        TG_SYNC();
    }

    TG_HD void pose_sweep(bool on, int sel) {
        PROG &P = tg_fresh(this->P);
        Real *sc = S + P.o_sc, *G = S + P.o_G;
        if (on) TG_FOR(j, P.n_joints) {
            const Real x = qval(sel, P.j_cfg[j]);
            if (P.j_kind[j] >= TG_RX) tg_sincos(x, &sc[2 * j], &sc[2 * j + 1]);
            else { sc[2 * j] = x; sc[2 * j + 1] = 0.0; }
        }
        TG_SYNC();
        if (on) {  // local transforms, four independent coefficient rows in flight per lane
            const int n12 = 12 * P.n_joints;
            for (int base = lane; base < n12; base += 4 * TEAM) {
                double k0[4], k1[4], k2[4];
                int jj[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int idx = base + u * TEAM;
                    const bool ok = idx < n12;
                    const int j = ok ? idx / 12 : 0, e = ok ? idx % 12 : 0;
                    const double *k = P.jcoef + 4 * (size_t)(16 * j + e);
                    k0[u] = k[0]; k1[u] = k[1]; k2[u] = k[2]; jj[u] = j;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int idx = base + u * TEAM;
                    if (idx < n12) G[idx] = k0[u] + k1[u] * sc[2 * jj[u] + 1] + k2[u] * sc[2 * jj[u]];
                }
            }
        }
        TG_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
        if (TEAM == 64) {
            const int *sched = (const int *)(S + P.o_sched);
            for (int r = 0; r < P.n_rounds; r++) {
                const int nch = P.sched_ok ? 16 : P.round_off[r + 1] - P.round_off[r];
                if (on) TG_FOR(idx, 4 * nch) {
                    const int row = idx & 3;
                    int o0, len, opar;
                    if (P.sched_ok) {
                        const int w0 = sched[2 * (16 * r + (idx >> 2))];
                        opar = sched[2 * (16 * r + (idx >> 2)) + 1];
                        o0 = w0 & 0xFFFF; len = w0 >> 16;
                    } else {
                        const int ch = P.round_off[r] + (idx >> 2);
                        o0 = 12 * P.ch_first[ch]; len = P.ch_len[ch]; opar = P.ch_parent[ch] >= 0 ? 12 * P.ch_parent[ch] : -1;
                    }
                    if (row < 3 && len > 0) {
                        Real p0, p1, p2, p3;
                        if (opar >= 0) { const Real *gp = G + opar + 4 * row; p0 = gp[0]; p1 = gp[1]; p2 = gp[2]; p3 = gp[3]; }
                        else { p0 = row == 0 ? 1.0 : 0.0; p1 = row == 1 ? 1.0 : 0.0; p2 = row == 2 ? 1.0 : 0.0; p3 = 0.0; }
                        Real m[12];
#pragma unroll
                        for (int e = 0; e < 12; e++) m[e] = G[o0 + e];
                        for (int s = 0; s < len; s++) {
                            Real *gj = G + o0 + 12 * s;
                            Real n[12];                                // next local transform: loads before this step's stores
                            const Real *gn = s + 1 < len ? gj + 12 : gj;
#pragma unroll
                            for (int e = 0; e < 12; e++) n[e] = gn[e];
                            const Real v0 = p0 * m[0] + p1 * m[4] + p2 * m[8];
                            const Real v1 = p0 * m[1] + p1 * m[5] + p2 * m[9];
                            const Real v2 = p0 * m[2] + p1 * m[6] + p2 * m[10];
                            const Real v3 = p0 * m[3] + p1 * m[7] + p2 * m[11] + p3;
                            Real *out = gj + 4 * row;
                            out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3;
                            p0 = v0; p1 = v1; p2 = v2; p3 = v3;
#pragma unroll
                            for (int e = 0; e < 12; e++) m[e] = n[e];
                        }
                    }
                }
                TG_SYNC();
            }
            return;
        }
#endif
        for (int L = 1; L < P.n_levels; L++) {
            const int l0 = P.level_off[L], cnt = P.level_off[L + 1] - l0;
            if (on) TG_FOR(idx, 4 * cnt) {
                const int j = P.lvl_joints[l0 + (idx >> 2)], cc = idx & 3;
                if (P.j_parent[j] < 0) continue;
                const Real *gp = G + 12 * P.j_parent[j];
                Real *gj = G + 12 * j;
                const Real m0 = gj[cc], m1 = gj[4 + cc], m2 = gj[8 + cc], u3 = (cc == 3) ? 1.0 : 0.0;
                const Real v0 = gp[0] * m0 + gp[1] * m1 + gp[2] * m2 + u3 * gp[3];
                const Real v1 = gp[4] * m0 + gp[5] * m1 + gp[6] * m2 + u3 * gp[7];
                const Real v2 = gp[8] * m0 + gp[9] * m1 + gp[10] * m2 + u3 * gp[11];
                gj[cc] = v0; gj[4 + cc] = v1; gj[8 + cc] = v2;
            }
            TG_SYNC();
        }
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // ---- two pose sweeps in one pass (rollout Newton loop, full-wave teams) ---------------------------------------
    // Every Newton iteration needs the joint poses at the midpoint (Lagrangian terms) AND at q2 (constraint values
    // and Jacobians).  Swept one after the other they are two latency-bound recurrences of 8 barriers; here both run
    // through the same phases: the second pose set lives in the W area and its sin/cos in the J area (both dead until
    // the Jacobians / prefix velocities are formed), every chain lane carries the two row recurrences side by side
    // (two independent FMA chains per lane), and the local-transform pass covers 2 x 12 x n_joints entries.
    TG_HD void pose_sweep_dual(bool on, bool rollout_lists) {
        PROG &P = tg_fresh(this->P);
        const int sjn = rollout_lists ? P.n_sj : 2 * P.n_joints;     // (what init_sweep_schedule filled jck from)
        double *sc = S + P.o_sc, *sc2 = S + P.o_J, *G = S + P.o_G, *G2 = S + P.o_W;
        const int nj = P.n_joints;
        const int n12 = 12 * nj, n24 = 2 * n12;
        // coefficient rows of local-transform entry idx2 (of the 2 x 12 x n_joints of both pose sets; clamped past the end).  They
        // come from global memory (hundreds of cycles), so every trip's rows are requested one trip ahead -- the first
        // trip's before the sin/cos pass, which does not need them.
#if !defined(TG_NO_QUAD_SWEEP)
        // quad-lane chain rounds (chain_round_quads): the lane's instance of every pass of the first round, requested a phase ahead
        SwDesc sw0;
        if constexpr (tg_static_sweep<typename std::remove_cv<PROG>::type>::value) {
            if (rollout_lists) sw0 = sw_fetch<typename std::remove_cv<PROG>::type, 0>((const int *)(S + P.o_sched));
        }
#endif
        if (P.tab_ok) {
            // One lane per (pose set, joint): sin / cos of the joint coordinate AND the joint's local transform pre_j lg(q) in the same
            // phase.  With the pre-transform's columns in the order (axis a, b = a + 1, c = a + 2, translation) -- rows (A, B, C, D) of
            // P.j_prm -- the local transform is, row by row,
            //     rotary:     column a = A,  b = B cos + C sin,  c = C cos - B sin,  translation = D
            //     prismatic:  columns a, b, c = A, B, C,  translation = D + A q
            // i.e. twelve numbers, six (three) operations and twelve stores per joint, against twelve table rows of three coefficients,
            // twelve evaluations A + B cos + C sin and a trip through LDS for the sin / cos values entry by entry (13 wavefront trips for
            // the puppet's 2 x 34 joints, the rows from global memory batch after batch).  Same numbers.
            double pr[2][12];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const double *src = P.j_prm + 12 * (size_t)((jck[u] >> 16) & 0xFFF);
#pragma unroll
                for (int e = 0; e < 12; e++) pr[u][e] = src[e];
            }
            if (on) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int idx = lane + u * TEAM;
                    if (u * TEAM < sjn && idx < sjn) {
                        const bool second = (jck[u] >> 28) != 0;
                        const int j = (jck[u] >> 16) & 0xFFF, kind = (jck[u] >> 12) & 0xF;
                        const double x = qval(second ? dsB : dsA, jck[u] & 0xFFF);
                        const bool rotary = kind >= TG_RX;
                        const int a = rotary ? kind - TG_RX : kind - TG_TX, b = a == 2 ? 0 : a + 1, c = a == 0 ? 2 : a - 1;
                        double *g = (second ? G2 : G) + 12 * j;
                        if (u * TEAM < sc_rot2) {          // (wave-uniform) a trip with rotary joints in it
                            double sn = 0.0, cs = 1.0;
                            if (rotary) tg_sincos(x, &sn, &cs);
                            const double tq = rotary ? 0.0 : x;
#pragma unroll
                            for (int l = 0; l < 3; l++) {
                                const double A_ = pr[u][4 * l], B_ = pr[u][4 * l + 1], C_ = pr[u][4 * l + 2], D_ = pr[u][4 * l + 3];
                                g[4 * l + a] = A_;
                                g[4 * l + b] = rotary ? fma(C_, sn, B_ * cs) : B_;
                                g[4 * l + c] = rotary ? fma(-B_, sn, C_ * cs) : C_;
                                g[4 * l + 3] = fma(A_, tq, D_);
                            }
                        } else {
#pragma unroll
                            for (int l = 0; l < 3; l++) {
                                const double A_ = pr[u][4 * l], B_ = pr[u][4 * l + 1], C_ = pr[u][4 * l + 2], D_ = pr[u][4 * l + 3];
                                g[4 * l + a] = A_; g[4 * l + b] = B_; g[4 * l + c] = C_;
                                g[4 * l + 3] = fma(A_, x, D_);
                            }
                        }
                    }
                }
            }
            TG_SYNC();
            TG_STAMP(5);
            TG_STAMP(15);
        } else {
        constexpr int LT = TG_LT_TRIPS;        // trips of the wavefront per batch of coefficient rows
        struct Rows { double k0[LT], k1[LT], k2[LT]; int j[LT]; };
        auto rows_of = [&](int first) {
            Rows r;
#pragma unroll
            for (int u = 0; u < LT; u++) {
                const int idx2 = first + u * TEAM;
                const int idx = idx2 < n24 ? (idx2 >= n12 ? idx2 - n12 : idx2) : 0;
                const int j = idx / 12, e = idx % 12;
                const double *k = P.jcoef + 4 * (size_t)(16 * j + e);
                r.k0[u] = k[0]; r.k1[u] = k[1]; r.k2[u] = k[2]; r.j[u] = j;
            }
            return r;
        };
        Rows cur = rows_of(lane);
        if (on) TG_FOR(idx, 2 * nj) {
            const bool second = idx >= nj;
            const int j = second ? idx - nj : idx;
            const double x = qval(second ? dsB : dsA, P.j_cfg[j]);
            double *dst = (second ? sc2 : sc) + 2 * j;
            if (P.j_kind[j] >= TG_RX) tg_sincos(x, &dst[0], &dst[1]);
            else { dst[0] = x; dst[1] = 0.0; }
        }
        TG_SYNC();
        TG_STAMP(5);
        for (int b0 = 0; b0 < n24; b0 += LT * TEAM) {       // (uniform trip count: unrolled when the schedule is compiled in)
            const Rows nxt = b0 + LT * TEAM < n24 ? rows_of(b0 + LT * TEAM + lane) : cur;
            if (on) {
#pragma unroll
                for (int u = 0; u < LT; u++) {
                    const int idx2 = b0 + lane + u * TEAM;
                    if (idx2 < n24) {
                        const bool second = idx2 >= n12;
                        const double *scx = second ? sc2 : sc;
                        (second ? G2 : G)[second ? idx2 - n12 : idx2] = cur.k0[u] + cur.k1[u] * scx[2 * cur.j[u] + 1] + cur.k2[u] * scx[2 * cur.j[u]];
                    }
                }
            }
            cur = nxt;
        }
        TG_SYNC();
        TG_STAMP(15);
        }
        const int *sched = (const int *)(S + P.o_sched);
#if !defined(TG_NO_QUAD_SWEEP)
        if constexpr (tg_static_sweep<typename std::remove_cv<PROG>::type>::value) {
            if (rollout_lists) {     // (the instance plan lists the chains the ROLLOUT reads; the derivative kernels sweep every chain below)
                typedef typename std::remove_cv<PROG>::type SP;
                __builtin_amdgcn_s_setprio(TG_CHAIN_PRIO);
                chain_round_quads<SP, 0>(on, sched, sw0);
                __builtin_amdgcn_s_setprio(0);
                return;
            }
        }
#endif
        if (P.sched_ok == 2) {
            // at most 8 chains per round: lanes 0-31 sweep the midpoint poses, lanes 32-63 the q2 poses -- one row recurrence
            // per lane, so the instruction stream of a chain step is half that of two recurrences side by side
            double *Gh = lane >= 32 ? G2 : G;
            const int slot = (lane & 31) >> 2, row = lane & 3;
            for (int r = 0; r < P.n_rounds; r++) {
                if (on) {
                    const int w0 = sched[2 * (16 * r + slot)];
                    const int opar = sched[2 * (16 * r + slot) + 1];
                    const int o0 = w0 & 0xFFFF, len = w0 >> 16;
                    if (row < 3 && len > 0) {
                        double p0, p1, p2, p3;
                        if (opar >= 0) { const double *gp = Gh + opar + 4 * row; p0 = gp[0]; p1 = gp[1]; p2 = gp[2]; p3 = gp[3]; }
                        else { p0 = row == 0 ? 1.0 : 0.0; p1 = row == 1 ? 1.0 : 0.0; p2 = row == 2 ? 1.0 : 0.0; p3 = 0.0; }
                        double m[12];
#pragma unroll
                        for (int e = 0; e < 12; e++) m[e] = Gh[o0 + e];
                        for (int s = 0; s < len; s++) {
                            double *gj = Gh + o0 + 12 * s;
                            double n[12];                                // next local transform: loads before this step's stores
                            const int nx = s + 1 < len ? 12 : 0;
#pragma unroll
                            for (int e = 0; e < 12; e++) n[e] = gj[nx + e];
                            const double v0 = p0 * m[0] + p1 * m[4] + p2 * m[8];
                            const double v1 = p0 * m[1] + p1 * m[5] + p2 * m[9];
                            const double v2 = p0 * m[2] + p1 * m[6] + p2 * m[10];
                            const double v3 = p0 * m[3] + p1 * m[7] + p2 * m[11] + p3;
                            double *out = gj + 4 * row;
                            out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3;
                            p0 = v0; p1 = v1; p2 = v2; p3 = v3;
#pragma unroll
                            for (int e = 0; e < 12; e++) m[e] = n[e];
                        }
                    }
                }
                TG_SYNC();
            }
            return;
        }
        for (int r = 0; r < P.n_rounds; r++) {
            if (on) TG_FOR(idx, 64) {
                const int row = idx & 3;
                const int w0 = sched[2 * (16 * r + (idx >> 2))];
                const int opar = sched[2 * (16 * r + (idx >> 2)) + 1];
                const int o0 = w0 & 0xFFFF, len = w0 >> 16;
                if (row < 3 && len > 0) {
                    double p0, p1, p2, p3, q0, q1, q2, q3;
                    if (opar >= 0) {
                        const double *gp = G + opar + 4 * row, *gq = G2 + opar + 4 * row;
                        p0 = gp[0]; p1 = gp[1]; p2 = gp[2]; p3 = gp[3]; q0 = gq[0]; q1 = gq[1]; q2 = gq[2]; q3 = gq[3];
                    } else {
                        p0 = q0 = row == 0 ? 1.0 : 0.0; p1 = q1 = row == 1 ? 1.0 : 0.0; p2 = q2 = row == 2 ? 1.0 : 0.0; p3 = q3 = 0.0;
                    }
                    double m[12], h[12];
#pragma unroll
                    for (int e = 0; e < 12; e++) { m[e] = G[o0 + e]; h[e] = G2[o0 + e]; }
                    for (int s = 0; s < len; s++) {
                        double *gj = G + o0 + 12 * s, *hj = G2 + o0 + 12 * s;
                        double n[12], g[12];                          // next local transforms: loads before this step's stores
                        const int nxt = s + 1 < len ? 12 : 0;
#pragma unroll
                        for (int e = 0; e < 12; e++) { n[e] = gj[nxt + e]; g[e] = hj[nxt + e]; }
                        const double v0 = p0 * m[0] + p1 * m[4] + p2 * m[8];
                        const double w0_ = q0 * h[0] + q1 * h[4] + q2 * h[8];
                        const double v1 = p0 * m[1] + p1 * m[5] + p2 * m[9];
                        const double w1 = q0 * h[1] + q1 * h[5] + q2 * h[9];
                        const double v2 = p0 * m[2] + p1 * m[6] + p2 * m[10];
                        const double w2 = q0 * h[2] + q1 * h[6] + q2 * h[10];
                        const double v3 = p0 * m[3] + p1 * m[7] + p2 * m[11] + p3;
                        const double w3 = q0 * h[3] + q1 * h[7] + q2 * h[11] + q3;
                        double *out = gj + 4 * row, *out2 = hj + 4 * row;
                        out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3;
                        out2[0] = w0_; out2[1] = w1; out2[2] = w2; out2[3] = w3;
                        p0 = v0; p1 = v1; p2 = v2; p3 = v3; q0 = w0_; q1 = w1; q2 = w2; q3 = w3;
#pragma unroll
                        for (int e = 0; e < 12; e++) { m[e] = n[e]; h[e] = g[e]; }
                    }
                }
            }
            TG_SYNC();
        }
    }
    // ---- the chain rounds with the 3 x 4 entries of a running pose on twelve lanes (system-specialised kernels) ----
    // Lane (r, c) of an instance (one chain of one pose set; five instances per pass) carries entry (r, c) of the pose: a chain step is
    //   G(r, c) <- sum_k G(r, k) L(k, c) (+ G(r, 3) for c = 3),
    // with G(r, k) read from lane k of the lane's own quad (DPP quad_perm) and column c of the local transform L -- three doubles per step, so
    // a whole chain's columns fit in registers and are ALL requested before the recurrence starts.  The row-per-lane sweep above needs
    // all twelve entries of L per step and lane, cannot hold more than one step's worth, and pays an LDS round trip per chain step (ten in a
    // row for the puppet); here a round costs one.  Passes of a round are independent and interleaved.  Same products; the translation
    // entry G(r, 3) opens the sum instead of closing it (three dependent operations per step instead of four).
    struct SwDesc { int w0[4], par[4]; };      // per pass: 12 * first joint | chain length << 16, 12 * parent joint (or -1) of the lane's instance
    // instance q (the lane's, 0 .. 4) of the pass whose five plan words start at `base`: chain slot | pose set << 8; a lane without an
    // instance gets slot 15, which the schedule leaves empty (length 0).  The plan words are compile-time constants: four selects.
    template <class SP> TG_HD static int sw_code(int base, int q) {
        auto w = [&](int i) { const int c = SP::sw_inst[base + i]; return c ? (c & 0x1FF) : 15; };
        return q == 0 ? w(0) : q == 1 ? w(1) : q == 2 ? w(2) : q == 3 ? w(3) : w(4);
    }
    template <class SP, int RD> TG_HD SwDesc sw_fetch(const int *sched) const {
        SwDesc d;
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            d.w0[ps] = 0; d.par[ps] = -1;
            if (ps < SP::sw_np[RD < 4 ? RD : 0]) {
                const int slot = swc[4 * (RD < 4 ? RD : 0) + ps] & 0xFF;
                d.w0[ps] = sched[2 * (16 * RD + slot)]; d.par[ps] = sched[2 * (16 * RD + slot) + 1];
            }
        }
        return d;
    }
    // (the round's descriptors arrive fetched -- by the previous round, the first round's by the local-transform pass -- and the next
    // round's are requested before this round's columns: no LDS round trip between "which chain" and "its transforms")
    template <class SP, int RD> TG_HD void chain_round_quads(bool on, const int *sched, const SwDesc &d) {
        if constexpr (RD < SP::n_rounds) {
            PROG &P = tg_fresh(this->P);
            constexpr int NP = SP::sw_np[RD], ML = SP::sw_maxlen;
            SwDesc nxt = d;
            if constexpr (RD + 1 < SP::n_rounds) nxt = sw_fetch<SP, RD + 1>(sched);
            const int r = swr, c = swcol;
            const bool act = swact;
            if (on) {
                double m[4][ML][3], p[4];
                int base[4], lim[4];
                const int dead = P.o_sc + (lane < 2 * P.n_joints ? lane : 0);
#pragma unroll
                for (int ps = 0; ps < 4; ps++) {
                    if (ps < NP) {
                        const int set = (swc[4 * RD + ps] >> 8) ? P.o_W : P.o_G, opar = d.par[ps];
                        lim[ps] = act ? d.w0[ps] >> 16 : 0;
                        base[ps] = set + (d.w0[ps] & 0xFFFF);
                        const double pv = S[set + (opar >= 0 ? opar : 0) + 4 * r + c];
                        p[ps] = opar >= 0 ? pv : (r == c ? 1.0 : 0.0);
#pragma unroll
                        for (int s = 0; s < ML; s++) {
                            if (s < SP::sw_len[4 * RD + ps]) {
                                // (past the end of a shorter chain of the pass: whatever follows it in LDS -- never stored)
                                const int o = base[ps] + 12 * s + c;
                                m[ps][s][0] = S[o]; m[ps][s][1] = S[o + 4]; m[ps][s][2] = S[o + 8];
                            }
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < ML; s++) {
#pragma unroll
                    for (int ps = 0; ps < 4; ps++) {
                        if (ps < NP && s < SP::sw_len[4 * RD + ps]) {
                            const double b0 = tg_quad_bcast<0>(p[ps]), b1 = tg_quad_bcast<1>(p[ps]), b2 = tg_quad_bcast<2>(p[ps]);
                            const double last = c == 3 ? p[ps] : 0.0;
                            const double v = fma(b2, m[ps][s][2], fma(b1, m[ps][s][1], fma(b0, m[ps][s][0], last)));
                            // branch-free store (lanes past their chain's end write a dead word of the sin/cos area): a guarded store
                            // would put every (step, pass) in a basic block of its own and the passes could no longer interleave
                            S[s < lim[ps] ? base[ps] + 12 * s + 4 * r + c : dead] = v;
                            p[ps] = v;
                        }
                    }
                }
            }
            TG_SYNC();
            chain_round_quads<SP, RD + 1>(on, sched, nxt);
        }
    }
    // the dual sweep needs the chain schedule in LDS and room for the second pose set in the J / W areas
    TG_HD bool dual_ok() const { return TEAM == 64 && !SPRINGS && P.sched_ok && P.nc > 0 && 6 * P.n_items >= 12 * P.n_joints; }

    // eval_midpoint followed by eval_constraints(on, 2, true, Dh2) with the two pose sweeps fused
    TG_HD void eval_both(bool on) {
        PROG &P = tg_fresh(this->P);
        if (on) TG_FOR(i, P.nq) S[P.o_dq + i] = (S[P.o_q2 + i] - S[P.o_q1 + i]) / dt;
        TG_SYNC();
        TG_STAMP(0);
        pose_sweep_dual(on, true);
        TG_STAMP(1);
        oGc = P.o_W;
        attach_points(on, true, true);            // bodies from the midpoint poses, end points from the q2 poses
        constraints(on, 2, true, S + P.o_Dh2, 0);
        oGc = P.o_G;
        TG_STAMP(6);
        jacobians(on);
        TG_STAMP(2);
        velocities(on);
        TG_STAMP(3);
        residual_dyn(on);
        TG_STAMP(4);
    }
    // ---- the same with every phase's table rows requested ONE PHASE AHEAD (P.tab_ok: each per-lane table walk fits two trips) ----
    // A phase of the Newton iteration starts with look-ups in the schedule tables (global memory: several hundred cycles) and
    // only then reads LDS; with the rows of phase k+1 requested before phase k runs, that latency is covered by phase k.
    // The arithmetic of every phase is that of attach_points / constraints / jacobians / velocities / residual_dyn, in the same order.
    struct AttachTab { double ce[2], c0[2], c1[2], c2[2]; int ga[2]; double o0, o1, o2, orr; int ea; };
    struct ConTab { int e1, e2, type, comp, cfg; double dist; int rec[2][6]; };
    struct ItemTab { int rec[2][4]; };
    struct ResTab { int n0, n1; int look[8]; };
    TG_HD AttachTab fetch_attach() const {
        AttachTab t;
        const int l = tg_opaque(lane);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int ii = l + u * TEAM < 12 * P.n_bodies ? l + u * TEAM : 0;
            const double *C = P.at_d + 4 * (size_t)ii;
            t.ce[u] = C[0]; t.c0[u] = C[1]; t.c1[u] = C[2]; t.c2[u] = C[3]; t.ga[u] = P.at_i[ii];
        }
        const int ii = l < 3 * P.n_endpoints ? l : 0;
        const double *o = P.ae_d + 4 * (size_t)ii;
        t.o0 = o[0]; t.o1 = o[1]; t.o2 = o[2]; t.orr = o[3]; t.ea = P.ae_i[ii];
        return t;
    }
    TG_HD ConTab fetch_constraints() const {
        ConTab t;
        const int l = tg_opaque(lane);
        const int c = l < P.nc ? l : 0;
        t.e1 = P.c_e1[c]; t.e2 = P.c_e2[c]; t.type = P.c_type[c]; t.comp = P.c_comp[c]; t.cfg = P.c_cfg[c]; t.dist = P.c_dist[c];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int *rec = P.dhr_pack + 8 * (size_t)(l + u * TEAM < P.n_dhr ? l + u * TEAM : 0);
#pragma unroll
            for (int i = 1; i < 6; i++) t.rec[u][i] = rec[i];
            t.rec[u][0] = rec[7];                               // the item's index in Dh2
        }
        return t;
    }
    TG_HD ItemTab fetch_items() const {
        ItemTab t;
        const int l = tg_opaque(lane);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int *rec = P.it_pack + 4 * (size_t)(l + u * TEAM < P.n_items ? l + u * TEAM : 0);
#pragma unroll
            for (int i = 0; i < 4; i++) t.rec[u][i] = rec[i];
        }
        return t;
    }
    TG_HD ResTab fetch_residual() const {
        ResTab t;
        const int l = tg_opaque(lane);
        const int i = l < P.nd ? l : 0;
        t.n0 = P.cfg_item_off[i]; t.n1 = P.cfg_item_off[i + 1];
#pragma unroll
        for (int c = 0; c < 8; c++) t.look[c] = c < P.nc ? P.dh_lookup[c * P.nq + i] : -1;
        return t;
    }

    // Six doubles of an item (its Jacobian column J, its W) as three 16-byte LDS accesses where the area starts on an even double (the
    // specialised kernels check it at compile time): at the items' 48-byte stride a 16-byte access is conflict free, an 8-byte one is
    // two-way and a ds_read2_b64 pair four times the LDS cycles of the 16-byte read (profiles/r04_lds_conflicts.txt)
    template <bool VEC> TG_HD static void ld6(const double *p, double (&x)[6]) {
        if constexpr (VEC) {
            typedef double tg_d2 __attribute__((ext_vector_type(2)));
            const tg_d2 *q = reinterpret_cast<const tg_d2 *>(p);
            const tg_d2 a = q[0], b = q[1], c = q[2];
            x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; x[4] = c.x; x[5] = c.y;
        } else {
#pragma unroll
            for (int r = 0; r < 6; r++) x[r] = p[r];
        }
    }
    template <bool VEC> TG_HD static void st6(double *p, const double (&x)[6]) {
        if constexpr (VEC) {
            typedef double tg_d2 __attribute__((ext_vector_type(2)));
            tg_d2 *q = reinterpret_cast<tg_d2 *>(p);
            q[0] = tg_d2{x[0], x[1]}; q[1] = tg_d2{x[2], x[3]}; q[2] = tg_d2{x[4], x[5]};
        } else {
#pragma unroll
            for (int r = 0; r < 6; r++) p[r] = x[r];
        }
    }
    TG_HD void eval_both_tab(bool on) {
        constexpr bool VJ = tg_items_aligned<typename std::remove_cv<PROG>::type>::value;
        PROG &P = tg_fresh(this->P);
        const AttachTab at = fetch_attach();
        if (!rates_ready) {      // (the rollout forms the rates in its step set-up and in the Newton update)
            if (on) TG_FOR(i, P.nq) S[P.o_dq + i] = (S[P.o_q2 + i] - S[P.o_q1 + i]) / dt;
            TG_SYNC();
        }
        TG_STAMP(0);
        pose_sweep_dual(on, true);
        TG_STAMP(1);
        const double *G = S + P.o_G, *G2 = S + P.o_W;
        // ---- body poses (midpoint) and constraint end points (q2 poses): attach_points ----
        const ConTab ct = fetch_constraints();
        if (on) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int idx = lane + u * TEAM;
                if (idx < 12 * P.n_bodies) {
                    const int e = idx % 12, r = e >> 2, c = e & 3;
                    const int anchor = at.ga[u];
                    const double *g = G + 12 * (anchor < 0 ? 0 : anchor) + 4 * r;
                    const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
                    const double val = g0 * at.c0[u] + g1 * at.c1[u] + g2 * at.c2[u] + (c == 3 ? g3 : 0.0);
                    S[P.o_gB + idx] = anchor < 0 ? at.ce[u] : val;
                }
            }
            if (lane < 3 * P.n_endpoints) {
                const int r = lane % 3;
                const double *g = G2 + 12 * (at.ea < 0 ? 0 : at.ea) + 4 * r;
                const double val = g[0] * at.o0 + g[1] * at.o1 + g[2] * at.o2 + g[3];
                S[P.o_pE + lane] = at.ea < 0 ? at.orr : val;
            }
        }
        TG_SYNC();
        // ---- constraint values and Dh2 at q2: constraints(on, 2, true, Dh2, 0) ----
        const ItemTab it = fetch_items();
        if (on) {
            if (lane < P.nc) {
                const double *a = S + P.o_pE + 3 * ct.e1, *b = S + P.o_pE + 3 * ct.e2;
                const double vx = a[0] - b[0], vy = a[1] - b[1], vz = a[2] - b[2];
                double h;
                if (ct.type == TG_CONSTRAINT_POINT) h = ct.comp == 0 ? vx : (ct.comp == 1 ? vy : vz);
                else {
                    const double len = ct.cfg >= 0 ? S[P.o_q2 + (ct.cfg >= 0 ? ct.cfg : 0)] : ct.dist;
                    h = (vx * vx + vy * vy + vz * vz) - len * len;
                }
                S[P.o_f + P.nd + lane] = h;
                res_hoff = fabs(h) > S[P.o_ctol + lane];
            } else res_hoff = false;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                // (only the items of the dynamic configs: nothing in the rollout reads Dh with respect to a kinematic config)
                if (u * TEAM < P.n_dhr && lane + u * TEAM < P.n_dhr) {
                    const int n = ct.rec[u][0];
                    const int k = ct.rec[u][1], oj = ct.rec[u][2], w = ct.rec[u][3], oe1 = ct.rec[u][4], oe2 = ct.rec[u][5];
                    const int side = w & 0xFF, kind = (w >> 8) & 0xFF, type = (w >> 16) & 0xFF, comp = w >> 24;
                    const int ojc = oj < 0 ? 0 : oj;
                    double d1[3], d2[3];
                    const double *gj = G2 + ojc;
                    const bool prismatic = kind <= TG_TZ;
                    const int ax = prismatic ? kind - TG_TX : kind - TG_RX;
                    const double wx = gj[ax], wy = gj[4 + ax], wz = gj[8 + ax];
                    {
                        const double *pe = S + P.o_pE + oe1;
                        const double dx = pe[0] - gj[3], dy = pe[1] - gj[7], dz = pe[2] - gj[11];
                        d1[0] = prismatic ? wx : wy * dz - wz * dy; d1[1] = prismatic ? wy : wz * dx - wx * dz; d1[2] = prismatic ? wz : wx * dy - wy * dx;
                    }
                    {
                        const double *pe = S + P.o_pE + oe2;
                        const double dx = pe[0] - gj[3], dy = pe[1] - gj[7], dz = pe[2] - gj[11];
                        d2[0] = prismatic ? wx : wy * dz - wz * dy; d2[1] = prismatic ? wy : wz * dx - wx * dz; d2[2] = prismatic ? wz : wx * dy - wy * dx;
                    }
                    const double s1 = (side & 1) ? 1.0 : 0.0, s2 = (side & 2) ? 1.0 : 0.0;
                    const double dx = s1 * d1[0] - s2 * d2[0], dy = s1 * d1[1] - s2 * d2[1], dz = s1 * d1[2] - s2 * d2[2];
                    double val;
                    if (type == TG_CONSTRAINT_POINT) val = comp == 0 ? dx : (comp == 1 ? dy : dz);
                    else {
                        const double *a = S + P.o_pE + oe1, *b = S + P.o_pE + oe2;
                        val = (a[0] - b[0]) * dx + (a[1] - b[1]) * dy + (a[2] - b[2]) * dz;
                        if (side & 4) val -= S[P.o_q2 + k];
                        val *= 2.0;
                    }
                    S[P.o_Dh2 + n] = val;
                }
            }
        }
        // (no barrier: the Jacobian columns below read the body poses and the midpoint joint poses, write J / dqi / gam -- nothing the
        // constraint rows above touch)
        TG_STAMP(6);
        // ---- body Jacobian columns and body-frame gravity: jacobians ----
        const ResTab rt = fetch_residual();
        if (on) {
            double Jv[2][6], dqv[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int b = it.rec[u][0], oj = it.rec[u][1], kind = it.rec[u][2], cfg = it.rec[u][3] & 0xFFFF;
                const double *gb = S + P.o_gB + 12 * b, *gj = G + oj;
                const bool prismatic = kind <= TG_TZ;
                const int ax = prismatic ? kind - TG_TX : kind - TG_RX;
                const double a0 = gj[ax], a1 = gj[4 + ax], a2 = gj[8 + ax];
                const double dx = gb[3] - gj[3], dy = gb[7] - gj[7], dz = gb[11] - gj[11];
                double lin[3], ang[3];
                ang[0] = prismatic ? 0.0 : a0; ang[1] = prismatic ? 0.0 : a1; ang[2] = prismatic ? 0.0 : a2;
                lin[0] = prismatic ? a0 : a1 * dz - a2 * dy;
                lin[1] = prismatic ? a1 : a2 * dx - a0 * dz;
                lin[2] = prismatic ? a2 : a0 * dy - a1 * dx;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    Jv[u][r] = gb[r] * lin[0] + gb[4 + r] * lin[1] + gb[8 + r] * lin[2];
                    Jv[u][3 + r] = gb[r] * ang[0] + gb[4 + r] * ang[1] + gb[8 + r] * ang[2];
                }
                dqv[u] = S[P.o_dq + cfg];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * TEAM;
                if (i < P.n_items) {
                    st6<VJ>(S + P.o_J + 6 * i, Jv[u]);
                    S[P.o_dqi + i] = dqv[u];
                }
            }
            if (lane < 3 * P.n_bodies) {
                const int b = lane / 3, r = lane % 3;
                const double *gb = S + P.o_gB + 12 * b;
                S[P.o_gam + lane] = gb[r] * P.grav[0] + gb[4 + r] * P.grav[1] + gb[8 + r] * P.grav[2];
            }
        }
        TG_SYNC();
        TG_STAMP(2);
        // ---- prefix velocities, W = [P, J], body velocities: velocities ----
        if (on && lane < 6 * P.n_bodies) {
            const int m = lane % 6;
            const int first = bio[0], last = bio[1];
            double acc = 0.0;
            constexpr int PG = TG_PREFIX_GROUP;     // items whose operands are requested together
            for (int k = first; k < last; k += PG) {
                double jv[PG], dv[PG];
#pragma unroll
                for (int u = 0; u < PG; u++) {
                    const int kk = k + u < last ? k + u : last - 1;
                    jv[u] = S[P.o_J + 6 * kk + m]; dv[u] = S[P.o_dqi + kk];
                }
#pragma unroll
                for (int u = 0; u < PG; u++) {
                    if (k + u < last) { S[P.o_W + 6 * (k + u) + m] = acc; acc = fma(jv[u], dv[u], acc); }
                }
            }
            S[P.o_vB + lane] = acc;
        }
        TG_SYNC();
        TG_STAMP(3);
        // ---- W = [P, J] of every item (velocities' second half) and, from it in registers, L_dq, L_ddq terms: residual_dyn ----
        // (the item's lane forms the bracket, uses it and stores it for the Newton matrix: as a phase of its own the bracket costs a
        // barrier and a second trip of J and W through LDS)
        double *terms = S + P.o_G;
        if (on) {
            double ta[2], tb[2], Wv[2][6];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * TEAM < P.n_items ? lane + u * TEAM : lane;
                const int b = it.rec[u][0];
                const double *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b;
                const double *J = S + P.o_J + 6 * i, *Pw = S + P.o_W + 6 * i, *gam = S + P.o_gam + 3 * b;
                double Jr[6], Pp[6];
                ld6<VJ>(J, Jr); ld6<VJ>(Pw, Pp);
                bracket(Pp, Jr, Wv[u]);
                ta[u] = inner6(I, Jr, v);
                tb[u] = inner6(I, Wv[u], v) + I[0] * (gam[0] * Jr[0] + gam[1] * Jr[1] + gam[2] * Jr[2]);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (lane + u * TEAM < P.n_items) {
                    const int slot = it.rec[u][3] >> 16; terms[2 * slot] = ta[u]; terms[2 * slot + 1] = tb[u];
                    st6<VJ>(S + P.o_W + 6 * (lane + u * TEAM), Wv[u]);
                }
            }
        }
        TG_SYNC();
        if (on && lane < P.nd) {
            const int i = lane;
            double ldq = 0.0, lddq = 0.0;
            if constexpr (tg_is_spec<typename std::remove_cv<PROG>::type>::value) {
                // compile-time trip count (the most items any config has) with all loads ahead of the adds: as a loop over [n0, n1) every
                // term waits out its own LDS read (ten in a row for a torso config).  Same order of the same additions.
                typedef typename std::remove_cv<PROG>::type SP;
                double ta_[SP::max_cfg_items > 0 ? SP::max_cfg_items : 1], tb_[SP::max_cfg_items > 0 ? SP::max_cfg_items : 1];
#pragma unroll
                for (int u = 0; u < SP::max_cfg_items; u++) { const int n = rt.n0 + u < rt.n1 ? rt.n0 + u : rt.n0; ta_[u] = terms[2 * n]; tb_[u] = terms[2 * n + 1]; }
#pragma unroll
                for (int u = 0; u < SP::max_cfg_items; u++) if (rt.n0 + u < rt.n1) { lddq += ta_[u]; ldq += tb_[u]; }
            } else
            for (int n = rt.n0; n < rt.n1; n++) { lddq += terms[2 * n]; ldq += terms[2 * n + 1]; }
            S[P.o_Ldq + i] = ldq; S[P.o_Lddq + i] = lddq;
            double force = -tdamp * S[P.o_dq + i];
            for (int k = 0; k < P.n_cf; k++) if (P.cf_cfg[k] == i) force += S[P.o_u + P.cf_in[k]];
            double f = S[P.o_p1 + i] + (0.5 * dt * ldq - lddq) + dt * force;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (c < P.nc) {
                    const int n = rt.look[c];
                    f -= (n >= 0 ? S[P.o_lam + c] : 0.0) * S[P.o_Dh1 + (n >= 0 ? n : 0)];
                }
            }
            S[P.o_f + i] = f;
            res_f2 = f * f;
        } else res_f2 = 0.0;
        // (no barrier: the convergence test that follows works from res_f2 / res_hoff in registers -- solved_fused -- and the next reader
        // of f, the Newton matrix's last phase, is several barriers away)
        TG_STAMP(4);
    }
    // midpointvi.c:709-716 on the values eval_both_tab left in registers: |f_dynamic|_2 <= tolerance (as |f|^2 <= tolerance^2: no square
    // root on the critical path) and every constraint inside its own tolerance.  The sum goes along DPP row shifts; as solved() -- every
    // lane re-reading f from LDS behind a barrier and summing it in four chains -- the test cost as much as a phase of the evaluation.
    TG_HD bool solved_fused(double tolerance) const {
        double v = res_f2;
        v += tg_dpp_f64<0x111>(v); v += tg_dpp_f64<0x112>(v); v += tg_dpp_f64<0x114>(v); v += tg_dpp_f64<0x118>(v);   // row_shr 1, 2, 4, 8
        double norm2 = tg_readlane_f64(v, 15);
        if (P.nd > 16) norm2 += tg_readlane_f64(v, 31);
        if (P.nd > 32) norm2 += tg_readlane_f64(v, 47);
        if (P.nd > 48) norm2 += tg_readlane_f64(v, 63);
        return !(norm2 > tolerance * tolerance) && !__any(res_hoff ? 1 : 0);
    }

    // ---- the evaluation of a rollout's Newton iteration in WORLD-frame form (round 5; system-specialised kernels with P.wev_ok) ----
    // eval_both_tab evaluates the Lagrangian terms of the residual per (body, path config) ITEM (88 for the puppet: two wavefront trips
    // for the Jacobian columns, a serial prefix sum per body, two trips for the brackets and the inner products, a per-config sum) and the
    // composite Newton matrix then re-derives world-frame twists and momenta from those body-frame quantities.  The same numbers per CONFIG:
    //     s_k   world twist of joint k (velocity of the point at the world origin, angular velocity), read off the joint's world pose,
    //     V_k^- = sum of s_j dq_j over the configs j above k on its path,   w_k = [V_k^-, s_k]   (= Ad(g_F) W_{F,k} for every body F below k),
    //     H_k   = sum over the bodies F below k of their spatial momentum about the world origin,  (M_k, C_k) their mass and first moment:
    //     L_ddq_k = s_k . H_k,        L_dq_k = w_k . H_k + g . (M_k v_k + omega_k x C_k)           (system.c:129-202; gravity.c:27-40)
    // -- the quantities the composite matrix needs anyway (newton_matrix_world takes them from here).  Phases:
    //   E3  body poses / constraint end points (as attach_points), and lane k < nd: s_k and u_k = s_k dq_k from the midpoint pose of joint k;
    //   E4  constraint values and Dh2 at q2 (as constraints), and one "sum u over my list" per lane: config lanes get V_k^- and form w_k
    //       (kept in registers), lane (body b, axis r) gets the body's world twist and writes its world inertia / momentum entries;
    //   E5  subtree composites (16 lanes, compile-time membership);
    //   E6  lane k < nd: L_dq, L_ddq, forces, the residual entry.
    // Five phases of one trip each instead of eight (three of them two trips); verified against the reference's L_dq / L_ddq in
    // tools/proto/world_eval.py (1e-16) before it was written here.  The derivative kernels keep the item form (they need J and W).
    template <int D0, int D1> TG_HD void wev_sum(const double *SW, double (&V)[6]) const {
        // V += the u-halves of list entries D0 .. D1-1 (padded entries point at the all-zero record): every load before the adds
        if constexpr (D0 < D1) {
            double x[D1 - D0][6];
#pragma unroll
            for (int d = D0; d < D1; d++) {
                const int idx = (wvl[d >> 2] >> (8 * (d & 3))) & 0xFF;
                ld6<true>(SW + 12 * idx + 6, x[d - D0]);
            }
#pragma unroll
            for (int d = D0; d < D1; d++) {
#pragma unroll
                for (int r = 0; r < 6; r++) V[r] += x[d - D0][r];
            }      // (pairwise sums -- a shorter dependent chain -- measured: no difference)
        }
    }
    TG_HD void eval_world(bool on) {
        typedef typename std::remove_cv<PROG>::type SP;
        constexpr int nd = SP::nd, NB = SP::n_bodies, NG = SP::n_cgroups, MAXD = SP::wev_depth;
        static_assert(TEAM == 64 && nd + 3 * NB < 64, "eval_world: one lane per config and per (body, axis), and the last lane free");
        static_assert((SP::o_csw & 1) == 0 && (SP::o_I & 1) == 0 && (SP::o_cmp & 1) == 0 && (SP::o_G & 1) == 0, "eval_world: 16-byte LDS accesses");
        PROG &P = tg_fresh(this->P);
        // table rows of E3, requested ahead of the pose sweep: the end point's anchor row (as fetch_attach), the body lane's constant offset
        const int l0 = tg_opaque(lane);
        const bool blane = l0 >= nd && l0 < nd + 3 * NB;
        const int wb_ = blane ? wvl[3] & 0xFF : 0, wr_ = blane ? (wvl[3] >> 8) & 3 : 0;
        double eo0, eo1, eo2, eor; int eanc;
        {
            const int ii = l0 < 3 * P.n_endpoints ? l0 : 0;
            const double *o = P.ae_d + 4 * (size_t)ii;
            eo0 = o[0]; eo1 = o[1]; eo2 = o[2]; eor = o[3]; eanc = P.ae_i[ii];
        }
        double Cb[12];
#pragma unroll
        for (int e = 0; e < 12; e++) Cb[e] = P.b_C[12 * (size_t)wb_ + e];
        const int banc = P.b_anchor[wb_];
        if (!rates_ready) {
            if (on) TG_FOR(i, P.nq) S[P.o_dq + i] = (S[P.o_q2 + i] - S[P.o_q1 + i]) / dt;
            TG_SYNC();
        }
        TG_STAMP(0);
        pose_sweep_dual(on, true);
        TG_STAMP(1);
        const double *G = S + P.o_G, *G2 = S + P.o_W;
        // per-body world entries: behind the q2 poses in the W area (with the dead body-velocity / gravity vectors that follow it) -- the body
        // lanes write them while other lanes still read the midpoint poses
        double *SW = S + P.o_csw, *CMP = S + P.o_cmp, *BW = S + P.o_W + 12 * P.n_joints;
        constexpr int BWS = 17;
        static_assert(12 * SP::n_joints + BWS * NB <= 6 * SP::n_items + 9 * NB, "eval_world: body entries do not fit behind the q2 poses");
        // ---- E3: constraint end points (q2 poses); lane k <= nd: world twist of config k; lane (body b, axis r): the body's world pose
        //      from its anchor joint's, and from it the mass entries of the body -- M, C_r = m p_r, row r of D = R I R' + m (|p|^2 1 - p p')
        //      (kept in registers for the momentum, which needs the body's velocity: E4)
        const ConTab ct = fetch_constraints();
        double bD0 = 0.0, bD1 = 0.0, bD2 = 0.0, bC0 = 0.0, bC1 = 0.0, bC2 = 0.0, bm = 0.0;
        if (on) {
            if (lane < 3 * P.n_endpoints) {
                const int r = lane % 3;
                const double *g = G2 + 12 * (eanc < 0 ? 0 : eanc) + 4 * r;
                const double val = g[0] * eo0 + g[1] * eo1 + g[2] * eo2 + g[3];
                S[P.o_pE + lane] = eanc < 0 ? eor : val;
            }
            if (lane < nd || lane == TEAM - 1) {       // (the last lane, which has no other role, writes the all-zero record the padded list entries point at)
                const bool real = lane < nd;
                const int rec = real ? lane : nd;
                const int oj = real ? wvl[3] & 0xFFFF : 0, kind = real ? (wvl[3] >> 16) & 0xFF : (int)TG_TX;
                const bool prismatic = kind <= TG_TZ;
                const int ax = prismatic ? kind - TG_TX : kind - TG_RX;
                const double *gj = G + oj;
                const double a0 = gj[ax], a1 = gj[4 + ax], a2 = gj[8 + ax];
                const double px = gj[3], py = gj[7], pz = gj[11];
                const double dqk = real ? S[P.o_dq + (real ? lane : 0)] : 0.0;
                double sv[6], uv[6];
                // rotary: angular velocity a about the joint's origin p: the point at the world origin moves with a x (0 - p) = p x a
                sv[0] = prismatic ? a0 : py * a2 - pz * a1; sv[1] = prismatic ? a1 : pz * a0 - px * a2; sv[2] = prismatic ? a2 : px * a1 - py * a0;
                sv[3] = prismatic ? 0.0 : a0; sv[4] = prismatic ? 0.0 : a1; sv[5] = prismatic ? 0.0 : a2;
#pragma unroll
                for (int r = 0; r < 6; r++) { sv[r] = real ? sv[r] : 0.0; uv[r] = sv[r] * dqk; }
                st6<true>(SW + 12 * rec, sv);
                st6<true>(SW + 12 * rec + 6, uv);
            } else if (blane) {
                const int b = wb_, r = wr_;
                const double *ga = G + 12 * (banc < 0 ? 0 : banc);
                double Ga[12];
                {
                    typedef double tg_d2 __attribute__((ext_vector_type(2)));
                    const tg_d2 *q = reinterpret_cast<const tg_d2 *>(ga);
#pragma unroll
                    for (int e = 0; e < 6; e++) { const tg_d2 v = q[e]; Ga[2 * e] = v.x; Ga[2 * e + 1] = v.y; }
                }
                if (banc < 0) {
#pragma unroll
                    for (int e = 0; e < 12; e++) Ga[e] = (e == 0 || e == 5 || e == 10) ? 1.0 : 0.0;
                }
                // world pose of the body: R = Ra Rc, p = Ra pc + pa (Rc = 1 for every body of most models -- the masses sit at translated frames:
                // a compile-time property of the schedule, wev_rc_ident)
#if defined(TG_WEV_NO_RC_IDENT)
                constexpr bool RCI = false;
#else
                constexpr bool RCI = SP::wev_rc_ident != 0;
#endif
                double R[3][3], p[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
#pragma unroll
                    for (int j = 0; j < 3; j++) R[i][j] = RCI ? Ga[4 * i + j] : Ga[4 * i] * Cb[j] + Ga[4 * i + 1] * Cb[4 + j] + Ga[4 * i + 2] * Cb[8 + j];
                    p[i] = Ga[4 * i] * Cb[3] + Ga[4 * i + 1] * Cb[7] + Ga[4 * i + 2] * Cb[11] + Ga[4 * i + 3];
                }
                const double *I = S + P.o_I + 4 * b;
                const double m = I[0], I1 = I[1], I2 = I[2], I3 = I[3];
                const double Rr0 = r == 0 ? R[0][0] : (r == 1 ? R[1][0] : R[2][0]), Rr1 = r == 0 ? R[0][1] : (r == 1 ? R[1][1] : R[2][1]);
                const double Rr2 = r == 0 ? R[0][2] : (r == 1 ? R[1][2] : R[2][2]), pr = r == 0 ? p[0] : (r == 1 ? p[1] : p[2]);
                const double p2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
                double Dr[3];
#pragma unroll
                for (int j = 0; j < 3; j++) Dr[j] = Rr0 * I1 * R[j][0] + Rr1 * I2 * R[j][1] + Rr2 * I3 * R[j][2] + m * ((j == r ? p2 : 0.0) - pr * p[j]);
                double *o = BW + BWS * b;
                if (r == 0) o[0] = m;
                o[1 + r] = m * pr;
                // D row r, columns j >= r: entries 4 + (xx xy xz | yy yz | zz)
#pragma unroll
                for (int j = 0; j < 3; j++) if (j >= r) o[4 + (r == 0 ? j : (r == 1 ? 2 + j : 5))] = Dr[j];
                bD0 = Dr[0]; bD1 = Dr[1]; bD2 = Dr[2]; bC0 = m * p[0]; bC1 = m * p[1]; bC2 = m * p[2]; bm = m;
            }
        }
        TG_SYNC();
        TG_STAMP(2);
        // ---- E4: constraint values and Dh2 at q2 (constraints(on, 2, true, Dh2, 0)) ----
        const ResTab rt = fetch_residual();
        if (on) {
            if (lane < P.nc) {
                const double *a = S + P.o_pE + 3 * ct.e1, *b = S + P.o_pE + 3 * ct.e2;
                const double vx = a[0] - b[0], vy = a[1] - b[1], vz = a[2] - b[2];
                double h;
                if (ct.type == TG_CONSTRAINT_POINT) h = ct.comp == 0 ? vx : (ct.comp == 1 ? vy : vz);
                else {
                    const double len = ct.cfg >= 0 ? S[P.o_q2 + (ct.cfg >= 0 ? ct.cfg : 0)] : ct.dist;
                    h = (vx * vx + vy * vy + vz * vz) - len * len;
                }
                S[P.o_f + P.nd + lane] = h;
                res_hoff = fabs(h) > S[P.o_ctol + lane];
            } else res_hoff = false;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (u * TEAM < P.n_dhr && lane + u * TEAM < P.n_dhr) {
                    const int n = ct.rec[u][0];
                    const int k = ct.rec[u][1], oj = ct.rec[u][2], w = ct.rec[u][3], oe1 = ct.rec[u][4], oe2 = ct.rec[u][5];
                    const int side = w & 0xFF, kind = (w >> 8) & 0xFF, type = (w >> 16) & 0xFF, comp = w >> 24;
                    const int ojc = oj < 0 ? 0 : oj;
                    double d1[3], d2[3];
                    const double *gj = G2 + ojc;
                    const bool prismatic = kind <= TG_TZ;
                    const int ax = prismatic ? kind - TG_TX : kind - TG_RX;
                    const double wx = gj[ax], wy = gj[4 + ax], wz = gj[8 + ax];
                    {
                        const double *pe = S + P.o_pE + oe1;
                        const double dx = pe[0] - gj[3], dy = pe[1] - gj[7], dz = pe[2] - gj[11];
                        d1[0] = prismatic ? wx : wy * dz - wz * dy; d1[1] = prismatic ? wy : wz * dx - wx * dz; d1[2] = prismatic ? wz : wx * dy - wy * dx;
                    }
                    {
                        const double *pe = S + P.o_pE + oe2;
                        const double dx = pe[0] - gj[3], dy = pe[1] - gj[7], dz = pe[2] - gj[11];
                        d2[0] = prismatic ? wx : wy * dz - wz * dy; d2[1] = prismatic ? wy : wz * dx - wx * dz; d2[2] = prismatic ? wz : wx * dy - wy * dx;
                    }
                    const double s1 = (side & 1) ? 1.0 : 0.0, s2 = (side & 2) ? 1.0 : 0.0;
                    const double dx = s1 * d1[0] - s2 * d2[0], dy = s1 * d1[1] - s2 * d2[1], dz = s1 * d1[2] - s2 * d2[2];
                    double val;
                    if (type == TG_CONSTRAINT_POINT) val = comp == 0 ? dx : (comp == 1 ? dy : dz);
                    else {
                        const double *a = S + P.o_pE + oe1, *b = S + P.o_pE + oe2;
                        val = (a[0] - b[0]) * dx + (a[1] - b[1]) * dy + (a[2] - b[2]) * dz;
                        if (side & 4) val -= S[P.o_q2 + k];
                        val *= 2.0;
                    }
                    S[P.o_Dh2 + n] = val;
                }
            }
            // ---- E4, second half: the lane's list sum (every lane: a lane without a role sums the zero record); w_k = [V_k^-, s_k] (config
            //      lanes), the body's momentum about the world origin f = M v - C x omega, tau = C x v + D omega (body lanes: rows r)
            {
                double V[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                constexpr int HALF = (MAXD + 1) / 2;
                wev_sum<0, HALF>(SW, V);
                wev_sum<HALF, MAXD>(SW, V);
                if (lane < nd) {
                    double sk[6];
                    ld6<true>(SW + 12 * lane, sk);
                    bracket(V, sk, wev_w);
                } else if (blane) {
                    const int r = wr_;
                    const double cw0 = bC1 * V[5] - bC2 * V[4], cw1 = bC2 * V[3] - bC0 * V[5], cw2 = bC0 * V[4] - bC1 * V[3];   // C x omega
                    const double cv0 = bC1 * V[2] - bC2 * V[1], cv1 = bC2 * V[0] - bC0 * V[2], cv2 = bC0 * V[1] - bC1 * V[0];   // C x v
                    const double fr = bm * (r == 0 ? V[0] : (r == 1 ? V[1] : V[2])) - (r == 0 ? cw0 : (r == 1 ? cw1 : cw2));
                    const double tr = (r == 0 ? cv0 : (r == 1 ? cv1 : cv2)) + (bD0 * V[3] + bD1 * V[4] + bD2 * V[5]);
                    double *o = BW + BWS * wb_;
                    o[10 + r] = fr;
                    o[13 + r] = tr;
                }
            }
        }
        TG_SYNC();
        TG_STAMP(6);
        // ---- E5: composites of the subtree groups (lane = entry; membership is compile-time).  The union of poses and Newton image is dead
        //      from here to the next evaluation (its last readers were the constraint lanes of E4): the image is cleared here, by lanes
        //      that would idle, instead of at the head of the assembly (wasted by a step's last evaluation: one in four)
        {
            typedef double tg_d2 __attribute__((ext_vector_type(2)));
            static_assert(((SP::o_Df | (SP::nf * SP::df_ld)) & 1) == 0, "eval_world: image not 16-byte aligned");
            tg_d2 *A2 = reinterpret_cast<tg_d2 *>(S + P.o_Df);
            const tg_d2 z2 = {0.0, 0.0};
#if defined(TG_WEV_CLEAR_EARLY)
            if (on) TG_FOR(i, (SP::nf * SP::df_ld) >> 1) A2[i] = z2;
#else
            (void)A2; (void)z2;
#endif
        }
        if (on && lane < 16) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                double acc = 0.0;
#pragma unroll
                for (int b = 0; b < NB; b++) if ((P.cmp_gmask[g] >> b) & 1) acc += BW[BWS * b + lane];
                CMP[16 * g + lane] = acc;
            }
        }
        TG_SYNC();
        TG_STAMP(3);
        // ---- E6: L_dq, L_ddq and the residual entry of config k; and, from the same operands, what the Newton matrix's pair lanes need of
        //      config k: I s_k, Z_k = Y_k + I w_k, GG_k (phase C of newton_matrix_composite; one evaluation in four does not use them) ----
        if (on && lane < nd) {
            const double *c = CMP + 16 * ((wvl[3] >> 24) & 0x7F);
            const double M = c[0], Cx = c[1], Cy = c[2], Cz = c[3], Dxx = c[4], Dxy = c[5], Dxz = c[6], Dyy = c[7], Dyz = c[8], Dzz = c[9];
            const double h0 = c[10], h1 = c[11], h2 = c[12], h3 = c[13], h4 = c[14], h5 = c[15];
            double sk[6];
            ld6<true>(SW + 12 * lane, sk);
            const double *w = wev_w;
            const double lddq = sk[0] * h0 + sk[1] * h1 + sk[2] * h2 + sk[3] * h3 + sk[4] * h4 + sk[5] * h5;
            const double Gx = M * sk[0] + (sk[4] * Cz - sk[5] * Cy), Gy = M * sk[1] + (sk[5] * Cx - sk[3] * Cz), Gz = M * sk[2] + (sk[3] * Cy - sk[4] * Cx);
            const double gx = P.grav[0], gy = P.grav[1], gz = P.grav[2];
            const double ldq = (w[0] * h0 + w[1] * h1 + w[2] * h2 + w[3] * h3 + w[4] * h4 + w[5] * h5) + (gx * Gx + gy * Gy + gz * Gz);
            const int i = lane;
            S[P.o_Ldq + i] = ldq; S[P.o_Lddq + i] = lddq;
            double force = -tdamp * S[P.o_dq + i];
            for (int k = 0; k < P.n_cf; k++) if (P.cf_cfg[k] == i) force += S[P.o_u + P.cf_in[k]];
            double f = S[P.o_p1 + i] + (0.5 * dt * ldq - lddq) + dt * force;
#pragma unroll
            for (int cc = 0; cc < 8; cc++) {
                if (cc < P.nc) {
                    const int n = rt.look[cc];
                    f -= (n >= 0 ? S[P.o_lam + cc] : 0.0) * S[P.o_Dh1 + (n >= 0 ? n : 0)];
                }
            }
            S[P.o_f + i] = f;
            res_f2 = f * f;
#if defined(TG_WEV_MERGE_C)
            auto apply = [&](const double *x, double *y) {      // spatial inertia times twist
                y[0] = M * x[0] - (Cy * x[5] - Cz * x[4]); y[1] = M * x[1] - (Cz * x[3] - Cx * x[5]); y[2] = M * x[2] - (Cx * x[4] - Cy * x[3]);
                y[3] = (Cy * x[2] - Cz * x[1]) + Dxx * x[3] + Dxy * x[4] + Dxz * x[5];
                y[4] = (Cz * x[0] - Cx * x[2]) + Dxy * x[3] + Dyy * x[4] + Dyz * x[5];
                y[5] = (Cx * x[1] - Cy * x[0]) + Dxz * x[3] + Dyz * x[4] + Dzz * x[5];
            };
            double Is[6], Iw[6];
            apply(sk, Is); apply(w, Iw);
            const double b0 = sk[0], b1 = sk[1], b2 = sk[2], b3 = sk[3], b4 = sk[4], b5 = sk[5];
            double Z[6];
            Z[0] = Iw[0] + (b4 * h2 - b5 * h1); Z[1] = Iw[1] + (b5 * h0 - b3 * h2); Z[2] = Iw[2] + (b3 * h1 - b4 * h0);
            Z[3] = Iw[3] + (b1 * h2 - b2 * h1) + (b4 * h5 - b5 * h4);
            Z[4] = Iw[4] + (b2 * h0 - b0 * h2) + (b5 * h3 - b3 * h5);
            Z[5] = Iw[5] + (b0 * h1 - b1 * h0) + (b3 * h4 - b4 * h3);
            double *o = S + P.o_ccz + 15 * lane;
#pragma unroll
            for (int r = 0; r < 6; r++) { o[r] = Is[r]; o[6 + r] = Z[r]; }
            o[12] = Gy * gz - Gz * gy; o[13] = Gz * gx - Gx * gz; o[14] = Gx * gy - Gy * gx;
            st6<true>(SW + 12 * lane + 6, wev_w);        // (the u-half of the record: its readers finished two barriers ago)
#else
            (void)Dxx; (void)Dxy; (void)Dxz; (void)Dyy; (void)Dyz; (void)Dzz;
#endif
        } else res_f2 = 0.0;
        TG_STAMP(4);
    }
#endif

    // ---- poses of the massive frames and positions of the constraint end points --------------------
    TG_HD void attach_points(bool on, bool bodies, bool endpoints) {
        PROG &P = tg_fresh(this->P);
        const Real *G = S + P.o_G;
        // Branch-free: an unanchored frame (anchor < 0: fixed to the world) reads joint 0 and weights it out.  A branch
        // would split the loop body into basic blocks that each wait for their own loads.
        if (on && bodies) TG_FOR(idx, 12 * P.n_bodies) {
            const int b = idx / 12, e = idx % 12, r = e >> 2, c = e & 3;
            const double *C = P.b_C + 12 * b;
            const int anchor = P.b_anchor[b];
            const double ce = C[e], c0 = C[c], c1 = C[4 + c], c2 = C[8 + c];
            const Real *g = G + 12 * (anchor < 0 ? 0 : anchor) + 4 * r;
            const Real g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
            const Real val = g0 * c0 + g1 * c1 + g2 * c2 + (c == 3 ? g3 : 0.0);
            S[P.o_gB + idx] = anchor < 0 ? ce : val;
        }
        if (on && endpoints) TG_FOR(idx, 3 * P.n_endpoints) {
            const int e = idx / 3, r = idx % 3;
            const double *o = P.e_off + 3 * e;
            const int anchor = P.e_anchor[e];
            const double o0 = o[0], o1 = o[1], o2 = o[2], orr = o[r];
            const Real *g = S + oGc + 12 * (anchor < 0 ? 0 : anchor) + 4 * r;
            const Real val = g[0] * o0 + g[1] * o1 + g[2] * o2 + g[3];
            S[P.o_pE + idx] = anchor < 0 ? orr : val;
        }
        if (on && endpoints && n_wrenches()) TG_FOR(idx, 9 * P.n_wrenches) {   // world rotation of every wrench frame
            const int w = idx / 9, r = (idx % 9) / 3, cc = idx % 3;
            const double *Rl = P.wr_Rloc + 9 * w;
            const int anchor = P.e_anchor[P.c_e1[P.nc + n_springs() + w]];
            const Real *g = G + 12 * (anchor < 0 ? 0 : anchor) + 4 * r;
            const Real val = g[0] * Rl[cc] + g[1] * Rl[3 + cc] + g[2] * Rl[6 + cc];
            S[P.o_wR + idx] = anchor < 0 ? Rl[3 * r + cc] : val;
        }
        if (on && endpoints && has_plane()) TG_FOR(idx, 3 * P.nc) {   // world normal of every plane constraint
            const int c = idx / 3, r = idx % 3;
            const double *nl = P.c_nloc + 3 * c;
            const int anchor = P.e_anchor[P.c_e1[c]];
            const Real *g = G + 12 * (anchor < 0 ? 0 : anchor) + 4 * r;
            const Real val = g[0] * nl[0] + g[1] * nl[1] + g[2] * nl[2];
            S[P.o_nE + idx] = anchor < 0 ? nl[r] : val;
        }
        TG_SYNC();
    }

    // ---- body Jacobian columns J_{F,k} and gravity in body coordinates ------------------------------
    TG_HD void jacobians(bool on) {
        PROG &P = tg_fresh(this->P);
        const Real *G = S + P.o_G;
        struct JacOut { Real J[6], dq; };
        if (on) for_pairs(P.n_items, [&](int it) {
            const int *rec = P.it_pack + 4 * (size_t)it;
            const int b = rec[0], oj = rec[1], kind = rec[2], cfg = rec[3] & 0xFFFF;
            const Real *gb = S + P.o_gB + 12 * b, *gj = G + oj;
            // branch-free: the joint axis column is read either way (column kind-TX or kind-RX of the joint pose)
            const bool prismatic = kind <= TG_TZ;
            const int ax = prismatic ? kind - TG_TX : kind - TG_RX;
            const Real a0 = gj[ax], a1 = gj[4 + ax], a2 = gj[8 + ax];
            const Real dx = gb[3] - gj[3], dy = gb[7] - gj[7], dz = gb[11] - gj[11];
            Real lin[3], ang[3];
            ang[0] = prismatic ? 0.0 : a0; ang[1] = prismatic ? 0.0 : a1; ang[2] = prismatic ? 0.0 : a2;
            lin[0] = prismatic ? a0 : a1 * dz - a2 * dy;
            lin[1] = prismatic ? a1 : a2 * dx - a0 * dz;
            lin[2] = prismatic ? a2 : a0 * dy - a1 * dx;
            JacOut o;
            for (int r = 0; r < 3; r++) {  // R_F^T applied to both parts
                o.J[r] = gb[r] * lin[0] + gb[4 + r] * lin[1] + gb[8 + r] * lin[2];
                o.J[3 + r] = gb[r] * ang[0] + gb[4 + r] * ang[1] + gb[8 + r] * ang[2];
            }
            o.dq = S[P.o_dq + cfg];  // rate of the item's config, for the prefix sums
            return o;
        }, [&](int it, const JacOut &o) {
            Real *J = S + P.o_J + 6 * it;
            for (int r = 0; r < 6; r++) J[r] = o.J[r];
            S[P.o_dqi + it] = o.dq;
        });
        if (on) TG_FOR(idx, 3 * P.n_bodies) {
            const int b = idx / 3, r = idx % 3;
            const Real *gb = S + P.o_gB + 12 * b;
            S[P.o_gam + idx] = gb[r] * P.grav[0] + gb[4 + r] * P.grav[1] + gb[8 + r] * P.grav[2];
        }
        TG_SYNC();
    }

    // ---- prefix velocities, W_j = [P_j, J_j], body velocity v_F --------------------------------------
    TG_HD void velocities(bool on) {
        PROG &P = tg_fresh(this->P);
        // (1) one lane per (body, twist component): serial prefix sum along the body's path,
        //     P_j = sum_{k<j} J_k dq_k written into the W slot of item j, total = body velocity;
        // (2) one lane per item: W_j = [P_j, J_j] in place.
        if (on) TG_FOR(idx, 6 * P.n_bodies) {
            const int b = idx / 6, m = idx % 6;
            const int first = P.b_item_off[b], last = P.b_item_off[b + 1];
            Real acc = 0.0;
            // four items per trip, their loads ahead of the stores (an LDS store may alias the next load as far as the compiler
            // knows, so a load-fma-store loop waits out the LDS latency once per item)
            for (int k = first; k < last; k += 4) {
                Real jv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = k + u < last ? k + u : last - 1;
                    jv[u] = S[P.o_J + 6 * kk + m]; dv[u] = S[P.o_dqi + kk];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (k + u < last) { S[P.o_W + 6 * (k + u) + m] = acc; acc = fma(jv[u], dv[u], acc); }
                }
            }
            S[P.o_vB + idx] = acc;
        }
        TG_SYNC();
        if (on) TG_FOR(it, P.n_items) {
            Real *W = S + P.o_W + 6 * it;
            const Real *J = S + P.o_J + 6 * it;
            const Real Pp[6] = {W[0], W[1], W[2], W[3], W[4], W[5]};
            bracket(Pp, J, W);
        }
        TG_SYNC();
    }

    // ---- L_dq, L_ddq per config and the dynamic part of the DEL residual (midpointvi.c:533-551) -------
    TG_HD void residual_dyn(bool on) {
        PROG &P = tg_fresh(this->P);
        // per-item terms <J,v> and <W,v> + m gam.Jv, stored in config-sorted order in the (now dead) joint
        // pose area, then one contiguous sum per dynamic config
        double *terms = S + P.o_G;
        struct TermOut { double a, b; int slot; };
        if (on) for_pairs(P.n_items, [&](int it) {
            const int b = P.it_pack[4 * (size_t)it], slot = P.it_pack[4 * (size_t)it + 3] >> 16;
            const double *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b;
            const double *J = S + P.o_J + 6 * it, *W = S + P.o_W + 6 * it, *gam = S + P.o_gam + 3 * b;
            TermOut o;
            o.a = inner6(I, J, v);
            o.b = inner6(I, W, v) + I[0] * (gam[0] * J[0] + gam[1] * J[1] + gam[2] * J[2]);
            o.slot = slot;
            return o;
        }, [&](int, const TermOut &o) { terms[2 * o.slot] = o.a; terms[2 * o.slot + 1] = o.b; });
        TG_SYNC();
        if (on) TG_FOR(i, P.nd) {
            double ldq = 0.0, lddq = 0.0;
            const int n1 = P.cfg_item_off[i + 1];
            for (int n = P.cfg_item_off[i]; n < n1; n++) { lddq += terms[2 * n]; ldq += terms[2 * n + 1]; }
            if (has_cs()) ldq -= cs_d1(i, qval(0, i));   // config springs (configspring.c:22-31, nonlinear_config_spring.c:24-34)
            if (n_springs()) ldq -= S[P.o_sV + i];
            S[P.o_Ldq + i] = ldq; S[P.o_Lddq + i] = lddq;
            double force = -P.damp[i] * S[P.o_dq + i];
            for (int k = 0; k < P.n_cf; k++) if (P.cf_cfg[k] == i) force += S[P.o_u + P.cf_in[k]];
            if (n_wrenches()) force += S[P.o_wF + i];
            if (has_damper()) force += S[P.o_sF + i];
            double f = S[P.o_p1 + i] + (0.5 * dt * ldq - lddq) + dt * force;
            for (int c = 0; c < P.nc; c++) {                 // branch-free: a missing entry reads item 0 with weight 0
                const int n = P.dh_lookup[c * P.nq + i];
                f -= (n >= 0 ? S[P.o_lam + c] : 0.0) * S[P.o_Dh1 + (n >= 0 ? n : 0)];
            }
            S[P.o_f + i] = f;
        }
        TG_SYNC();
    }

    // d p_E / d q_k in world coordinates
    TG_HD void dpos(int e, int j, Real *d) const {
        const Real *gj = S + P.o_G + 12 * j;
        const int kind = P.j_kind[j];
        if (kind <= TG_TZ) { const int a = kind - TG_TX; d[0] = gj[a]; d[1] = gj[4 + a]; d[2] = gj[8 + a]; }
        else {
            const int a = kind - TG_RX;
            const Real wx = gj[a], wy = gj[4 + a], wz = gj[8 + a];
            const Real *pe = S + P.o_pE + 3 * e;
            const Real dx = pe[0] - gj[3], dy = pe[1] - gj[7], dz = pe[2] - gj[11];
            d[0] = wy * dz - wz * dy; d[1] = wz * dx - wx * dz; d[2] = wx * dy - wy * dx;
        }
    }

    // same with the joint given by its pose offset and kind, the end point by its offset (packed dh records)
    TG_HD void dpos_rec(int oe, int oj, int kind, Real *d) const {
        const Real *gj = S + oGc + oj, *pe = S + P.o_pE + oe;
        const bool prismatic = kind <= TG_TZ;
        const int a = prismatic ? kind - TG_TX : kind - TG_RX;
        const Real wx = gj[a], wy = gj[4 + a], wz = gj[8 + a];
        const Real dx = pe[0] - gj[3], dy = pe[1] - gj[7], dz = pe[2] - gj[11];
        d[0] = prismatic ? wx : wy * dz - wz * dy;
        d[1] = prismatic ? wy : wz * dx - wx * dz;
        d[2] = prismatic ? wz : wx * dy - wy * dx;
    }

    // ---- constraint values (into f[nd..]) and Jacobian Dh (into dest) at the swept state -------------
    // distance.c:16-63, point.c:16-38.  `sel` picks the config vector for the length configs.
    TG_HD void constraints(bool on, int sel, bool want_h, Real *Dh, int ld) {
        PROG &P = tg_fresh(this->P);
        if (on && want_h) TG_FOR(c, P.nc) {
            const Real *a = S + P.o_pE + 3 * P.c_e1[c], *b = S + P.o_pE + 3 * P.c_e2[c];
            const Real vx = a[0] - b[0], vy = a[1] - b[1], vz = a[2] - b[2];
            Real h;
            if (has_plane() && P.c_type[c] == TG_CONSTRAINT_PLANE) { const Real *nw = S + P.o_nE + 3 * c; h = nw[0] * vx + nw[1] * vy + nw[2] * vz; }
            else if (P.c_type[c] == TG_CONSTRAINT_POINT) h = P.c_comp[c] == 0 ? vx : (P.c_comp[c] == 1 ? vy : vz);
            else {
                const Real len = P.c_cfg[c] >= 0 ? qval(sel, P.c_cfg[c]) : P.c_dist[c];
                h = (vx * vx + vy * vy + vz * vz) - len * len;
            }
            S[P.o_f + P.nd + c] = h;
        }
        if (on) TG_FOR(n, P.n_dh) {
            const int *rec = P.dh_pack + 8 * (size_t)n;   // one record instead of chained table look-ups
            const int c = rec[0], k = rec[1], oj = rec[2], w = rec[3], oe1 = rec[4], oe2 = rec[5];
            const int side = w & 0xFF, kind = (w >> 8) & 0xFF, type = (w >> 16) & 0xFF, comp = w >> 24;
            Real d1[3], d2[3];
            const int ojc = oj < 0 ? 0 : oj;                 // the length config drives no joint: both sides weighted out
            dpos_rec(oe1, ojc, kind, d1);
            dpos_rec(oe2, ojc, kind, d2);
            const Real s1 = (side & 1) ? 1.0 : 0.0, s2 = (side & 2) ? 1.0 : 0.0;
            const Real dx = s1 * d1[0] - s2 * d2[0], dy = s1 * d1[1] - s2 * d2[1], dz = s1 * d1[2] - s2 * d2[2];
            Real val;
            if (has_plane() && type == TG_CONSTRAINT_PLANE) {   // plane.c:28-45: (dR n) . (p1 - p2) + (R n) . d(p1 - p2)
                const Real *nw = S + P.o_nE + 3 * c, *a = S + P.o_pE + oe1, *b = S + P.o_pE + oe2, *gj = S + P.o_G + ojc;
                const bool turns = kind >= TG_RX && (side & 1);
                const int ax = kind >= TG_RX ? kind - TG_RX : 0;
                const Real wx = turns ? gj[ax] : 0.0, wy = turns ? gj[4 + ax] : 0.0, wz = turns ? gj[8 + ax] : 0.0;
                const Real n1x = wy * nw[2] - wz * nw[1], n1y = wz * nw[0] - wx * nw[2], n1z = wx * nw[1] - wy * nw[0];
                val = n1x * (a[0] - b[0]) + n1y * (a[1] - b[1]) + n1z * (a[2] - b[2]) + nw[0] * dx + nw[1] * dy + nw[2] * dz;
            }
            else if (type == TG_CONSTRAINT_POINT) val = comp == 0 ? dx : (comp == 1 ? dy : dz);
            else {
                const Real *a = S + P.o_pE + oe1, *b = S + P.o_pE + oe2;
                val = (a[0] - b[0]) * dx + (a[1] - b[1]) * dy + (a[2] - b[2]) * dz;
                if (side & 4) val -= qval(sel, k);
                val *= 2.0;
            }
            if (ld > 0) Dh[c * ld + k] = val; else Dh[n] = val;
        }
        TG_SYNC();
    }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_CMP)
    // ---- Newton matrix, inertial part in COMPOSITE form (system-specialised rollout kernels of full-wave teams) -----------------------
    // newton_matrix() below walks every (body, item a <= item b) pair -- 442 for the puppet -- and evaluates L_dqdq, L_ddqdq, L_ddqddq
    // (system.c:158-202, 294-334, 459-489) of that body for that pair from its body-frame Jacobian columns.  All bodies below config b
    // see the same WORLD-frame joint twists s_k = Ad(g_F) J_{F,k} and w_k = Ad(g_F) W_{F,k} ([.,.] commutes with Ad), so the sum over
    // the bodies of a config pair (a, b), a on the path to b, is a bilinear form in the composite quantities of the subtree below b:
    //     M = sum m_F,  C = sum m_F p_F,  D = sum R_F I_F R_F' - m_F [p_F]^2  (world-frame spatial inertia [[M, -[C]], [[C], D]]),
    //     H = sum Ad(g_F)^-T (I_F v_F)  (spatial momentum about the world origin).
    // With I s = (M v - C x w, C x v + D w) for a twist s = (v, w), Y_b the co-adjoint vector with [X, s_b] . H = X . Y_b, and
    // Z_b = Y_b + I w_b,  GG_b = (M v_b + w_b x C) x g:
    //     L_ddqddq(a, b) = s_a . I s_b,   L_dqdq(a, b) = w_a . Z_b + omega_a . GG_b,   L_ddqdq(dq a, q b) = s_a . Z_b,   L_ddqdq(dq b, q a) = w_a . I s_b
    // -- 27 FMAs per CONFIG pair (157 for the puppet: one per matrix entry pair, so no two lanes meet at an entry) after 16 numbers per
    // body, 16 per subtree group (9) and 27 per config.  Same matrix up to rounding; the scratch lives in the J / W area, which is dead
    // between the residual and the next evaluation (the structured solve's scratch follows in the same place).
    TG_HD void newton_matrix_composite(bool on) {
        PROG &P = tg_fresh(this->P);
        typedef typename std::remove_cv<PROG>::type SP;
        constexpr int nd = SP::nd, nf = SP::nf, ld = SP::df_ld, NB = SP::n_bodies, NG = SP::n_cgroups, NP = SP::n_cmpairs;
        constexpr int TP = (NP + TEAM - 1) / TEAM;
        static_assert(3 * NB <= TEAM && nd <= TEAM, "newton_matrix_composite: one trip per phase");
        // scratch: per-body world entries BW in the dead joint-pose area of the union (the body poses next to it are still needed; the whole
        // union becomes the matrix image two phases on); composites, world twists and per-config vectors in the J / W area
        double *A = S + P.o_Df, *CMP = S + P.o_cmp, *SW = S + P.o_csw, *CZ = S + P.o_ccz, *BW = S + P.o_sc;
        // record strides: 17 doubles per body (16 put every second body on the same LDS banks); the twists keep 12 per config -- 13 would be
        // conflict free (profiles/r04_lds_conflicts.txt) but turns the pair lanes' six 16-byte reads of a twist pair into twelve 8-byte ones:
        // measured slower (37.3 against 36.6 ms)
        constexpr int BWS = 17, SWS = 12;
        static_assert(BWS * NB <= SP::o_gB - SP::o_sc, "newton_matrix_composite: body entries do not fit the dead pose area");
        const int *rep = (const int *)(S + P.o_cmpt);
        // ---- phase A.  (i) lane (body b, axis r): the body's world entries -- M, C_r = m p_r, row r of D = R I R' + m (|p|^2 1 - p p'),
        //      f_r = (R m v_B)_r, tau_r = (R I w_B)_r + (p x f)_r -- written straight to BW (nothing there is read in this phase);
        if (on && lane < 3 * NB) {
            const int b = lane / 3, r = lane - 3 * b;
            const double *gb = S + P.o_gB + 12 * b, *I = S + P.o_I + 4 * b, *vb = S + P.o_vB + 6 * b;
            const double m = I[0], px = gb[3], py = gb[7], pz = gb[11], pr = gb[4 * r + 3];
            const double R0 = gb[4 * r], R1 = gb[4 * r + 1], R2 = gb[4 * r + 2];
            double f[3], t[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                f[c] = m * (gb[4 * c] * vb[0] + gb[4 * c + 1] * vb[1] + gb[4 * c + 2] * vb[2]);
                t[c] = gb[4 * c] * (I[1] * vb[3]) + gb[4 * c + 1] * (I[2] * vb[4]) + gb[4 * c + 2] * (I[3] * vb[5]);
            }
            const double tx = t[0] + (py * f[2] - pz * f[1]), ty = t[1] + (pz * f[0] - px * f[2]), tz = t[2] + (px * f[1] - py * f[0]);
            const double p2 = px * px + py * py + pz * pz;
            double *o = BW + BWS * b;
            if (r == 0) o[0] = m;
            o[1 + r] = m * pr;
            // D row r, columns j >= r: entries 4 + (xx xy xz | yy yz | zz)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double pj = j == 0 ? px : (j == 1 ? py : pz);
                const double d = R0 * I[1] * gb[4 * j] + R1 * I[2] * gb[4 * j + 1] + R2 * I[3] * gb[4 * j + 2] + m * ((j == r ? p2 : 0.0) - pr * pj);
                if (j >= r) o[4 + (r == 0 ? j : (r == 1 ? 2 + j : 5))] = d;
            }
            o[10 + r] = r == 0 ? f[0] : (r == 1 ? f[1] : f[2]);
            o[13 + r] = r == 0 ? tx : (r == 1 ? ty : tz);
        }
        //      (ii) lane k: world twists s_k = Ad(g_F) J, w_k = Ad(g_F) W of config k from its representative item (kept in registers: SW
        //      lies in the J / W area other lanes are still reading)
        double sw[12];
        {
            const int k = lane < nd ? lane : 0, rw = rep[k], it = rw & 0xFFF, b = (rw >> 12) & 0xFF;
            const double *gb = S + P.o_gB + 12 * b, *J = S + P.o_J + 6 * it, *W = S + P.o_W + 6 * it;
            const double px = gb[3], py = gb[7], pz = gb[11];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const double *X = h ? W : J;
                double v[3], w[3];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    v[r] = gb[4 * r] * X[0] + gb[4 * r + 1] * X[1] + gb[4 * r + 2] * X[2];
                    w[r] = gb[4 * r] * X[3] + gb[4 * r + 1] * X[4] + gb[4 * r + 2] * X[5];
                }
                // velocity of the point at the world origin: R v_B - w x p
                sw[6 * h + 0] = v[0] - (w[1] * pz - w[2] * py); sw[6 * h + 1] = v[1] - (w[2] * px - w[0] * pz); sw[6 * h + 2] = v[2] - (w[0] * py - w[1] * px);
                sw[6 * h + 3] = w[0]; sw[6 * h + 4] = w[1]; sw[6 * h + 5] = w[2];
            }
        }
        TG_SYNC();
        // the pair records of all trips: requested here (behind the phase that holds twelve twist components in registers), used two phases later
        int prec[TP];
#pragma unroll
        for (int u = 0; u < TP; u++) prec[u] = P.cmp_pair[lane + u * TEAM < NP ? lane + u * TEAM : 0];
        // ---- phase B: twists to SW; composites of the subtree groups (lane = entry; membership is compile-time) to CMP
        if (on && lane < nd) {
#pragma unroll
            for (int r = 0; r < 12; r++) SW[SWS * lane + r] = sw[r];
        }
        if (on && lane < 16) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                double acc = 0.0;
#pragma unroll
                for (int b = 0; b < NB; b++) if ((P.cmp_gmask[g] >> b) & 1) acc += BW[BWS * b + lane];
                CMP[16 * g + lane] = acc;
            }
        }
        TG_SYNC();
        TG_STAMP(7);
        // ---- phase C: the image (the whole union: poses and BW are dead now) is cleared; per config b: I s_b, Z_b = Y_b + I w_b, GG_b
        {
            typedef double tg_d2 __attribute__((ext_vector_type(2)));
            static_assert(((SP::o_Df | (nf * ld)) & 1) == 0, "newton_matrix_composite: image not 16-byte aligned");
            tg_d2 *A2 = reinterpret_cast<tg_d2 *>(A);
            const tg_d2 z2 = {0.0, 0.0};
            if (on) TG_FOR(i, (nf * ld) >> 1) A2[i] = z2;
        }
        if (on && lane < nd) {
            const double *c = CMP + 16 * (rep[lane] >> 20);
            const double M = c[0], Cx = c[1], Cy = c[2], Cz = c[3], Dxx = c[4], Dxy = c[5], Dxz = c[6], Dyy = c[7], Dyz = c[8], Dzz = c[9];
            const double h0 = c[10], h1 = c[11], h2 = c[12], h3 = c[13], h4 = c[14], h5 = c[15];
            const double *s = SW + SWS * lane, *w = s + 6;
            auto apply = [&](const double *x, double *y) {      // spatial inertia times twist
                y[0] = M * x[0] - (Cy * x[5] - Cz * x[4]); y[1] = M * x[1] - (Cz * x[3] - Cx * x[5]); y[2] = M * x[2] - (Cx * x[4] - Cy * x[3]);
                y[3] = (Cy * x[2] - Cz * x[1]) + Dxx * x[3] + Dxy * x[4] + Dxz * x[5];
                y[4] = (Cz * x[0] - Cx * x[2]) + Dxy * x[3] + Dyy * x[4] + Dyz * x[5];
                y[5] = (Cx * x[1] - Cy * x[0]) + Dxz * x[3] + Dyz * x[4] + Dzz * x[5];
            };
            double Is[6], Iw[6];
            apply(s, Is); apply(w, Iw);
            const double b0 = s[0], b1 = s[1], b2 = s[2], b3 = s[3], b4 = s[4], b5 = s[5];
            double Z[6];
            Z[0] = Iw[0] + (b4 * h2 - b5 * h1); Z[1] = Iw[1] + (b5 * h0 - b3 * h2); Z[2] = Iw[2] + (b3 * h1 - b4 * h0);
            Z[3] = Iw[3] + (b1 * h2 - b2 * h1) + (b4 * h5 - b5 * h4);
            Z[4] = Iw[4] + (b2 * h0 - b0 * h2) + (b5 * h3 - b3 * h5);
            Z[5] = Iw[5] + (b0 * h1 - b1 * h0) + (b3 * h4 - b4 * h3);
            // G = M v + w x C;  GG = G x g
            const double Gx = M * b0 + (b4 * Cz - b5 * Cy), Gy = M * b1 + (b5 * Cx - b3 * Cz), Gz = M * b2 + (b3 * Cy - b4 * Cx);
            const double gx = P.grav[0], gy = P.grav[1], gz = P.grav[2];
            double *o = CZ + 15 * lane;
#pragma unroll
            for (int r = 0; r < 6; r++) { o[r] = Is[r]; o[6 + r] = Z[r]; }
            o[12] = Gy * gz - Gz * gy; o[13] = Gz * gx - Gx * gz; o[14] = Gx * gy - Gy * gx;
        }
        TG_SYNC();
        // ---- phase D: the constant entries (right-hand side, damping, -Dh1' / Dh2) and the config pairs: one lane per pair, one pair per
        //      entry and its transpose (the damping joins the diagonal pair's sum as an LDS add of its own: same wave, fixed order)
        if (on) {
            if (lane < nf) A[lane * ld + nf] = S[P.o_f + lane];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int n = tck[u][0] >> 8, c = tck[u][0] & 0xFF, k = tck[u][1];
                if (u * TEAM < P.n_dhr && lane + u * TEAM < P.n_dhr) { A[k * ld + nd + c] = -S[P.o_Dh1 + n]; A[(nd + c) * ld + k] = S[P.o_Dh2 + n]; }
            }
        }
        const double qdt = 0.25 * dt, rdt = inv_dt;
#pragma unroll
        for (int u = 0; u < TP; u++) {
            if (on && lane + u * TEAM < NP) {
                const int a = prec[u] & 0xFFFF, b = prec[u] >> 16;
                const double *sa = SW + SWS * a, *wa = sa + 6, *Is = CZ + 15 * b, *Z = Is + 6, *GG = Is + 12;
                double s_[6], w_[6], i_[6], z_[6];
#pragma unroll
                for (int r = 0; r < 6; r++) { s_[r] = sa[r]; w_[r] = wa[r]; i_[r] = Is[r]; z_[r] = Z[r]; }
                const double g0 = GG[0], g1 = GG[1], g2 = GG[2];
                double mab = 0.0, lqq = s_[3] * g0 + s_[4] * g1 + s_[5] * g2, cab = 0.0, cba = 0.0;
#pragma unroll
                for (int r = 0; r < 6; r++) { mab = fma(s_[r], i_[r], mab); lqq = fma(w_[r], z_[r], lqq); cab = fma(s_[r], z_[r], cab); cba = fma(w_[r], i_[r], cba); }
                const double sym = qdt * lqq - rdt * mab, skew = 0.5 * (cba - cab);
                // one lane per entry and its transpose: plain stores.  The first nd pairs are the diagonal ones in config order (program.hpp),
                // so lane c of the first trip adds config c's damping from its own table row
                if (u == 0 && a == b) A[a * ld + a] = sym - tdamp;
                else { A[a * ld + b] = sym + (a != b ? skew : 0.0); if (a != b) A[b * ld + a] = sym - skew; }
            }
        }
        TG_SYNC();
        TG_STAMP(8);
    }

    // The Newton matrix after eval_world: phases C and D of newton_matrix_composite -- the twists s_k are in LDS already, w_k in the lane's
    // registers, the composites in CMP -- so the two phases that re-derived them from the body-frame items (A, B) are gone.
    // PK: the image in the structured solve's own order (bbd.hpp, BbdPacked) -- every entry goes to its one place there, from the lane's
    // table rows; !PK: the dense image [nf][ld], which the pivoting fallback solves.  SKIPC: the per-config vectors are in place already
    // (the dense re-assembly after a failed structured solve).
    template <bool PK, bool SKIPC = false> TG_HD void newton_matrix_world(bool on) {
        PROG &P = tg_fresh(this->P);
        typedef typename std::remove_cv<PROG>::type SP;
        constexpr int nd = SP::nd, nf = SP::nf, ld = SP::df_ld, NP = SP::n_cmpairs;
        constexpr int NCLEAR = PK ? SP::bbd_pk_size : nf * ld;
        constexpr int TP = (NP + TEAM - 1) / TEAM;
        double *A = S + P.o_Df, *SW = S + P.o_csw, *CZ = S + P.o_ccz;
        constexpr int SWS = 12;
        // phase C: the image is cleared, per config b: I s_b, Z_b = Y_b + I w_b, GG_b (measured variants, both slower: -DTG_WEV_MERGE_C forms the per-config vectors in
        // eval_world's last phase -- 30.4 against 30.0 ms: every evaluation then pays for them --, -DTG_WEV_CLEAR_EARLY clears the image in E5)
#if !defined(TG_WEV_CLEAR_EARLY)
        {
            typedef double tg_d2 __attribute__((ext_vector_type(2)));
            static_assert((NCLEAR & 1) == 0, "newton_matrix_world: image size");
            tg_d2 *A2 = reinterpret_cast<tg_d2 *>(A);
            const tg_d2 z2 = {0.0, 0.0};
            if (on) TG_FOR(i, NCLEAR >> 1) A2[i] = z2;
        }
#endif
#if !defined(TG_WEV_MERGE_C)
        if (!SKIPC && on && lane < nd) {
            const double *c = S + P.o_cmp + 16 * ((wvl[3] >> 24) & 0x7F);
            const double M = c[0], Cx = c[1], Cy = c[2], Cz = c[3], Dxx = c[4], Dxy = c[5], Dxz = c[6], Dyy = c[7], Dyz = c[8], Dzz = c[9];
            const double h0 = c[10], h1 = c[11], h2 = c[12], h3 = c[13], h4 = c[14], h5 = c[15];
            double s[6];
            ld6<true>(SW + SWS * lane, s);
            const double *w = wev_w;
            auto apply = [&](const double *x, double *y) {
                y[0] = M * x[0] - (Cy * x[5] - Cz * x[4]); y[1] = M * x[1] - (Cz * x[3] - Cx * x[5]); y[2] = M * x[2] - (Cx * x[4] - Cy * x[3]);
                y[3] = (Cy * x[2] - Cz * x[1]) + Dxx * x[3] + Dxy * x[4] + Dxz * x[5];
                y[4] = (Cz * x[0] - Cx * x[2]) + Dxy * x[3] + Dyy * x[4] + Dyz * x[5];
                y[5] = (Cx * x[1] - Cy * x[0]) + Dxz * x[3] + Dyz * x[4] + Dzz * x[5];
            };
            double Is[6], Iw[6];
            apply(s, Is); apply(w, Iw);
            const double b0 = s[0], b1 = s[1], b2 = s[2], b3 = s[3], b4 = s[4], b5 = s[5];
            double Z[6];
            Z[0] = Iw[0] + (b4 * h2 - b5 * h1); Z[1] = Iw[1] + (b5 * h0 - b3 * h2); Z[2] = Iw[2] + (b3 * h1 - b4 * h0);
            Z[3] = Iw[3] + (b1 * h2 - b2 * h1) + (b4 * h5 - b5 * h4);
            Z[4] = Iw[4] + (b2 * h0 - b0 * h2) + (b5 * h3 - b3 * h5);
            Z[5] = Iw[5] + (b0 * h1 - b1 * h0) + (b3 * h4 - b4 * h3);
            const double Gx = M * b0 + (b4 * Cz - b5 * Cy), Gy = M * b1 + (b5 * Cx - b3 * Cz), Gz = M * b2 + (b3 * Cy - b4 * Cx);
            const double gx = P.grav[0], gy = P.grav[1], gz = P.grav[2];
            double *o = CZ + 15 * lane;
#pragma unroll
            for (int r = 0; r < 6; r++) { o[r] = Is[r]; o[6 + r] = Z[r]; }
            o[12] = Gy * gz - Gz * gy; o[13] = Gz * gx - Gx * gz; o[14] = Gx * gy - Gy * gx;
            st6<true>(SW + SWS * lane + 6, wev_w);
        }
#endif
        TG_SYNC();
        TG_STAMP(7);
        // ---- phase D: the constant entries (right-hand side, damping, -Dh1' / Dh2) and the config pairs: one lane per pair
        if (on) {
            if (PK && wone >= 0) A[wone] = 1.0;       // identity entries of the plan's padding rows (behind the clear: same lane order, same wave)
            if (lane < nf) A[PK ? wrhs : lane * ld + nf] = S[P.o_f + lane];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (u * TEAM < P.n_dhr && lane + u * TEAM < P.n_dhr) {
                    if constexpr (PK) { const int n = wdhx[u] & 0xFF; A[(wdhx[u] >> 8) & 0x3FF] = -S[P.o_Dh1 + n]; A[(wdhx[u] >> 18) & 0x3FF] = S[P.o_Dh2 + n]; }
                    else { const int n = tck[u][0] >> 8, c = tck[u][0] & 0xFF, k = tck[u][1]; A[k * ld + nd + c] = -S[P.o_Dh1 + n]; A[(nd + c) * ld + k] = S[P.o_Dh2 + n]; }
                }
            }
        }
        const double qdt = 0.25 * dt, rdt = inv_dt;
#pragma unroll
        for (int u = 0; u < TP; u++) {
            if (on && lane + u * TEAM < NP) {
                constexpr bool PKW = tg_static_pk<SP>::value;       // (which form the lane's pair records have, whichever image is written)
                const int a = PKW ? wpair[u] & 63 : wpair[u] & 0xFFFF, b = PKW ? (wpair[u] >> 6) & 63 : wpair[u] >> 16;
                const double *sa = SW + SWS * a, *wa = sa + 6, *Is = CZ + 15 * b, *Z = Is + 6, *GG = Is + 12;
                double s_[6], w_[6], i_[6], z_[6];
#pragma unroll
                for (int r = 0; r < 6; r++) { s_[r] = sa[r]; w_[r] = wa[r]; i_[r] = Is[r]; z_[r] = Z[r]; }
                const double g0 = GG[0], g1 = GG[1], g2 = GG[2];
                double mab = 0.0, lqq = s_[3] * g0 + s_[4] * g1 + s_[5] * g2, cab = 0.0, cba = 0.0;
#pragma unroll
                for (int r = 0; r < 6; r++) { mab = fma(s_[r], i_[r], mab); lqq = fma(w_[r], z_[r], lqq); cab = fma(s_[r], z_[r], cab); cba = fma(w_[r], i_[r], cba); }
                const double sym = qdt * lqq - rdt * mab, skew = 0.5 * (cba - cab);
                const int e_ab = PK ? (int)((unsigned)wpair[u] >> 12) & 0x3FF : a * ld + b, e_ba = PK ? (int)((unsigned)wpair[u] >> 22) & 0x3FF : b * ld + a;
                if (u == 0 && a == b) A[e_ab] = sym - tdamp;
                else { A[e_ab] = sym + (a != b ? skew : 0.0); if (a != b) A[e_ba] = sym - skew; }
            }
        }
        TG_SYNC();
        TG_STAMP(8);
    }
#endif

    // ---- Newton matrix [Df | f] (midpointvi.c:577-670) ---------------------------------------------------
    TG_HD void newton_matrix(bool on) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_CMP)
        if constexpr (TEAM == 64 && !SPRINGS && tg_static_wev<typename std::remove_cv<PROG>::type>::value) { if (wev_on) { if (tg_static_pk<typename std::remove_cv<PROG>::type>::value && pk_image) newton_matrix_world<true>(on); else newton_matrix_world<false>(on); return; } }
        if constexpr (TEAM == 64 && !SPRINGS && tg_static_cmp<typename std::remove_cv<PROG>::type>::value) { newton_matrix_composite(on); return; }
#endif
        PROG &P = tg_fresh(this->P);
        const int nd = P.nd, nf = P.nf, ld = P.df_ld;
        double *A = S + P.o_Df;
        // zero fill, then the few structurally non-zero constant entries: damping on the diagonal
        // (damping.c:21-27), the right-hand side f, and -Dh1^T / Dh2 from the (constraint, config) items
#if defined(__HIP_DEVICE_COMPILE__)
        if (TEAM == 64 && ((P.o_Df | (nf * ld)) & 1) == 0) {   // 16-byte stores: half the LDS instructions of the fill (the image starts on an even offset of a slice at LDS address 0)
            typedef double tg_d2 __attribute__((ext_vector_type(2)));
            tg_d2 *A2 = reinterpret_cast<tg_d2 *>(A);
            const tg_d2 z2 = {0.0, 0.0};
            if (on) TG_FOR(i, (nf * ld) >> 1) A2[i] = z2;
        } else
#endif
        if (on) TG_FOR(i, nf * ld) A[i] = 0.0;
        TG_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
        if (TEAM == 64 && P.tab_ok) {    // table rows held in registers since the kernel started (init_sweep_schedule)
            if (on) {
                if (lane < nf) {
                    A[lane * ld + nf] = S[P.o_f + lane];
                    if (lane < nd) A[lane * ld + lane] = -tdamp - (has_cs() ? 0.25 * dt * cs_d2(lane, qval(0, lane)) : 0.0);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int n = tck[u][0] >> 8, c = tck[u][0] & 0xFF, k = tck[u][1];
                    if (u * TEAM < P.n_dhr && lane + u * TEAM < P.n_dhr) { A[k * ld + nd + c] = -S[P.o_Dh1 + n]; A[(nd + c) * ld + k] = S[P.o_Dh2 + n]; }
                }
            }
        } else
#endif
        if (on) {
            TG_FOR(r, nf) {
                A[r * ld + nf] = S[P.o_f + r];
                if (r < nd) A[r * ld + r] = -P.damp[r] - (has_cs() ? 0.25 * dt * cs_d2(r, qval(0, r)) : 0.0);   // + dt/4 (-V_dqdq)
            }
            TG_FOR(n, P.n_dh) {
                const int c = P.dh_pack[8 * (size_t)n], k = P.dh_pack[8 * (size_t)n + 1];
                if (k < nd) { A[k * ld + nd + c] = -S[P.o_Dh1 + n]; A[(nd + c) * ld + k] = S[P.o_Dh2 + n]; }
            }
        }
        TG_SYNC();
        TG_STAMP(7);
        const double qdt = 0.25 * dt, rdt = 1.0 / dt;
        // All (item,item) pairs of all bodies in one flat pass; several bodies contribute to the same
        // matrix entry, so the accumulation uses LDS floating-point atomics (ds_add_f64).  One wavefront
        // owns the trajectory and its LDS operations retire in order, so the summation order -- and with
        // it the result -- is the same on every run.
        if (on) for (int pp = tg_opaque(lane), r0 = 0, r1 = 0, r2 = 0, r3 = 0, first_ = 1; pp < P.n_npairs; pp += TEAM) {
            // the record of the next trip is fetched while this trip computes (a table look-up per trip otherwise sits in
            // front of the trip's LDS reads)
            if (first_) {
                if (TEAM == 64 && P.tab_ok) { r0 = tpair[0]; r1 = tpair[1]; r2 = tpair[2]; r3 = tpair[3]; }
                else { const int *p0 = P.pair4 + 4 * (size_t)pp; r0 = p0[0]; r1 = p0[1]; r2 = p0[2]; r3 = p0[3]; }
                first_ = 0;
            }
            const int ia = r0, ib = r1, ca = r2 & 0xFFFF, cb = r2 >> 16, b = r3;
            {
                const int pn = pp + TEAM < P.n_npairs ? pp + TEAM : pp;
                const int *p1 = P.pair4 + 4 * (size_t)pn;
                r0 = p1[0]; r1 = p1[1]; r2 = p1[2]; r3 = p1[3];
            }
            const double *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b;
            const double *Ja = S + P.o_J + 6 * ia, *Jb = S + P.o_J + 6 * ib;
            const double *Wa = S + P.o_W + 6 * ia, *Wb = S + P.o_W + 6 * ib;
            double t[6];
            bracket(Wa, Jb, t);
            // L_dqdq(a,b) (system.c:158-202) with -V_dqdq = m gam . (w_a x v_b)
            const double lqq = inner6(I, t, v) + inner6(I, Wa, Wb) +
                               I[0] * (gam[0] * (Ja[4] * Jb[2] - Ja[5] * Jb[1]) + gam[1] * (Ja[5] * Jb[0] - Ja[3] * Jb[2]) +
                                       gam[2] * (Ja[3] * Jb[1] - Ja[4] * Jb[0]));
            const double mab = inner6(I, Ja, Jb);  // L_ddqddq (system.c:459-489)
            const double sym = qdt * lqq - rdt * mab;
            bracket(Ja, Jb, t);                    // zero when a == b
            const double c_ab = inner6(I, t, v) + inner6(I, Ja, Wb);  // L_ddqdq(dq a, q b) (system.c:294-334)
            const double c_ba = inner6(I, Jb, Wa);                    // L_ddqdq(dq b, q a)
            const double skew = 0.5 * (c_ba - c_ab);                  // exactly 0 for a == b
            lds_add(&A[ca * ld + cb], sym + skew);
            if (ia != ib) lds_add(&A[cb * ld + ca], sym - skew);
        }
        // (`SPRINGS &&`: the laundered lane index keeps a loop with a compile-time-zero bound alive, so compile it out here)
        if (SPRINGS && on) TG_FOR(pp, n_wpair()) {   // D2 fm2 = dt/2 F_dq of the point forces (midpointvi.c:593-600)
            const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + n_spair() + pp);
            const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
            if (ka >= nd || kb >= nd) continue;
            lds_add(&A[ka * ld + kb], 0.5 * dt * S[P.o_wH + 2 * pp]);
            if (pw[1] != pw[2]) lds_add(&A[kb * ld + ka], 0.5 * dt * S[P.o_wH + 2 * pp + 1]);
        }
        if (SPRINGS && on) TG_FOR(pp, n_spair()) {   // dt/4 (-V_dqdq) of the two-point springs, from the midpoint evaluation
            const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
            const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
            if (ka >= nd || kb >= nd) continue;
            double a_ = -qdt * S[P.o_sH + pp], b_ = a_;
            if (has_damper()) {   // D2 fm2 = dt/2 F_dq + F_ddq
                double fab, fba, fdd;
                damper_pair(pp, fab, fba, fdd);
                a_ += 0.5 * dt * fab + fdd; b_ += 0.5 * dt * fba + fdd;
            }
            lds_add(&A[ka * ld + kb], a_);
            if (pw[1] != pw[2]) lds_add(&A[kb * ld + ka], b_);
        }
        TG_SYNC();
        TG_STAMP(8);
    }

    // ---- Gauss-Jordan on [A | rhs] with implicit-scaled partial pivoting (pivot rule of
    //      math-code.c:337-432).  n_rhs right-hand-side columns follow the n matrix columns.
    //      Returns false (team-uniform) if a scaled pivot is <= 1e-20.  Solution left in the rhs
    //      columns divided through, i.e. A[i][n + r] = x_i.
    //      Lane mapping is 2-D (row group x power-of-two column tile) so the inner loops need no
    //      integer division; the pivot is an arg-max butterfly over the team.
    TG_HD bool gauss_jordan(bool on, Real *A, int n, int n_rhs, int ld, Real *scal) {
        bool ok = true;
        if (on) TG_FOR(i, n) {
            Real s = -1.0;
            for (int j = 0; j < n; j++) { const Real a = fabs(A[i * ld + j]); if (a > s) s = a; }
            scal[i] = 1.0 / s;
        }
        TG_SYNC();
        TG_STAMP(9);
        const int w = n + n_rhs;
        for (int k = 0; k < n; k++) {
            double best = -1.0;
            int piv = k;
            if (on && ok) {
                for (int i = k + lane; i < n; i += TEAM) {
                    const double a = tgdual::primal(fabs(A[i * ld + k] * scal[i]));
                    if (a > best) { best = a; piv = i; }
                }
            }
            team_argmax<TEAM>(best, piv);
            if (on && ok && !(best > 1.0e-20)) ok = false;
            const bool go = on && ok;
            if (go && piv != k) {
                for (int j = k + lane; j < w; j += TEAM) {
                    const Real t = A[k * ld + j]; A[k * ld + j] = A[piv * ld + j]; A[piv * ld + j] = t;
                }
                if (lane == 0) scal[piv] = scal[k];
            }
            TG_SYNC();
            TG_STAMP(10);
            if (go) {
                const int cols = w - k - 1;              // columns k+1 .. w-1 are updated
                const int cwl = tile_log2<TEAM>(cols);   // column tile = 2^cwl lanes, rows share the rest
                const int cw = 1 << cwl, rstep = TEAM >> cwl;
                const int jc = lane & (cw - 1);
                const Real rinv = 1.0 / A[k * ld + k];
                // A lane keeps its column(s): the pivot-row entry is loaded once.  Rows are taken four at a time with
                // all LDS loads issued before the first store: the compiler must assume that a store may alias the
                // next load, so a load-store-load-store sequence would pay the full LDS latency per row.
                for (int j = k + 1 + jc; j < w; j += cw) {
                    const Real pk = A[k * ld + j];
                    const int i0 = lane >> cwl;
                    int i = i0;
                    for (; i + 3 * rstep < n; i += 4 * rstep) {
                        const int r0 = i, r1 = i + rstep, r2 = i + 2 * rstep, r3 = i + 3 * rstep;
                        const Real l0 = A[r0 * ld + k], l1 = A[r1 * ld + k], l2 = A[r2 * ld + k], l3 = A[r3 * ld + k];
                        const Real a0 = A[r0 * ld + j], a1 = A[r1 * ld + j], a2 = A[r2 * ld + j], a3 = A[r3 * ld + j];
                        if (r0 != k) A[r0 * ld + j] = fma(-(l0 * rinv), pk, a0);
                        if (r1 != k) A[r1 * ld + j] = fma(-(l1 * rinv), pk, a1);
                        if (r2 != k) A[r2 * ld + j] = fma(-(l2 * rinv), pk, a2);
                        if (r3 != k) A[r3 * ld + j] = fma(-(l3 * rinv), pk, a3);
                    }
                    for (; i < n; i += rstep) {
                        const Real l = A[i * ld + k] * rinv, a = A[i * ld + j];
                        if (i != k) A[i * ld + j] = fma(-l, pk, a);
                    }
                }
            }
            TG_SYNC();
            TG_STAMP(11);
        }
        // divide the right-hand sides through by the pivots: reciprocals first (one lane per row), then every lane
        // scales its column(s) with the loads of four rows in flight
        if (on && ok) for (int i = lane; i < n; i += TEAM) scal[i] = 1.0 / A[i * ld + i];
        TG_SYNC();
        if (on && ok) {
            const int cwl = tile_log2<TEAM>(n_rhs), cw = 1 << cwl, rstep = TEAM >> cwl;
            for (int j = n + (lane & (cw - 1)); j < w; j += cw) {
                int i = lane >> cwl;
                for (; i + 3 * rstep < n; i += 4 * rstep) {
                    const int r0 = i, r1 = i + rstep, r2 = i + 2 * rstep, r3 = i + 3 * rstep;
                    const Real d0 = scal[r0], d1 = scal[r1], d2 = scal[r2], d3 = scal[r3];
                    const Real a0 = A[r0 * ld + j], a1 = A[r1 * ld + j], a2 = A[r2 * ld + j], a3 = A[r3 * ld + j];
                    A[r0 * ld + j] = a0 * d0; A[r1 * ld + j] = a1 * d1; A[r2 * ld + j] = a2 * d2; A[r3 * ld + j] = a3 * d3;
                }
                for (; i < n; i += rstep) A[i * ld + j] *= scal[i];
            }
        }
        TG_SYNC();
        return ok;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // ---- Gauss-Jordan with one matrix ROW PER LANE held in registers (n <= N <= TEAM) ---------------------
    //      Same pivot rule as gauss_jordan() but pivoting "in place": rows never move, the lane that owns
    //      the pivot row of step k broadcasts it (v_readlane for a full-wave team, ds_bpermute otherwise)
    //      and every other lane eliminates in registers.  No LDS traffic inside the k loop.  N is the
    //      matrix size rounded up to a multiple of 4 (identity padding), so every loop bound is a
    //      compile-time constant and the body carries no guards.
    //      Reads [A | rhs(1 column)] from LDS, leaves x in A[i*ld + n] like gauss_jordan().
    // Slow path of the pivot search (out of line: the 28-times unrolled solver must stay small enough for the instruction
    // cache): exact maximum of the candidates' doubles and, among the rows that attain it, the first in the reference's order.
    static __device__ __noinline__ int pivot_exact(double cand64, bool cand_ok, int pos, int lane) {
        unsigned long long best = cand_ok ? (unsigned long long)__double_as_longlong(cand64) : 0ull;   // non-negative doubles order like their bits
        if (TEAM == 64) {
            best = __ockl_wfred_max_u64(best);
        } else {
#pragma unroll
            for (int m = TEAM / 2; m >= 1; m >>= 1) {
                const unsigned long long o = __shfl_xor(best, m, TEAM);
                best = o > best ? o : best;
            }
        }
        const bool at_max = cand_ok && (unsigned long long)__double_as_longlong(cand64) == best;
        unsigned int k2 = at_max ? ((unsigned int)(63 - pos) << 6) | (unsigned int)(lane & 63) : 0u;   // position first, lane to identify the row
        if (TEAM == 64) {
            k2 = __ockl_wfred_max_u32(k2);
        } else {
#pragma unroll
            for (int m = TEAM / 2; m >= 1; m >>= 1) {
                const unsigned int o = __shfl_xor(k2, m, TEAM);
                k2 = o > k2 ? o : k2;
            }
        }
        int piv = (int)(k2 & 0x3Fu);
        if (TEAM != 64) piv = (piv & (TEAM - 1));
        return piv;
    }

    template <int N, bool TRACE = false>
    static __device__ __noinline__ bool gj_rows_exact(bool on, double *A_generic, int n, int ld, int lane, int *trace = nullptr) {
        typedef __attribute__((address_space(3))) double lds_double;
        lds_double *A = (lds_double *)A_generic;
        double row[N], rhs = 0.0, scale = 0.0, diag = 1.0;
        int mycol = -1;
        // position of this lane's row in the reference's row order (math-code.c swaps rows physically; here rows never move):
        // only needed to break EXACT ties the way the reference's strict `>` scan does -- first row in its current order
        int pos = lane;
        const bool mine = on && lane < N;
        const int wl = (int)(threadIdx.x & 63u), team_base = wl - lane;
#pragma unroll
        for (int j = 0; j < N; j++)
            row[j] = (mine && lane < n && j < n) ? A[lane * ld + j] : ((mine && lane >= n && j == lane) ? 1.0 : 0.0);
        if (mine) {
            rhs = lane < n ? A[lane * ld + n] : 0.0;
            double s = -1.0;
#pragma unroll
            for (int j = 0; j < N; j++) { const double a = fabs(row[j]); s = a > s ? a : s; }
            scale = 1.0 / s;
        }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; k++) {
            // arg-max of |a_ik| * scale_i over the rows not yet used as pivots, ties to the row that comes first in the
            // reference's (swapped) row order -- its scan uses a strict `>`.  The candidates are ranked by ONE 32-bit wave max
            // of (single-precision magnitude with the 6 low mantissa bits replaced by 63 - position): cast and mask are
            // monotonic, so the exact fp64 maximum is among the lanes that attain the truncated maximum, and among EXACTLY
            // equal candidates (mirror-symmetric mechanisms produce them all the time) the key already prefers the smallest
            // position.  Only if several lanes share the truncated maximum with DIFFERENT doubles (within 2^-17 relative, rare)
            // the slow path compares the doubles exactly.
            const bool cand_ok = mine && mycol < 0;
            const double cand64 = cand_ok ? fabs(row[k] * scale) : 0.0;
            // key = magnitude (22 bits) | 31 - position | lane: N <= 32, so position and lane take 5 bits each
            const unsigned int tkey = __float_as_uint((float)cand64) & ~0x3FFu;
            unsigned int key = tkey | ((unsigned int)(31 - (pos & 31)) << 5) | (unsigned int)(lane & 31);
            if (TEAM == 64) {
                key = __ockl_wfred_max_u32(key);
            } else {
#pragma unroll
                for (int m = TEAM / 2; m >= 1; m >>= 1) {
                    const unsigned int o = __shfl_xor(key, m, TEAM);
                    key = o > key ? o : key;
                }
            }
            const unsigned long long team_mask = TEAM == 64 ? ~0ull : (((1ull << TEAM) - 1ull) << team_base);
            const bool at_tmax = cand_ok && tkey == (key & ~0x3FFu);
            const unsigned long long tied = __ballot(at_tmax) & team_mask;
            int piv = (int)(key & 31u);
            if (TEAM < 32) piv &= (TEAM - 1);
            if (__any((tied & (tied - 1ull)) != 0ull ? 1 : 0)) {   // some team has several lanes at the truncated maximum
                const double w = TEAM == 64 ? __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(cand64) >> 32), piv) << 32) |
                                                                   (unsigned int)__builtin_amdgcn_readlane((int)(__double_as_longlong(cand64) & 0xFFFFFFFFLL), piv))
                                            : __shfl(cand64, piv, TEAM);
                if (__any((at_tmax && cand64 != w) ? 1 : 0))       // ... and they are not all exactly equal: exact comparison
                    piv = pivot_exact(cand64, cand_ok, pos, lane);
            }
            // singular test (math-code.c:393: scaled pivot <= 1e-20): decided by the truncated maximum unless that lies within a
            // factor of two of the threshold -- only then the winner's exact value is looked at
            const int src = (TEAM == 64) ? __builtin_amdgcn_readfirstlane(piv) : piv;
            const float best = __uint_as_float(key & ~0x3FFu);
            if (__any((best < 2.0e-20f && best > 0.5e-20f) ? 1 : 0)) {
                const unsigned long long big = __ballot(cand64 > 1.0e-20);
                if (on && ok && !((big >> (team_base + src)) & 1ull)) ok = false;
            } else if (on && ok && !(best > 1.0e-20f)) ok = false;
            const bool go = on && ok;
            // broadcast the pivot row (columns k..N-1 and the rhs)
            auto bcast = [&](double v) -> double {
                if (TEAM == 64) {
                    const long long b = __double_as_longlong(v);
                    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), src);
                    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
                    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                }
                return __shfl(v, src, TEAM);
            };
            const double pkk = bcast(row[k]);
            const double prhs = bcast(rhs);
            const bool is_piv = mine && (lane & (TEAM - 1)) == src;
            // bookkeeping of the reference's row order: it swaps the pivot row with the row at position k
            {
                const int pos_p = (TEAM == 64) ? __builtin_amdgcn_readlane(pos, src) : __shfl(pos, src, TEAM);
                if (__any(pos_p != k ? 1 : 0)) {     // (wave-uniform) almost never taken: the pivot usually is the row at position k
                    if (pos == k) pos = pos_p;
                    if (is_piv) pos = k;
                }
            }
            if (TRACE && go && is_piv) trace[k] = lane;
            // 1/pivot: hardware seed + two Newton steps (the multipliers need not be correctly rounded)
            double rp = tg_rcp(pkk);
            const double l = (go && mine && !is_piv) ? row[k] * rp : 0.0;
#pragma unroll
            for (int j = k + 1; j < N; j++) row[j] = fma(-l, bcast(row[j]), row[j]);
            rhs = fma(-l, prhs, rhs);
            if (go && is_piv) { mycol = k; diag = row[k]; }
            // Keep the elimination pivot-major: left alone, the instruction selector linearises the fully
            // unrolled body column by column (fma -> readlane of the same register -> fma ...): one long
            // dependent chain padded with hazard s_nops.  Passing the updated row through ordered empty asm
            // statements pins step k before step k+1; inside a step the broadcasts and fmas are independent.
#pragma unroll
            for (int j = k + 1; j < N; j += 8) {
                if (j + 7 < N) asm volatile("" : "+v"(row[j]), "+v"(row[j + 1]), "+v"(row[j + 2]), "+v"(row[j + 3]),
                                                 "+v"(row[j + 4]), "+v"(row[j + 5]), "+v"(row[j + 6]), "+v"(row[j + 7]));
                else {
#pragma unroll
                    for (int jj = j; jj < N; jj++) asm volatile("" : "+v"(row[jj]));
                }
            }
        }
        if (mine && ok && mycol >= 0 && mycol < n) A[mycol * ld + n] = rhs / diag;
        __syncthreads();
        return ok;
    }

    // ---- the default solver: same elimination, pivot candidates ranked in single precision -------------------------------
    //      One 32-bit wave max per step over (float bits of |a_ik| * scale_i with the 6 low mantissa bits replaced by
    //      63 - lane) and no branch anywhere in the unrolled body.  Candidates closer than 2^-17 relative are taken in lane
    //      (= original row) order.  That is NOT always the reference's choice: every row's largest entry scales to 1 +- 1 ulp,
    //      so whenever two rows have their largest entry in the same column (two string constraints and a shared torso
    //      config: 95 % of the puppet's Newton systems) the reference's strict `>` scan decides by that last ulp.  Either
    //      row is an exact arg-max to 16 digits and the solutions agree to rounding (1e-13 relative on the test matrices), but
    //      the pivot SEQUENCE can differ; gj_rows_exact() reproduces it exactly (RunArgs::exact_pivot, tg_batch_set_pivot_rule)
    //      at +9 % rollout time -- each variant of an in-line exact test (position bookkeeping +2.3 %, tie block +3.4 %, exact
    //      singular test +3.7 %; a branch-free "detect and redo" fires on 95 % of the solves) was measured and rejected.
#if defined(TG_GJ_INLINE)
#define TG_GJ_ATTR __forceinline__
#else
#define TG_GJ_ATTR __noinline__
#endif
    template <int N, bool TRACE = false>
    static __device__ TG_GJ_ATTR bool gj_rows(bool on, double *A_generic, int n, int ld, int lane, int *trace = nullptr) {
        typedef __attribute__((address_space(3))) double lds_double;
        lds_double *A = (lds_double *)A_generic;
        double row[N], rhs = 0.0, scale = 0.0, diag = 1.0;
        int mycol = -1;
        const bool mine = on && lane < N;
#pragma unroll
        for (int j = 0; j < N; j++)
            row[j] = (mine && lane < n && j < n) ? A[lane * ld + j] : ((mine && lane >= n && j == lane) ? 1.0 : 0.0);
        if (mine) {
            rhs = lane < n ? A[lane * ld + n] : 0.0;
            double s = -1.0;
#pragma unroll
            for (int j = 0; j < N; j++) { const double a = fabs(row[j]); s = a > s ? a : s; }
            scale = 1.0 / s;
        }
        bool ok = true;
#if defined(GJR_EXEC32)
        if (lane < 32)
#endif
#pragma unroll
        for (int k = 0; k < N; k++) {
            const float cand = (mine && mycol < 0) ? (float)fabs(row[k] * scale) : 0.0f;
            unsigned int key = (__float_as_uint(cand) & ~0x3Fu) | (unsigned int)(63 - (lane & 63));
            if (TEAM == 64) {
                key = tg_max_u32_lanes32(key);       // N <= 32: only lanes 0..31 hold rows (the others carry key 0 | lane bits)
            } else {
#pragma unroll
                for (int m = TEAM / 2; m >= 1; m >>= 1) {
                    const unsigned int o = __shfl_xor(key, m, TEAM);
                    key = o > key ? o : key;
                }
            }
            int piv = 63 - (int)(key & 0x3Fu);
            const float best = __uint_as_float(key & ~0x3Fu);
            if (TEAM != 64) piv = (piv & (TEAM - 1));
            if (on && ok && !(best > 1.0e-20f)) ok = false;
            const bool go = on && ok;
            // broadcast the pivot row (columns k..N-1 and the rhs)
            const int src = (TEAM == 64) ? __builtin_amdgcn_readfirstlane(piv) : piv;
            auto bcast = [&](double v) -> double {
                if (TEAM == 64) {
                    const long long b = __double_as_longlong(v);
                    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), src);
                    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
                    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                }
                return __shfl(v, src, TEAM);
            };
            const double prhs = bcast(rhs);
            const bool is_piv = mine && (lane & (TEAM - 1)) == src;
            if (TRACE && go && is_piv) trace[k] = lane;
            const double pkk = bcast(row[k]);
            double rp = tg_rcp(pkk);          // 1/pivot (the multipliers need not be correctly rounded)
            const double l = (go && mine && !is_piv) ? row[k] * rp : 0.0;
            // two pivot-row entries are broadcast before their two FMAs: a v_readlane result cannot be consumed by the next VALU
            // instruction (two wait states), so one broadcast-FMA pair at a time costs an s_nop per column (groups of four make
            // the unroller give up on the row registers: 420 instead of 66 ms)
#pragma unroll
            for (int j = k + 1; j < N; j += 2) {
                const double b0 = bcast(row[j]);
                const double b1 = bcast(row[j + 1 < N ? j + 1 : j]);
                row[j] = fma(-l, b0, row[j]);
                if (j + 1 < N) row[j + 1] = fma(-l, b1, row[j + 1]);
            }
            rhs = fma(-l, prhs, rhs);
            if (go && is_piv) { mycol = k; diag = row[k]; }
            // keep the elimination pivot-major (see gj_rows_exact)
#pragma unroll
            for (int j = k + 1; j < N; j += 8) {
                if (j + 7 < N) asm volatile("" : "+v"(row[j]), "+v"(row[j + 1]), "+v"(row[j + 2]), "+v"(row[j + 3]),
                                                 "+v"(row[j + 4]), "+v"(row[j + 5]), "+v"(row[j + 6]), "+v"(row[j + 7]));
                else {
#pragma unroll
                    for (int jj = j; jj < N; jj++) asm volatile("" : "+v"(row[jj]));
                }
            }
        }
        if (mine && ok && mycol >= 0 && mycol < n) A[mycol * ld + n] = rhs / diag;
        __syncthreads();
        return ok;
    }

    // ---- the full-wave solver (TEAM == 64, 16 < n < 32): panels of four columns + trailing update on the matrix cores ---------
    //      gj_rows keeps one matrix row per lane (28 of 64 lanes busy) and pays two v_readlane per FMA for the pivot-row
    //      broadcast: ~68 VALU instructions per pivot step, and the solver is bound by exactly that instruction stream.  Here the
    //      [A | b] matrix (b = column n) stays in LDS in its [n][ld] layout AND lives in the accumulator layout of
    //      v_mfma_f64_16x16x4_f64 on all 64 lanes: lane (g = lane >> 4, c = lane & 15), register v of tile (TR, TC) holds entry
    //      [16 TR + 4 v + g][16 TC + c] -- 16 doubles per lane instead of 29.  Per panel of four columns:
    //        1. lane i < n reads the four panel entries of row i from LDS (one row per lane);
    //        2. the four columns are eliminated exactly like gj_rows does it (scaled partial pivoting, rows never move, same
    //           single-precision ranking) -- but the broadcasts only cover the other panel columns and the columns of Z: 3 per step;
    //        3. Z [32][4] accumulates what the four elementary row operations do to any OTHER column: after the panel,
    //           A' = A + Z A[R, :] with R the four pivot rows as they were at the panel's start (block Gauss-Jordan; Z[:, t] is
    //           column r_t of the accumulated row-operation matrix minus the identity: z_t <- l at step t, z_s += l z_s[r_t] for s < t);
    //        4. Z goes through 1 KB of LDS into A-operand form, lane group t reads pivot row r_t straight from the LDS image as
    //           its B operand, the rank-4 update of the four (later two) 16 x 16 tiles is one v_mfma_f64_16x16x4_f64 each, and the
    //           live tiles are written back to the LDS image.
    //      No branch, no run-time register index.  Same pivot rule as gj_rows (the default rule); the trailing columns see the block
    //      update instead of four rank-1 updates, so results differ from gj_rows' by rounding only.  `scratch`: 128 doubles of LDS
    //      outside [A | b] (the Z table).
    //      Always an out-of-line function: inlined into the 20 k-instruction rollout kernel it shares that kernel's register
    //      allocation and schedule (91 instead of 53 SGPR spills, every other phase ~10 % slower: 63.3 ms per benchmark launch);
    //      as a call it keeps its own (61.7 ms; gj_rows: 65.1 ms).
    template <int N, bool TRACE = false>
    static __device__ __noinline__ bool gj_panel(bool on, double *A_generic, int n, int ld, int lane, double *scratch_generic, int *trace = nullptr) {
        static_assert(N % 4 == 0 && N > 16 && N <= 32, "gj_panel: 16 < N <= 32");
        typedef __attribute__((address_space(3))) double lds_double;
        typedef double v4d __attribute__((ext_vector_type(4)));
        lds_double *A = (lds_double *)A_generic, *WL = (lds_double *)scratch_generic;
        const int g = (lane >> 4) & 3, c = lane & 15;
        // TEAM == 64: the workgroup is ONE wavefront, whose LDS operations execute in program order -- a read issued after a write of
        // the same wave sees it, so no fence / s_waitcnt stands between the phases below; the compiler only has to keep may-alias
        // LDS accesses in program order, which it does (WL and A are both plain LDS pointers)
        auto lds_fence = [] { asm volatile("" ::: "memory"); };
        // rows 16 TR + 4 v + g of a register exist for every lane group / for none / for some (only when n is not a multiple of 4)
        auto rows_all = [&](int TR, int v) { return 16 * TR + 4 * v + 3 < n; };
        auto rows_none = [&](int TR, int v) { return 16 * TR + 4 * v >= n; };
        const bool in1 = 16 + c <= n;             // this lane's column of tile column 1 exists (b is column n)
        const int c1 = in1 ? 16 + c : 0;
        // ---- the matrix into the accumulator layout, the rows' scale factors one row per lane
        v4d T[2][2];
#pragma unroll
        for (int TR = 0; TR < 2; TR++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int row = 16 * TR + 4 * v + g;
                if (rows_none(TR, v)) { T[TR][0][v] = 0.0; T[TR][1][v] = 0.0; continue; }
                const bool rin = rows_all(TR, v) || row < n;
                const int ro = rin ? row * ld : 0;
                const double a0 = A[ro + c], a1 = A[ro + c1];
                T[TR][0][v] = rin ? a0 : 0.0;
                T[TR][1][v] = (rin && in1) ? a1 : 0.0;
            }
        const bool mine = lane < n;
        const int myrow = (mine ? lane : 0) * ld;     // lanes without a row mirror row 0: finite values that end up nowhere
        double scale = 0.0;
        {
            double s = -1.0;
#pragma unroll
            for (int j = 0; j < N; j++) if (j < n) { const double a = fabs(A[myrow + j]); s = a > s ? a : s; }
            scale = 1.0 / s;
        }
        bool ok = true;
        if (!mine) scale = 0.0;
        const unsigned int lanetag = (unsigned int)(63 - (lane & 63));
        int mycol = -1;
        double rdiag = 0.0;
#pragma unroll
        for (int p = 0; p < N / 4; p++) {
            // 1. the panel's entries of this lane's row
            double cp[4], z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 4; t++) cp[t] = A[myrow + (4 * p + t < n ? 4 * p + t : 0)];
            // 2. / 3. the four pivot steps
            int srcs[4] = {0, 0, 0, 0};
            double b0 = 0.0, b1 = 0.0;
            const bool live0 = 4 * p + 4 < 16;
#if !defined(GJP_SKIP_STEPS)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 4 * p + t;
                if (k < n) {
                    const double own_rp = tg_rcp(cp[t]);      // every lane inverts its own candidate while the search runs
                    // rows already used as pivots (and lanes without a row) carry scale 0: their key is the bare lane tag
                    const float cand = (float)(cp[t] * scale);
                    unsigned int key = (__float_as_uint(cand) & 0x7FFFFFC0u) | lanetag;
                    key = tg_max_u32_lanes32(key);
                    if (!((key & ~0x3Fu) > 0x1E3CE508u)) ok = false;     // scaled pivot <= 1e-20 (compared as bits: non-negative floats)
                    const int src = __builtin_amdgcn_readfirstlane(63 - (int)(key & 0x3Fu));
                    auto bcast = [&](double v) -> double {
                        const long long b = __double_as_longlong(v);
                        const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), src);
                        const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
                        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                    };
                    const bool is_piv = lane == src;
                    if (TRACE && on && ok && is_piv) trace[k] = lane;
                    srcs[t] = src;
                    if (t == 3 || k == n - 1) {
                        // lane group t fetches pivot row r_t of the image (as of the panel's start) as its B operand: requested
                        // here, as soon as the last pivot row is known, so that the loads travel under the last elimination step
                        const int prow = (g == 0 ? srcs[0] : (g == 1 ? srcs[1] : (g == 2 ? srcs[2] : srcs[3]))) * ld;
                        b1 = A[prow + c1];
                        if (live0) b0 = A[prow + c];
                    }
                    const double rp = bcast(own_rp);
                    // branch-free: a divergent if / else costs more (exec bookkeeping, two skipped-block branches) than three selects
                    const double l = is_piv ? 0.0 : cp[t] * -rp;
#pragma unroll
                    for (int t2 = t + 1; t2 < 4; t2++) cp[t2] = fma(l, bcast(cp[t2]), cp[t2]);
#pragma unroll
                    for (int s = 0; s < t; s++) z[s] = fma(l, bcast(z[s]), z[s]);
                    z[t] = l;
                    mycol = is_piv ? k : mycol;
                    rdiag = is_piv ? own_rp : rdiag;
                    scale = is_piv ? 0.0 : scale;
                }
            }
#endif
#if !defined(GJP_SKIP_UPDATE)
            // 4. Z -> A-operand form; pivot rows from the LDS image (as of the panel's start); trailing update; write back
            //    (tile column 0 is dead once the panel has passed column 11)
            if (lane < 32) {
#pragma unroll
                for (int t = 0; t < 4; t++) WL[lane * 4 + t] = z[t];
            }
            lds_fence();
            const double a0 = WL[c * 4 + g], a1 = WL[(16 + c) * 4 + g];
            if (!in1) b1 = 0.0;
            if (live0) {
                T[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, T[0][0], 0, 0, 0);
                T[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, T[1][0], 0, 0, 0);
            }
            T[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, T[0][1], 0, 0, 0);
            T[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, T[1][1], 0, 0, 0);
            lds_fence();      // every lane has its operands before the image changes
            auto write_back = [&](int TC) {
#pragma unroll
                for (int TR = 0; TR < 2; TR++)
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int row = 16 * TR + 4 * v + g;
                        if (!rows_none(TR, v) && (rows_all(TR, v) || row < n)) A[row * ld + 16 * TC + c] = T[TR][TC][v];
                    }
            };
            if (live0) write_back(0);
            if (in1) write_back(1);
            lds_fence();
#endif
        }
        // x = b / pivot, row by row: the right-hand side is column n of the image
        const double xr = A[myrow + n] * rdiag;
        lds_fence();
        if (on && mine && ok && mycol >= 0) A[mycol * ld + n] = xr;
        __syncthreads();
        return ok;
    }
    // ---- gj_panel for MANY right-hand sides (the derivative solves: n <= 31 rows, [A | B] with w <= 16 NTC columns) ------------
    //      The same panels, pivot rule and block update as gj_panel, with NTC tile columns in the accumulator layout (two tile
    //      rows x NTC tiles x 4 doubles per lane) instead of two: every right-hand side rides in the rank-4 matrix-core update
    //      (2 NTC v_mfma_f64_16x16x4 per panel for ~110 columns, where gj_cols spends 28 x 56 lane-wide FMAs plus the pivot-column
    //      traffic per PIVOT step).  The LDS image [n][ld] is refreshed after every panel (live tile columns only); at the end every
    //      lane scales its entries by the reciprocal pivot of their row and stores them in the row of the variable that row solved:
    //      A[i][n + j] = x_i of right-hand side j, like gauss_jordan() / gj_cols.  `scratch`: 128 + 64 doubles of LDS outside the image.
    template <int N, int NTC>
    static __device__ __noinline__ bool gj_panel_rhs(bool on, double *A_generic, int n, int w, int ld, int lane, double *scratch_generic) {
        static_assert(N % 4 == 0 && N > 16 && N <= 32 && NTC >= 2 && NTC <= 8, "gj_panel_rhs: 16 < N <= 32, 32 .. 128 columns");
        typedef __attribute__((address_space(3))) double lds_double;
        typedef double v4d __attribute__((ext_vector_type(4)));
        lds_double *A = (lds_double *)A_generic, *WL = (lds_double *)scratch_generic, *RD = WL + 128;
        __attribute__((address_space(3))) int *MC = (__attribute__((address_space(3))) int *)(WL + 160);
        const int g = (lane >> 4) & 3, c = lane & 15;
        auto lds_fence = [] { asm volatile("" ::: "memory"); };
        auto rows_all = [&](int TR, int v) { return 16 * TR + 4 * v + 3 < n; };
        auto rows_none = [&](int TR, int v) { return 16 * TR + 4 * v >= n; };
        // ---- [A | B] into the accumulator layout
        v4d T[2][NTC];
#pragma unroll
        for (int TR = 0; TR < 2; TR++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int row = 16 * TR + 4 * v + g;
                const bool rin = !rows_none(TR, v) && (rows_all(TR, v) || row < n);
#pragma unroll
                for (int TC = 0; TC < NTC; TC++) {
                    const int col = 16 * TC + c;
                    const bool in = rin && col < w;
                    const double a = A[in ? row * ld + col : 0];
                    T[TR][TC][v] = in ? a : 0.0;
                }
            }
        const bool mine = lane < n;
        const int myrow = (mine ? lane : 0) * ld;
        double scale = 0.0;
        {
            double s = -1.0;
#pragma unroll
            for (int j = 0; j < N; j++) if (j < n) { const double a = fabs(A[myrow + j]); s = a > s ? a : s; }
            scale = 1.0 / s;
        }
        bool ok = true;
        if (!mine) scale = 0.0;
        const unsigned int lanetag = (unsigned int)(63 - (lane & 63));
        int mycol = -1;
        double rdiag = 0.0;
#pragma unroll
        for (int p = 0; p < N / 4; p++) {
            double cp[4], z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 4; t++) cp[t] = A[myrow + (4 * p + t < n ? 4 * p + t : 0)];
            int srcs[4] = {0, 0, 0, 0};
            const int TC0 = (4 * p + 4) >> 4;        // first tile column with columns right of this panel
            double bop[NTC];
#pragma unroll
            for (int TC = 0; TC < NTC; TC++) bop[TC] = 0.0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 4 * p + t;
                if (k < n) {
                    const double own_rp = tg_rcp(cp[t]);
                    const float cand = (float)(cp[t] * scale);
                    unsigned int key = (__float_as_uint(cand) & 0x7FFFFFC0u) | lanetag;
                    key = tg_max_u32_lanes32(key);
                    if (!((key & ~0x3Fu) > 0x1E3CE508u)) ok = false;
                    const int src = __builtin_amdgcn_readfirstlane(63 - (int)(key & 0x3Fu));
                    auto bcast = [&](double v) -> double {
                        const long long b = __double_as_longlong(v);
                        const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), src);
                        const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
                        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                    };
                    const bool is_piv = lane == src;
                    srcs[t] = src;
                    if (t == 3 || k == n - 1) {   // B operands: lane group t takes pivot row r_t of the image as of the panel's start
                        const int prow = (g == 0 ? srcs[0] : (g == 1 ? srcs[1] : (g == 2 ? srcs[2] : srcs[3]))) * ld;
#pragma unroll
                        for (int TC = 0; TC < NTC; TC++) if (TC >= TC0) { const int col = 16 * TC + c; bop[TC] = A[prow + (col < w ? col : 0)]; }
                    }
                    const double rp = bcast(own_rp);
                    const double l = is_piv ? 0.0 : cp[t] * -rp;
#pragma unroll
                    for (int t2 = t + 1; t2 < 4; t2++) cp[t2] = fma(l, bcast(cp[t2]), cp[t2]);
#pragma unroll
                    for (int s = 0; s < t; s++) z[s] = fma(l, bcast(z[s]), z[s]);
                    z[t] = l;
                    mycol = is_piv ? k : mycol;
                    rdiag = is_piv ? own_rp : rdiag;
                    scale = is_piv ? 0.0 : scale;
                }
            }
            if (lane < 32) {
#pragma unroll
                for (int t = 0; t < 4; t++) WL[lane * 4 + t] = z[t];
            }
            lds_fence();
            const double a0 = WL[c * 4 + g], a1 = WL[(16 + c) * 4 + g];
#pragma unroll
            for (int TC = 0; TC < NTC; TC++) if (TC >= TC0) {
                const double b = 16 * TC + c < w ? bop[TC] : 0.0;
                T[0][TC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, T[0][TC], 0, 0, 0);
                T[1][TC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, T[1][TC], 0, 0, 0);
            }
            lds_fence();
            if (p + 1 < N / 4) {        // refresh the image (the last panel's result leaves through the scaled store below)
#pragma unroll
                for (int TC = 0; TC < NTC; TC++) if (TC >= TC0 && 16 * TC + c < w) {
#pragma unroll
                    for (int TR = 0; TR < 2; TR++)
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const int row = 16 * TR + 4 * v + g;
                            if (!rows_none(TR, v) && (rows_all(TR, v) || row < n)) A[row * ld + 16 * TC + c] = T[TR][TC][v];
                        }
                }
            }
            lds_fence();
        }
        // x = B / pivot, row by row, into the row of the variable each row solved
        if (mine) { RD[lane] = rdiag; MC[lane] = mycol; }
        lds_fence();
        if (on && ok) {
#pragma unroll
            for (int TR = 0; TR < 2; TR++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int row = 16 * TR + 4 * v + g;
                    if (rows_none(TR, v) || !(rows_all(TR, v) || row < n)) continue;
                    const double rd = RD[row];
                    const int mc = MC[row];
#pragma unroll
                    for (int TC = 0; TC < NTC; TC++) {
                        const int col = 16 * TC + c;
                        if (col >= n && col < w && mc >= 0) A[mc * ld + col] = T[TR][TC][v] * rd;
                    }
                }
        }
        TG_SYNC();
        return ok;
    }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
    // ---- Gauss-Jordan with two matrix COLUMNS PER LANE in registers: few rows, many right-hand sides --------------
    //      (the derivative solves: n <= NR <= 32 rows, up to 128 columns [A | rhs]).  Lane c holds columns c and
    //      c + 64 of every row.  Per pivot column k: lane k publishes its column in LDS (NR doubles); lane i < NR
    //      scales entry i and one 32-bit wave max picks the pivot row r; every lane reads the multipliers back with
    //      uniform (broadcast) LDS reads.  Rows never move and r is only known at run time, so a lane picks its
    //      pivot-row entries with a 0/1-weighted FMA sum over its rows instead of an indexed register read.  The
    //      matrix itself never touches LDS during the elimination.  Leaves x_i in A[i*ld + n + rhs] like
    //      gauss_jordan().  `scal` is 4*NR doubles of scratch (scale factors, pivot reciprocals, row -> variable
    //      map, current column).
    template <int NR>
    static __device__ TG_GJ_ATTR bool gj_cols(bool on, double *A_generic, int n, int w, int ld, double *scal_generic, int lane) {
        typedef __attribute__((address_space(3))) double lds_double;
        lds_double *A = (lds_double *)A_generic, *scal = (lds_double *)scal_generic;
        lds_double *dinv = scal + NR;                      // pivot reciprocal of physical row i
        double a0[NR], a1[NR];
        const bool c0 = on && lane < w, c1 = on && lane + 64 < w;
        // implicit scaling factors 1 / max_j |a_ij| over the matrix columns: lane i < n owns row i
        if (on && lane < NR) {
            double s = -1.0;
            if (lane < n) for (int j = 0; j < n; j++) { const double v = fabs(A[lane * ld + j]); s = v > s ? v : s; }
            scal[lane] = lane < n ? 1.0 / s : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#pragma unroll
        for (int i = 0; i < NR; i++) {
            a0[i] = (c0 && i < n) ? A[i * ld + lane] : 0.0;
            a1[i] = (c1 && i < n) ? A[i * ld + lane + 64] : 0.0;
        }
        bool ok = true;
        unsigned int used = 0u;
        lds_double *colbuf = scal + 3 * NR;                // column k of the current step, published by its lane
        for (int k = 0; k < n; k++) {
            if (lane == k) {
#pragma unroll
                for (int i = 0; i < NR; i++) colbuf[i] = a0[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            // scaled pivot search, one row per lane: the magnitude only ranks candidates, so single precision with the
            // lane in the low mantissa bits and one 32-bit wave max (scal is 0 for padding rows)
            float mf = 0.0f;
            if (lane < NR && !((used >> lane) & 1u)) mf = (float)(fabs(colbuf[lane]) * scal[lane]);
            const unsigned int key = __ockl_wfred_max_u32((__float_as_uint(mf) & ~0x3Fu) | (unsigned int)(63 - lane));
            const int r = __builtin_amdgcn_readfirstlane(63 - (int)(key & 0x3Fu));
            if (on && ok && !(__uint_as_float(key & ~0x3Fu) > 1.0e-20f)) ok = false;
            used |= 1u << r;
            // the pivot column in registers first (wave-uniform addresses: LDS broadcasts, all in flight together): a load inside
            // a `(i == r) ? .. : ..` arm turns into a scalar branch with its own s_waitcnt per row
            double cb[NR];
#pragma unroll
            for (int i = 0; i < NR; i++) cb[i] = colbuf[i];
            // this lane's pivot-row entries and the pivot: r is wave-uniform but not a compile-time register index, so a chain of
            // uniform branches picks them (a 0/1-weighted FMA sum is NR dependent fp64 FMAs at ~30 cycles each)
            double p0 = 0.0, p1 = 0.0, piv = 1.0;
#pragma unroll
            for (int i = 0; i < NR; i++) if (i == r) { p0 = a0[i]; p1 = a1[i]; piv = cb[i]; }
            double inv = tg_rcp(piv);
            if (lane == 0 && on) { ((__attribute__((address_space(3))) int *)(scal + 2 * NR))[r] = k; dinv[r] = 1.0 / piv; }
            const double ginv = (on && ok) ? inv : 0.0;
#pragma unroll
            for (int i = 0; i < NR; i++) {
                const double l = cb[i] * ((i == r) ? 0.0 : ginv);
                a0[i] = fma(-l, p0, a0[i]); a1[i] = fma(-l, p1, a1[i]);
            }
            // the next step overwrites colbuf: its reads above are ordered before those writes (same wavefront, in-order LDS)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        // physical row i solved variable var_i; write x = a / pivot into the row of the variable (right-hand sides only)
        if (on && ok) {
            const __attribute__((address_space(3))) int *var = (const __attribute__((address_space(3))) int *)(scal + 2 * NR);
#pragma unroll
            for (int i = 0; i < NR; i++) {
                if (i < n) {
                    const int v = var[i];
                    const double d = dinv[i];
                    if (c0 && lane >= n) A[v * ld + lane] = a0[i] * d;
                    if (c1) A[v * ld + lane + 64] = a1[i] * d;
                }
            }
        }
        TG_SYNC();
        return ok;
    }
#endif

    // =====================================================================================================
    // First derivatives of the step map (reference MidpointVI_calc_deriv1, midpointvi.c:749-1120).
    // The reference factors M2 and the projected matrix -Dh2 M2^-1 Dh1T separately and solves each
    // derivative variable in turn; here all nq+nd+nu+nk right-hand sides are appended to the SAME
    // Newton/KKT matrix [[M2, -Dh1T],[Dh2, 0]] (M2 = Df11 at the solution) and eliminated together.
    // =====================================================================================================

    // sum_c lambda_c h_c,dqdq(q_i, q_o) added to the q1 right-hand sides (calc_h1_deriv1 :864-889 and
    // the DDh1T term of calc_deriv1 :964-966); state q1 must be swept, pE valid.
    TG_HD void constraint_hessian_rhs(bool on, double *AUG, int ld) {
        const int nd = P.nd, nf = P.nf;
        // one lane per (constraint, a <= b) pair of the flat list (second derivatives are symmetric); several
        // constraints reach the same entry, hence LDS atomics (one wavefront: deterministic order; helper waves take the two
        // parts of the list that never meet at an entry)
        int p_lo, p_hi;
        wave_part(0, P.wc_split, P.n_cpair, p_lo, p_hi);
        const int *cp4 = nw > 1 ? P.wcp4 : P.cpair4;
        if (on) for (int pp = p_lo + tg_opaque(lane); pp < p_hi; pp += TEAM) {
            const int *pw = cp4 + 4 * (size_t)pp;
            const int c = pw[0], na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
            const double h = S[P.o_lam + c] * con_d2(c, na, nb);
            if (kb < nd) lds_add(&AUG[kb * ld + nf + ka], h);
            if (na != nb && ka < nd) lds_add(&AUG[ka * ld + nf + kb], h);
        }
        TG_WSYNC();
    }

    // builds and solves the augmented KKT system; `extra` appends nc unit columns e_{nd+c}.  Helper-wave kernels: called by every
    // wave of the trajectory; the sweeps, the constant blocks and the solve are wave 0's, the (constraint, a <= b) and (item, item)
    // pair loops are shared, the last helper wave clears the tables while wave 0 sweeps; every wave returns the solve's verdict.
    TG_HD bool deriv1_solve(bool on, bool extra) {
        const bool w0 = wave == 0;
        const int nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc, nf = P.nf;
        const bool cpt = d1_compact;      // compact slice: T12 in the pose union, Dh1 / Dh2 where T22 goes later, both tables cleared late
        const int ld = cpt ? P.a_aug_ld : P.d_aug_ld, R = P.d_nrhs;
        double *AUG = S + (cpt ? P.a_o_AUG : P.d_o_AUG), *T12 = S + (cpt ? P.a_o_T12 : P.d_o_T12), *T22 = S + (cpt ? P.a_o_T22 : P.d_o_T22);
        double *Dh1 = cpt ? T22 : S + P.d_o_Dh1, *Dh2 = cpt ? T22 + nc * nq : S + P.d_o_Dh2;
        const int c_q1 = nf, c_p1 = nf + nq, c_u1 = nf + nq + nd, c_k2 = nf + nq + nd + nu;
        if (on) {
            if (wave == nw - 1) {
                TG_FOR(i, nf * ld) AUG[i] = 0.0;
                if (!cpt) TG_FOR(i, nq * nd) { T12[i] = 0.0; T22[i] = 0.0; }
            }
            if (w0) TG_FOR(i, nc * nq) { Dh1[i] = 0.0; Dh2[i] = 0.0; }
        }
        TG_SYNC();
        // constraints at q1: Jacobian (held in the KKT matrix) and lambda-weighted Hessian
        if (nc) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_DUAL_SWEEP) && defined(TG_GJ_PANEL_DEFAULT)
            // (system-specialised kernels only: with the schedule interpreted at run time the fused sweep is 5 % slower here)
            if (dual_ok()) {      // the q1 and the q2 poses in one fused sweep (second set in the W area, dead until the midpoint evaluation)
                if (w0) {
                    dsA = 1; dsB = 2;
                    pose_sweep_dual(on, false);
                    dsA = 0;
                    attach_points(on, false, true);
                    constraints(on, 1, false, Dh1, nq);
                }
                TG_WSYNC();
                constraint_hessian_rhs(on, AUG, ld);
                if (w0) {
                    oGc = P.o_W;
                    attach_points(on, false, true);
                    constraints(on, 2, false, Dh2, nq);
                    oGc = P.o_G;
                }
            } else
#endif
            {
            if (w0) {
                pose_sweep(on, 1);
                attach_points(on, false, true);
                constraints(on, 1, false, Dh1, nq);
            }
            TG_WSYNC();
            constraint_hessian_rhs(on, AUG, ld);
            if (w0) {
                pose_sweep(on, 2);
                attach_points(on, false, true);
                constraints(on, 2, false, Dh2, nq);
            }
            }
        } else TG_WSYNC();   // (the cleared tables)
        if (w0) eval_midpoint(on);
        // constant blocks: forces (damping.c:21-27, configforce.c:27-33), -Dh1T, Dh2, unit p1 columns, k2 constraint rows
        if (on && w0) {
            TG_FOR(o, nd) {
                AUG[o * ld + o] -= P.damp[o];              // D2D1L2_D2fm2: + dF_o/d(dq_o)
                AUG[o * ld + c_q1 + o] -= P.damp[o];       // -(D1D1L2_D1fm2): -( - dF_o/d(dq_o) )
                AUG[o * ld + c_p1 + o] = -1.0;
                if (has_cs()) {   // a = dt/4 (-V_dqdq) on the diagonal of all four second-order tables
                    const double a_ = -0.25 * dt * cs_d2(o, qval(0, o));
                    AUG[o * ld + c_q1 + o] -= a_; AUG[o * ld + o] += a_;
                    T12[o * nd + o] += a_; T22[o * nd + o] += a_;
                }
                for (int k = 0; k < P.n_cf; k++) if (P.cf_cfg[k] == o) AUG[o * ld + c_u1 + P.cf_in[k]] -= dt;
                for (int c = 0; c < nc; c++) AUG[o * ld + nd + c] = -Dh1[c * nq + o];
            }
            TG_FOR(c, nc) {
                for (int o = 0; o < nd; o++) AUG[(nd + c) * ld + o] = Dh2[c * nq + o];
                for (int i = 0; i < nk; i++) AUG[(nd + c) * ld + c_k2 + i] = -Dh2[c * nq + nd + i];
                if (extra) AUG[(nd + c) * ld + nf + R + c] = 1.0;
            }
        }
        TG_WSYNC();
        if (cpt) {     // the poses and the full-width Jacobians are dead: their places become the two tables
            if (on) TG_FORW(i, nq * nd) { T12[i] = 0.0; T22[i] = 0.0; }
            TG_WSYNC();
        }
        // second-order discrete-Lagrangian tables from the (item,item) pairs (calc_deriv1_cache :749-861):
        //   a = dt/4 L_qq, b = L_dqdq/dt, c(r,o) = 1/2 L(dq_r, q_o);  D1D1 = a+b-c-cT, D2D1 = a-b+c-cT,
        //   D1D2 = a-b-c+cT, D2D2 = a+b+c+cT.
        const double qdt = 0.25 * dt, rdt = 1.0 / dt;
#if defined(__HIP_DEVICE_COMPILE__)
        // full-wave teams: ALL (item, item) pairs of all bodies in one flat pass (the per-body passes below fill 21 .. 55 of 64 lanes and
        // put a barrier between bodies); several bodies reach the same table entry, so the accumulation uses LDS atomics -- one
        // wavefront, whose LDS operations retire in order: the summation order is the same on every run (as in newton_matrix)
        const bool flat = TEAM == 64;
#else
        const bool flat = false;
#endif
        for (int b = 0; b < (flat ? 1 : P.n_bodies); b++) {
            const int p0 = flat ? 0 : P.b_pair_off[b], np = flat ? P.n_pairs : P.b_pair_off[b + 1] - p0;
            int p_lo = 0, p_hi = np;
            if (flat) wave_part(0, P.wp_split, np, p_lo, p_hi);
            const int *pa = flat && nw > 1 ? P.wp_a : P.pair_a + p0, *pb = flat && nw > 1 ? P.wp_b : P.pair_b + p0;
            if (on && (flat || w0)) for (int pp = p_lo + tg_opaque(lane); pp < p_hi; pp += TEAM) {
                const int ia = pa[pp], ib = pb[pp];
                const int ca = P.it_cfg[ia], cb = P.it_cfg[ib];
                if (ca >= nd && cb >= nd) continue;
                const int bb = flat ? P.it_body[ia] : b;
                const double *I = S + P.o_I + 4 * bb, *v = S + P.o_vB + 6 * bb, *gam = S + P.o_gam + 3 * bb;
                const double *Ja = S + P.o_J + 6 * ia, *Jb = S + P.o_J + 6 * ib;
                const double *Wa = S + P.o_W + 6 * ia, *Wb = S + P.o_W + 6 * ib;
                double tb[6];
                bracket(Wa, Jb, tb);
                const double lqq = inner6(I, tb, v) + inner6(I, Wa, Wb) +
                                   I[0] * (gam[0] * (Ja[4] * Jb[2] - Ja[5] * Jb[1]) + gam[1] * (Ja[5] * Jb[0] - Ja[3] * Jb[2]) +
                                           gam[2] * (Ja[3] * Jb[1] - Ja[4] * Jb[0]));
                const double a_ = qdt * lqq, b_ = rdt * inner6(I, Ja, Jb);
                double c_ab, c_ba;
                if (ia == ib) { c_ab = c_ba = 0.5 * inner6(I, Ja, Wa); }
                else {
                    bracket(Ja, Jb, tb);
                    c_ab = 0.5 * (inner6(I, tb, v) + inner6(I, Ja, Wb));
                    c_ba = 0.5 * inner6(I, Jb, Wa);
                }
                auto add = [&](int r, int o, double c_ro, double c_or) {
                    if (o >= nd) return;
                    const double d21 = a_ - b_ + c_ro - c_or;
                    if (flat) {
                        lds_add(&AUG[o * ld + c_q1 + r], -(a_ + b_ - c_ro - c_or));
                        if (r < nd) lds_add(&AUG[o * ld + r], d21);
                        else lds_add(&AUG[o * ld + c_k2 + (r - nd)], -d21);
                        lds_add(&T12[r * nd + o], a_ - b_ - c_ro + c_or);
                        lds_add(&T22[r * nd + o], a_ + b_ + c_ro + c_or);
                    } else {
                        AUG[o * ld + c_q1 + r] -= a_ + b_ - c_ro - c_or;
                        if (r < nd) AUG[o * ld + r] += d21;
                        else AUG[o * ld + c_k2 + (r - nd)] -= d21;
                        T12[r * nd + o] += a_ - b_ - c_ro + c_or;
                        T22[r * nd + o] += a_ + b_ + c_ro + c_or;
                    }
                };
                add(ca, cb, c_ab, c_ba);
                if (ia != ib) add(cb, ca, c_ba, c_ab);
            }
            TG_WSYNC();
        }
        bool ok = false;
        if (w0) ok = kkt_rest(on, extra);
        if (nw > 1) {     // (the matrix part of the image is dead after the solve)
            if (w0 && lane == 0) AUG[0] = ok ? 1.0 : 0.0;
            TG_WSYNC();
            ok = AUG[0] != 0.0;
        }
        return ok;
    }
    // remaining force terms and the solve (one wave)
    TG_HD bool kkt_rest(bool on, bool extra) {
        const int nq = P.nq, nd = P.nd, nu = P.nu, nc = P.nc, nf = P.nf;
        const bool cpt = d1_compact;
        const int ld = cpt ? P.a_aug_ld : P.d_aug_ld, R = P.d_nrhs;
        double *AUG = S + (cpt ? P.a_o_AUG : P.d_o_AUG), *T12 = S + (cpt ? P.a_o_T12 : P.d_o_T12), *T22 = S + (cpt ? P.a_o_T22 : P.d_o_T22);
        const int c_q1 = nf, c_u1 = nf + nq + nd, c_k2 = nf + nq + nd + nu;
        const double qdt = 0.25 * dt;
        if (n_wrenches()) {   // D1 fm2 = D2 fm2 = dt/2 F_dq into the q1 columns and M2 / the k2 columns; D3 fm2 = dt F_du
            if (on && lane == 0) {
                const int m0 = P.n_dh + n_sdh(), p0 = P.n_cpair + n_spair(), c0 = nc + n_springs();
                for (int pp = 0; pp < n_wpair(); pp++) {
                    const int *pw = P.cpair4 + 4 * (size_t)(p0 + pp);
                    const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                    auto add = [&](int r, int o, double a_) {   // a_ = dt/2 F_dq(o; r): force on config o, derivative variable r
                        if (o >= nd) return;
                        AUG[o * ld + c_q1 + r] -= a_;
                        if (r < nd) AUG[o * ld + r] += a_;
                        else AUG[o * ld + c_k2 + (r - nd)] -= a_;
                    };
                    add(kb, ka, 0.5 * dt * S[P.o_wH + 2 * pp]);
                    if (pw[1] != pw[2]) add(ka, kb, 0.5 * dt * S[P.o_wH + 2 * pp + 1]);
                }
                for (int n = 0; n < n_wdh(); n++) {
                    const int m = m0 + n, o = P.dh_cfg[m], w = P.dh_c[m] - c0;
                    if (o >= nd) continue;
                    for (int s6 = 0; s6 < 6; s6++) {
                        const int in = P.wr_in[6 * w + s6];
                        if (in >= 0) AUG[o * ld + c_u1 + in] -= dt * S[P.o_wD + 6 * n + s6];
                    }
                }
            }
            TG_SYNC();
        }
        if (n_springs()) {   // a = dt/4 (-V_dqdq) of the two-point springs enters all four tables like the gravity part of L_qq
            if (on && lane == 0) for (int pp = 0; pp < n_spair(); pp++) {
                const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
                const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                const double a_ = -qdt * S[P.o_sH + pp];
                auto add = [&](int r, int o) {
                    if (o >= nd) return;
                    AUG[o * ld + c_q1 + r] -= a_;
                    if (r < nd) AUG[o * ld + r] += a_;
                    else AUG[o * ld + c_k2 + (r - nd)] -= a_;
                    T12[r * nd + o] += a_;
                    T22[r * nd + o] += a_;
                };
                add(ka, kb);
                if (pw[1] != pw[2]) add(kb, ka);
                if (has_damper()) {   // D1 fm2 = dt/2 F_dq - F_ddq (q1 columns), D2 fm2 = dt/2 F_dq + F_ddq (M2 / k2 columns)
                    double fab, fba, fdd;
                    damper_pair(pp, fab, fba, fdd);
                    auto addf = [&](int r, int o, double fq) {   // force on config o, derivative variable r
                        if (o >= nd) return;
                        const double d1 = 0.5 * dt * fq - fdd, d2 = 0.5 * dt * fq + fdd;
                        AUG[o * ld + c_q1 + r] -= d1;
                        if (r < nd) AUG[o * ld + r] += d2;
                        else AUG[o * ld + c_k2 + (r - nd)] -= d2;
                    };
                    addf(kb, ka, fab);
                    if (pw[1] != pw[2]) addf(ka, kb, fba);
                }
            }
            TG_SYNC();
        }
        TG_STAMP(8);
#if defined(__HIP_DEVICE_COMPILE__) && defined(TG_GJ_PANEL_DEFAULT) && !defined(TG_NO_GJ_PANEL)
        // system-specialised kernels, full-wave team, 17..31 unknowns, at most 128 columns: panels of four columns with every
        // right-hand side in the rank-4 matrix-core update (sizes are compile-time constants of the specialisation header)
        {
            typedef typename std::remove_const<PROG>::type SP;
            constexpr int NF = SP::nf, W0 = SP::nf + SP::d_nrhs, W1 = W0 + SP::nc;
            if constexpr (TEAM == 64 && NF > 16 && NF <= 31 && W1 <= 128) {
                constexpr int NP = ((NF + 3) >> 2) << 2;
                double *sc = S + (cpt ? P.o_J : P.o_G);   // the joint poses (compact slice, where T12 sits there: the Jacobians) are dead during the solve
                if (extra) return Core<64, SPRINGS, PROG>::template gj_panel_rhs<NP, (W1 + 15) / 16>(on, AUG, NF, W1, ld, lane, sc);
                return Core<64, SPRINGS, PROG>::template gj_panel_rhs<NP, (W0 + 15) / 16>(on, AUG, NF, W0, ld, lane, sc);
            }
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        {   // small KKT matrix, many right-hand sides: register-resident column elimination (whole wavefront)
            const int w = nf + R + (extra ? nc : 0), nb4 = (nf + 3) >> 2;
            if (TEAM == 64 && w <= 128 && nb4 <= 8 && P.gjc_ok) {
                double *sc = S + (cpt ? P.o_J : P.o_G);   // the joint poses / the Jacobians are dead during the solve
                switch (nb4) {
                case 1: return Core<TEAM>::template gj_cols<4>(on, AUG, nf, w, ld, sc, lane);
                case 2: return Core<TEAM>::template gj_cols<8>(on, AUG, nf, w, ld, sc, lane);
                case 3: return Core<TEAM>::template gj_cols<12>(on, AUG, nf, w, ld, sc, lane);
                case 4: return Core<TEAM>::template gj_cols<16>(on, AUG, nf, w, ld, sc, lane);
                case 5: return Core<TEAM>::template gj_cols<20>(on, AUG, nf, w, ld, sc, lane);
                case 6: return Core<TEAM>::template gj_cols<24>(on, AUG, nf, w, ld, sc, lane);
                case 7: return Core<TEAM>::template gj_cols<28>(on, AUG, nf, w, ld, sc, lane);
                default: return Core<TEAM>::template gj_cols<32>(on, AUG, nf, w, ld, sc, lane);
                }
            }
        }
#endif
        return gauss_jordan(on, AUG, nf, R + (extra ? nc : 0), ld, S + P.o_scal);
    }

    TG_HD void deriv1(bool on, CArgs &A, size_t t) {
        const int nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc, nf = P.nf;
        const bool cpt = d1_compact;
        const int ld = cpt ? P.a_aug_ld : P.d_aug_ld, R = P.d_nrhs;
        double *AUG = S + (cpt ? P.a_o_AUG : P.d_o_AUG), *T12 = S + (cpt ? P.a_o_T12 : P.d_o_T12), *T22 = S + (cpt ? P.a_o_T22 : P.d_o_T22);
        const bool ok = deriv1_solve(on, false);
        const bool w0 = wave == 0;
        TG_STAMP(11);     // everything since the pair loop: the KKT solve
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        struct ProfDump { Core &c; CArgs &A; size_t t; int lane;
            __device__ ~ProfDump() { long long t_ = (long long)__builtin_amdgcn_s_memtime(); c.prof[13] += t_ - c.prof_last;
                          if (A.prof_out && t == 0 && lane == 0 && c.wave == 0) for (int i = 0; i < 16; i++) A.prof_out[i] = c.prof[i]; } } dump_{*this, A, t, lane};
#endif
        if (A.A_out) {
            // linearisation of the DSystem state map X_{k+1} = f(X_k, U_k), X = [Q; p; v], U = [u; rho]
            // (dsystem.py:284-317): rows Qd and p hold the transposed derivative blocks, rows Qk / v the
            // constant entries.  Lanes run along a row, so the global writes are contiguous.
            if (on) {
                const int nX = P.nX, nU = nu + nk, nqd = nq + nd;
                double *Ao = A.A_out + t * (size_t)nX * nX, *Bo = A.B_out + t * (size_t)nX * nU;
                bool rows_done = false;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_AB_MFMA)
                if (TEAM == 64 && nd <= 32) {
                    // The p2 rows are a dense product, P[o][v] = T1x[v][o] + sum_i T22[i][o] X[i][v] (nd x (nX + nU), K = nd): on the
                    // matrix cores, one v_mfma_f64_16x16x4 tile per 16 outputs x 16 variables, four tiles of a tile row side by side
                    // (independent accumulator chains).  A operand = T22' (rows o contiguous over the lanes), B operand = the solved
                    // right-hand sides in the KKT image, both straight from LDS; the accumulator layout has the variable index on the
                    // lanes, so the stores into A_k / B_k rows are 128-byte segments.  The VALU version below (one LDS read per FMA,
                    // 22 x 8 FMAs per lane and pass) was 29 % of this kernel.
                    typedef double v4d __attribute__((ext_vector_type(4)));
                    const int lr = lane & 15, lk = lane >> 4, nV = nX + nU, NTCv = (nV + 15) >> 4, KS = (nd + 3) >> 2;
                    auto svar = [&](int vv, int &kind, int &i, bool &vpart) {   // column vv of [A_k | B_k] -> index among (q1, p1, u1, k2)
                        vpart = vv >= nqd && vv < nX;
                        const int sv = vv < nqd ? vv : vv - nk;
                        kind = sv < nq ? 0 : (sv < nqd ? 1 : (sv < nqd + nu ? 2 : 3));
                        i = kind == 0 ? sv : (kind == 3 ? sv - nqd - nu : 0);
                        return sv;
                    };
                    for (int TR = wave; 16 * TR < nd; TR += nw) {     // (helper waves: a tile row each)
                        double av[8];
#pragma unroll
                        for (int ks = 0; ks < 8; ks++) {
                            const int i2 = 4 * ks + lk, o = 16 * TR + lr;
                            const bool in = ks < KS && i2 < nd && o < nd;
                            const double a = T22[in ? i2 * nd + o : 0];
                            av[ks] = in ? a : 0.0;
                        }
                        for (int TC0 = 0; TC0 < NTCv; TC0 += 4) {
                            v4d acc[4];
                            int bcol[4];
                            bool bin[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                acc[j] = (v4d){0.0, 0.0, 0.0, 0.0};
                                const int vv = 16 * (TC0 + j) + lr;
                                int kind, i; bool vpart;
                                const int sv = svar(vv < nV ? vv : 0, kind, i, vpart);
                                bin[j] = ok && TC0 + j < NTCv && vv < nV && !vpart;
                                bcol[j] = nf + sv;
                            }
#pragma unroll
                            for (int ks = 0; ks < 8; ks++) if (ks < KS) {
                                const int i2 = 4 * ks + lk;
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const bool in = bin[j] && i2 < nd;
                                    const double b = AUG[in ? i2 * ld + bcol[j] : 0];
                                    acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], in ? b : 0.0, acc[j], 0, 0, 0);
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int vv = 16 * (TC0 + j) + lr;
                                if (TC0 + j >= NTCv || vv >= nV) continue;
                                int kind, i; bool vpart;
                                const int sv = svar(vv, kind, i, vpart);
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const int o = 16 * TR + lk + 4 * r;
                                    if (o >= nd) continue;
                                    double x = NAN, pv = NAN;
                                    if (vpart) { x = 0.0; pv = 0.0; }
                                    else if (ok) {
                                        x = AUG[o * ld + nf + sv];
                                        pv = acc[j][r] + (kind == 0 ? T12[i * nd + o] : (kind == 3 ? T22[(nd + i) * nd + o] : 0.0));
                                    }
                                    if (vv < nX) { Ao[(size_t)o * nX + vv] = x; Ao[(size_t)(nq + o) * nX + vv] = pv; }
                                    else { Bo[(size_t)o * nU + vv - nX] = x; Bo[(size_t)(nq + o) * nU + vv - nX] = pv; }
                                }
                            }
                        }
                    }
                    rows_done = true;
                }
#endif
                // p2 derivative = T12 / T22 row + T22' x: OB output rows per pass share the loads of x and run as OB independent
                // accumulation chains (a single chain is nd dependent fp64 FMAs at ~30 cycles each)
                constexpr int OB = 8;
                if (!rows_done && w0) for (int o0 = 0; o0 < nd; o0 += OB) TG_FOR(vv, nX + nU) {
                    double x[OB], p[OB];
#pragma unroll
                    for (int j = 0; j < OB; j++) { x[j] = NAN; p[j] = NAN; }
                    if (vv >= nqd && vv < nX) {                           // columns of the v part of X
#pragma unroll
                        for (int j = 0; j < OB; j++) { x[j] = 0.0; p[j] = 0.0; }
                    } else if (ok) {
                        const int sv = vv < nqd ? vv : vv - nk;           // index among (q1, p1, u1, k2)
                        const int kind = sv < nq ? 0 : (sv < nqd ? 1 : (sv < nqd + nu ? 2 : 3));
                        const int i = kind == 0 ? sv : (kind == 3 ? sv - nqd - nu : 0);
#pragma unroll
                        for (int j = 0; j < OB; j++) {
                            const int o = o0 + j < nd ? o0 + j : nd - 1;
                            x[j] = AUG[o * ld + nf + sv];
                            p[j] = kind == 0 ? T12[i * nd + o] : (kind == 3 ? T22[(nd + i) * nd + o] : 0.0);
                        }
#pragma unroll 2
                        for (int i2 = 0; i2 < nd; i2++) {
                            const double a = AUG[i2 * ld + nf + sv];
                            const double *tr = T22 + i2 * nd;
#pragma unroll
                            for (int j = 0; j < OB; j++) p[j] = fma(tr[o0 + j < nd ? o0 + j : nd - 1], a, p[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < OB; j++) {
                        const int o = o0 + j;
                        if (o < nd) {
                            if (vv < nX) { Ao[(size_t)o * nX + vv] = x[j]; Ao[(size_t)(nq + o) * nX + vv] = p[j]; }
                            else { Bo[(size_t)o * nU + vv - nX] = x[j]; Bo[(size_t)(nq + o) * nU + vv - nX] = p[j]; }
                        }
                    }
                }
                const double rdt = 1.0 / dt;
                for (int i = wave; i < nk; i += nw) TG_FOR(vv, nX + nU) {        // Qk_{k+1} = rho_k, v_{k+1} = (rho_k - Qk_k)/dt
                    if (vv < nX) { Ao[(size_t)(nd + i) * nX + vv] = 0.0; Ao[(size_t)(nqd + i) * nX + vv] = vv == nd + i ? -rdt : 0.0; }
                    else {
                        const bool hit = vv - nX == nu + i;
                        Bo[(size_t)(nd + i) * nU + vv - nX] = hit ? 1.0 : 0.0;
                        Bo[(size_t)(nqd + i) * nU + vv - nX] = hit ? rdt : 0.0;
                    }
                }
            }
            return;
        }
        // outputs in the reference layout [derivative variable][output] (trep.h:425-437)
        if (on && w0) {
            const int cwl = tile_log2<TEAM>(nd), cw = 1 << cwl, rstep = TEAM >> cwl;
            for (int vv = lane >> cwl; vv < R; vv += rstep) {
                int kind, i;
                if (vv < nq) { kind = 0; i = vv; }
                else if (vv < nq + nd) { kind = 1; i = vv - nq; }
                else if (vv < nq + nd + nu) { kind = 2; i = vv - nq - nd; }
                else { kind = 3; i = vv - nq - nd - nu; }
                const int rows = kind == 0 ? nq : (kind == 1 ? nd : (kind == 2 ? nu : nk));
                for (int o = lane & (cw - 1); o < nd; o += cw) {
                    const double x = ok ? AUG[o * ld + nf + vv] : NAN;
                    double p = kind == 0 ? T12[i * nd + o] : (kind == 3 ? T22[(nd + i) * nd + o] : 0.0);
#pragma unroll 4
                    for (int i2 = 0; i2 < nd; i2++) p += T22[i2 * nd + o] * AUG[i2 * ld + nf + vv];
                    A.d1[kind][(t * rows + i) * nd + o] = x;
                    A.d1[4 + kind][(t * rows + i) * nd + o] = ok ? p : NAN;
                }
                for (int c = lane & (cw - 1); c < nc; c += cw)
                    A.d1[8 + kind][(t * rows + i) * nc + c] = ok ? AUG[(nd + c) * ld + nf + vv] : NAN;
            }
        }
    }

    // =====================================================================================================
    // Second derivatives, contracted with z over the output index (what DSystem.fdxdx/fdxdu/fdudu consume,
    // dsystem.py:320-386): HZ[a][b] = sum_o z_Qd[o] q2''[a][b][o] + z_p[o] p2''[a][b][o] for all pairs of the
    // derivative variables (q1, p1, u1, k2).  The reference materialises six nq x nq x nd tables and ten
    // [A][B][out] tensors (midpointvi.c:1122-2545); here the contraction is pushed through the
    // implicit-function solve with one adjoint vector w = K^-T [z_Qd + D2D2L2 z_p ; 0], so only three
    // nq x nq contracted Hessians are ever formed:
    //   H11 = sum_o -w_o (D1D1D1L2 - sum_c lambda_c DDDh1T)[.,.,o] + z_p,o D1D1D2L2[.,.,o]
    //   H12 = sum_o -w_o D1D2D1L2[.,.,o] + z_p,o D1D2D2L2[.,.,o]
    //   H22 = sum_o -w_o D2D2D1L2[.,.,o] + z_p,o D2D2D2L2[.,.,o] - sum_c w_lambda,c DDh2[c]
    // and HZ[a][b] = H11[a1][b1] + (H12 y_b + G1 l_b)[a1] + (H12 y_a + G1 l_a)[b1] + y_a^T H22 y_b with the
    // first-derivative tangents y (slot-2 configs) and l (multipliers); G1[i][c] = sum_o w_o DDh1T[i][o][c].
    // Third-order Lagrangian derivatives come from nested brackets: d3v = [[W_i,J_j],J_k], d2J_i = [[J_i,J_j],J_k].
    // =====================================================================================================
    TG_HD void pos_d2(int e, int ja, int jb, Real *out) const {  // d2 p_E / dq_a dq_b, ja nearer the root
        out[0] = out[1] = out[2] = 0.0;
        if (P.j_kind[ja] < TG_RX) return;
        const Real *ga = S + P.o_G + 12 * ja;
        const int ax = P.j_kind[ja] - TG_RX;
        const Real wx = ga[ax], wy = ga[4 + ax], wz = ga[8 + ax];
        Real d[3];
        dpos(e, jb, d);
        out[0] = wy * d[2] - wz * d[1]; out[1] = wz * d[0] - wx * d[2]; out[2] = wx * d[1] - wy * d[0];
    }
    TG_HD void pos_d3(int e, int ja, int jb, int jc, Real *out) const {  // ja <= jb <= jc along the path
        out[0] = out[1] = out[2] = 0.0;
        if (P.j_kind[ja] < TG_RX) return;
        Real d[3];
        pos_d2(e, jb, jc, d);
        const Real *ga = S + P.o_G + 12 * ja;
        const int ax = P.j_kind[ja] - TG_RX;
        const Real wx = ga[ax], wy = ga[4 + ax], wz = ga[8 + ax];
        out[0] = wy * d[2] - wz * d[1]; out[1] = wz * d[0] - wx * d[2]; out[2] = wx * d[1] - wy * d[0];
    }
    // difference (end point 1 - end point 2) of the 1st/2nd/3rd position derivative of constraint c
    TG_HD void cdiff1(int c, int n, Real *v) const {
        Real a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        const int s = P.dh_side[n], j = P.dh_joint[n];
        if (j >= 0 && (s & 1)) dpos(P.c_e1[c], j, a);
        if (j >= 0 && (s & 2)) dpos(P.c_e2[c], j, b);
        v[0] = a[0] - b[0]; v[1] = a[1] - b[1]; v[2] = a[2] - b[2];
    }
    TG_HD void cdiff2(int c, int n1, int n2, Real *v) const {
        Real a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        int j1 = P.dh_joint[n1], j2 = P.dh_joint[n2];
        const int s = P.dh_side[n1] & P.dh_side[n2];
        if (j1 >= 0 && j2 >= 0) {
            if (j1 > j2) { const int t_ = j1; j1 = j2; j2 = t_; }
            if (s & 1) pos_d2(P.c_e1[c], j1, j2, a);
            if (s & 2) pos_d2(P.c_e2[c], j1, j2, b);
        }
        v[0] = a[0] - b[0]; v[1] = a[1] - b[1]; v[2] = a[2] - b[2];
    }
    TG_HD void cdiff3(int c, int n1, int n2, int n3, Real *v) const {
        Real a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        int j1 = P.dh_joint[n1], j2 = P.dh_joint[n2], j3 = P.dh_joint[n3];
        const int s = P.dh_side[n1] & P.dh_side[n2] & P.dh_side[n3];
        if (j1 >= 0 && j2 >= 0 && j3 >= 0) {
            if (j1 > j2) { const int t_ = j1; j1 = j2; j2 = t_; }
            if (j2 > j3) { const int t_ = j2; j2 = j3; j3 = t_; }
            if (j1 > j2) { const int t_ = j1; j1 = j2; j2 = t_; }
            if (s & 1) pos_d3(P.c_e1[c], j1, j2, j3, a);
            if (s & 2) pos_d3(P.c_e2[c], j1, j2, j3, b);
        }
        v[0] = a[0] - b[0]; v[1] = a[1] - b[1]; v[2] = a[2] - b[2];
    }
    // Two-point springs V = 1/2 k (|p1 - p2| - x0)^2 at the swept state (linearspring.c:30-78): gradient per config
    // into sV, Hessian per (item, item) pair into sH.  Both live outside the storage the Newton matrix shares with the
    // poses, because the matrix is assembled after the poses are gone.  Poses and end points must be valid.
    TG_HD void spring_terms(bool on) {
        if (n_springs() == 0) return;
        Real *sV = S + P.o_sV, *sH = S + P.o_sH;
        if (on) TG_FOR(i, P.nq) sV[i] = 0.0;
        TG_SYNC();
        if (on) {
            TG_FOR(n, n_sdh()) {
                const int m = P.n_dh + n, c = P.dh_c[m], k = P.dh_cfg[m], sp = c - P.nc;
                const Real *a = S + P.o_pE + 3 * P.c_e1[c], *b = S + P.o_pE + 3 * P.c_e2[c];
                const Real v[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
                const Real x = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                Real v1[3];
                cdiff1(c, m, v1);
                const Real dx = (1.0 / x) * (v[0] * v1[0] + v[1] * v1[1] + v[2] * v1[2]);
                if (has_damper()) S[P.o_sX + n] = dx == dx ? dx : 0.0;
                if (dx != dx && P.s_x0[sp] == 0.0) continue;   // coincident end points of a zero-length spring (:44-45)
                lds_add(&sV[k], P.s_k[sp] * (x - P.s_x0[sp]) * dx);
            }
            TG_FOR(pp, n_spair()) {
                const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
                const int c = pw[0], na = pw[1], nb = pw[2], sp = c - P.nc;
                const Real *a = S + P.o_pE + 3 * P.c_e1[c], *b = S + P.o_pE + 3 * P.c_e2[c];
                const Real v[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
                const Real x = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                Real vi[3], vj[3], vij[3];
                cdiff1(c, na, vi); cdiff1(c, nb, vj); cdiff2(c, na, nb, vij);
                const Real vvi = v[0] * vi[0] + v[1] * vi[1] + v[2] * vi[2];
                const Real dix = (1.0 / x) * vvi;
                const Real djx = (1.0 / x) * (v[0] * vj[0] + v[1] * vj[1] + v[2] * vj[2]);
                const Real ddx = -djx / (x * x) * vvi + 1.0 / x * (vj[0] * vi[0] + vj[1] * vi[1] + vj[2] * vi[2]) +
                                   1.0 / x * (v[0] * vij[0] + v[1] * vij[1] + v[2] * vij[2]);
                sH[pp] = P.s_k[sp] * dix * djx + P.s_k[sp] * (x - P.s_x0[sp]) * ddx;
                if (has_damper()) S[P.o_sXX + pp] = ddx;
            }
        }
        TG_SYNC();
        if (has_damper()) {   // lineardamper.c:12-45: rate of every element, d(rate)/dq per item, generalized force
            const Real *dqv = S + P.o_dq;
            if (on) {
                TG_FOR(i, n_springs()) S[P.o_svel + i] = 0.0;
                TG_FOR(i, n_sdh()) S[P.o_sVq + i] = 0.0;
                TG_FOR(i, P.nd) S[P.o_sF + i] = 0.0;
            }
            TG_SYNC();
            if (on) {
                TG_FOR(n, n_sdh()) {
                    const int m = P.n_dh + n;
                    lds_add(&S[P.o_svel + (P.dh_c[m] - P.nc)], S[P.o_sX + n] * dqv[P.dh_cfg[m]]);
                }
                TG_FOR(pp, n_spair()) {
                    const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
                    const int na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                    const Real xab = S[P.o_sXX + pp];
                    lds_add(&S[P.o_sVq + (na - P.n_dh)], xab * dqv[kb]);
                    if (na != nb) lds_add(&S[P.o_sVq + (nb - P.n_dh)], xab * dqv[ka]);
                }
            }
            TG_SYNC();
            if (on) TG_FOR(n, n_sdh()) {
                const int m = P.n_dh + n, sp = P.dh_c[m] - P.nc, k = P.dh_cfg[m];
                if (k < P.nd) lds_add(&S[P.o_sF + k], -P.s_c[sp] * S[P.o_svel + sp] * S[P.o_sX + n]);
            }
            TG_SYNC();
            if (d2w) damper_second(on);
        }
    }

    // d3|p1 - p2| / dq_a dq_b dq_c of spring element c for three of its dh items (tapemeasure.c:98-168); poses alive
    TG_HD Real length_d3(int c, int na, int nb, int nc_) const {
        const Real *pa = S + P.o_pE + 3 * P.c_e1[c], *pb = S + P.o_pE + 3 * P.c_e2[c];
        const Real v[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
        Real a[3], b[3], cc[3], ab[3], ac[3], bc[3], abc[3];
        cdiff1(c, na, a); cdiff1(c, nb, b); cdiff1(c, nc_, cc);
        cdiff2(c, na, nb, ab); cdiff2(c, na, nc_, ac); cdiff2(c, nb, nc_, bc);
        cdiff3(c, na, nb, nc_, abc);
        const Real x = sqrt(dot3(v, v));
        const Real xa = dot3(v, a) / x, xb = dot3(v, b) / x, xc = dot3(v, cc) / x;
        const Real xab = (dot3(a, b) + dot3(v, ab) - xa * xb) / x, xac = (dot3(a, cc) + dot3(v, ac) - xa * xc) / x;
        const Real xbc = (dot3(b, cc) + dot3(v, bc) - xb * xc) / x;
        return (dot3(ac, b) + dot3(a, bc) + dot3(cc, ab) + dot3(v, abc) - xac * xb - xa * xbc - xab * xc) / x;
    }
    // Second-derivative kernel, dampers (lineardamper.c:47-92 contracted with the adjoint weights w): per element pair
    //   P = sum_o w_o f_dqdq(o; a, b),  R = sum_o w_o f_ddqdq(o; dq a, q b)
    // (R is symmetric in (a, b) because the reference's f_ddqdq uses length_dq(q2) where length_dqdq(q, q2) is meant,
    // :88 -- reproduced).  Needs the poses and the first-order damper quantities.
    TG_HD void damper_second(bool on) {
        Real *sT = S + P.e_o_sT, *WX = S + P.e_o_sWX, *WXq = S + P.e_o_sWXq;
        const Real *dqv = S + P.o_dq;
        if (on) {
            TG_FOR(i, 2 * n_springs()) WX[i] = 0.0;
            TG_FOR(i, n_sdh()) WXq[i] = 0.0;
        }
        TG_SYNC();
        if (on) {
            TG_FOR(n, n_sdh()) {
                const int m = P.n_dh + n, sp = P.dh_c[m] - P.nc, k = P.dh_cfg[m];
                if (k < P.nd) { lds_add(&WX[2 * sp], d2w[k] * S[P.o_sX + n]); lds_add(&WX[2 * sp + 1], d2w[k]); }
            }
            TG_FOR(pp, n_spair()) {
                const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
                const int na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                const Real xab = S[P.o_sXX + pp];
                if (kb < P.nd) lds_add(&WXq[na - P.n_dh], d2w[kb] * xab);
                if (na != nb && ka < P.nd) lds_add(&WXq[nb - P.n_dh], d2w[ka] * xab);
            }
        }
        TG_SYNC();
        if (on) TG_FOR(pp, n_spair()) {
            const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
            const int c = pw[0], sp = c - P.nc, na = pw[1], nb = pw[2], ia = na - P.n_dh, ib = nb - P.n_dh;
            const Real cc = P.s_c[sp], vel = S[P.o_svel + sp], xa = S[P.o_sX + ia], xb = S[P.o_sX + ib], xab = S[P.o_sXX + pp];
            Real vab = 0.0, wxab = 0.0;
            for (int no = P.cu_off[c]; no < P.cu_off[c + 1]; no++) {
                const int ko = P.dh_cfg[no];
                const Real x3 = length_d3(c, no, na, nb);
                vab += x3 * dqv[ko];
                if (ko < P.nd) wxab += d2w[ko] * x3;
            }
            sT[2 * pp] = -cc * (vab * WX[2 * sp] + S[P.o_sVq + ia] * WXq[ib] + S[P.o_sVq + ib] * WXq[ia] + vel * wxab);
            sT[2 * pp + 1] = -cc * (xab * WX[2 * sp] + xa * xb * WX[2 * sp + 1]);
        }
        TG_SYNC();
    }

    // derivatives of the damper force of the element pair pp = (na, nb): F_dq(a; b), F_dq(b; a) and F_ddq (symmetric)
    TG_HD void damper_pair(int pp, Real &fq_ab, Real &fq_ba, Real &fdd) const {
        const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
        const int sp = pw[0] - P.nc, ia = pw[1] - P.n_dh, ib = pw[2] - P.n_dh;
        const Real c = P.s_c[sp], vel = S[P.o_svel + sp], xa = S[P.o_sX + ia], xb = S[P.o_sX + ib], xab = S[P.o_sXX + pp];
        fq_ab = -c * (S[P.o_sVq + ib] * xa + vel * xab);
        fq_ba = -c * (S[P.o_sVq + ia] * xb + vel * xab);
        fdd = -c * xa * xb;
    }

    // Wrenches (hybridwrench.c:17-205) at the swept state.  With dp_n = d p/dq_n and om_n = the world axis of joint n (zero
    // for a prismatic one) the generalized force is F . dp_n + tau . om_n (into wF per dynamic config); wD keeps (dp_n, om_n)
    // per item (the input columns of the derivatives); per item pair, F_dq is F . d2p (symmetric) plus tau . (om_b x om_a)
    // when joint b comes before joint a on the path -- not symmetric, so wH holds F_dq(a; b) and F_dq(b; a).
    // the six coefficients of wrench element c for its dh item n: (dp/dq, axis) for a HybridWrench, the joint's spatial
    // twist (dp/dq - axis x p, axis) for a SpatialWrench (spatialwrench.c:16-38: unhat(g_dq g^-1))
    TG_HD void wrench_coeff(int c, int n, int kind, Real *xi) const {
        cdiff1(c, n, xi);
        plane_axis(n, xi + 3);
        if (kind == 1) {
            const Real *p = S + P.o_pE + 3 * P.c_e1[c];
            Real t_[3];
            cross3(xi + 3, p, t_);
            xi[0] -= t_[0]; xi[1] -= t_[1]; xi[2] -= t_[2];
        } else if (kind == 2) {   // BodyWrench (bodywrench.c:16-38, unhat(g^-1 g_dq)): the same two vectors in the frame's axes
            const Real *R = S + P.o_wR + 9 * (c - P.nc - n_springs());
            const Real a[6] = {xi[0], xi[1], xi[2], xi[3], xi[4], xi[5]};
            for (int i = 0; i < 3; i++) {
                xi[i] = R[i] * a[0] + R[3 + i] * a[1] + R[6 + i] * a[2];
                xi[3 + i] = R[i] * a[3] + R[3 + i] * a[4] + R[6 + i] * a[5];
            }
        }
    }
    TG_HD void wrench_terms(bool on) {
        if (n_wrenches() == 0) return;
        Real *wF = S + P.o_wF, *wH = S + P.o_wH, *wD = S + P.o_wD;
        const int m0 = P.n_dh + n_sdh(), p0 = P.n_cpair + n_spair(), c0 = P.nc + n_springs();
        if (on) TG_FOR(i, P.nd) wF[i] = 0.0;
        TG_SYNC();
        auto component = [&](int w, int s6) { const int in = P.wr_in[6 * w + s6]; return in >= 0 ? S[P.o_u + in] : P.wr_const[6 * w + s6]; };
        if (on) {
            TG_FOR(n, n_wdh()) {
                const int m = m0 + n, c = P.dh_c[m], k = P.dh_cfg[m], w = c - c0;
                Real xi[6], f = 0.0;
                wrench_coeff(c, m, P.wr_kind[w], xi);
                for (int r = 0; r < 6; r++) { wD[6 * n + r] = xi[r]; f += component(w, r) * xi[r]; }
                if (k < P.nd) lds_add(&wF[k], f);
            }
            TG_FOR(pp, n_wpair()) {
                const int *pw = P.cpair4 + 4 * (size_t)(p0 + pp);
                const int c = pw[0], w = c - c0, na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                const int ja = P.dh_joint[na], jb = P.dh_joint[nb];
                if (P.wr_kind[w] != 0) {
                    // SpatialWrench: d xi_a/dq_b = [xi_b, xi_a] for b before a;  BodyWrench: d xi_a/dq_b = [xi_a, xi_b] for b
                    // after a.  Nothing symmetric.  With br = [xi_a, xi_b]:
                    const int kind = P.wr_kind[w];
                    const bool body = kind == 2;
                    Real Wv[6], xa[6], xb[6], br[6];
                    for (int r = 0; r < 6; r++) Wv[r] = component(w, r);
                    wrench_coeff(c, na, kind, xa); wrench_coeff(c, nb, kind, xb);
                    bracket(xa, xb, br);
                    Real t_ab = 0.0;
                    for (int r = 0; r < 6; r++) t_ab += Wv[r] * br[r];
                    const bool d_ab = body ? ja < jb : jb < ja;   // does coefficient a depend on joint b?
                    const bool d_ba = body ? jb < ja : ja < jb;   // does coefficient b depend on joint a?
                    const Real sg = body ? 1.0 : -1.0;          // d xi_a/dq_b = sg br,  d xi_b/dq_a = -sg br
                    wH[2 * pp] = d_ab ? sg * t_ab : 0.0;
                    wH[2 * pp + 1] = d_ba ? -sg * t_ab : 0.0;
                    if (d2w) {
                        Real acc = 0.0;
                        const int n1 = ja <= jb ? na : nb, n2 = ja <= jb ? nb : na, j1 = ja <= jb ? ja : jb, j2 = ja <= jb ? jb : ja;
                        Real x1[6], x2[6];
                        wrench_coeff(c, n1, kind, x1); wrench_coeff(c, n2, kind, x2);
                        for (int no = P.cu_off[c]; no < P.cu_off[c + 1]; no++) {
                            const int ko = P.dh_cfg[no], jo = P.dh_joint[no];
                            if (ko >= P.nd || !(body ? jo < j1 : j2 < jo)) continue;
                            Real xo[6], u1[6], u2[6], term = 0.0;
                            wrench_coeff(c, no, kind, xo);
                            if (body) { bracket(xo, x1, u1); bracket(u1, x2, u2); }     // [[xi_o, xi_1], xi_2]
                            else { bracket(x2, xo, u1); bracket(x1, u1, u2); }          // [xi_1, [xi_2, xi_o]]
                            for (int r = 0; r < 6; r++) term += Wv[r] * u2[r];
                            acc += d2w[ko] * term;
                        }
                        S[P.e_o_wT + pp] = acc;
                        Real *Hu = S + P.e_o_Hu;
                        for (int s6 = 0; s6 < 6; s6++) {
                            const int in = P.wr_in[6 * w + s6];
                            if (in < 0) continue;
                            // F_dudq(o = b, u; a) = d xi_b/dq_a and F_dudq(o = a, u; b) = d xi_a/dq_b, component s6
                            if (kb < P.nd && d_ba) lds_add(&Hu[ka * P.nu + in], -0.5 * dt * d2w[kb] * (-sg * br[s6]));
                            if (na != nb && ka < P.nd && d_ab) lds_add(&Hu[kb * P.nu + in], -0.5 * dt * d2w[ka] * (sg * br[s6]));
                        }
                    }
                    continue;
                }
                const Real F[3] = {component(w, 0), component(w, 1), component(w, 2)};
                const Real tq[3] = {component(w, 3), component(w, 4), component(w, 5)};
                Real d2[3], oa[3], ob[3], x_ab[3];
                cdiff2(c, na, nb, d2);
                plane_axis(na, oa); plane_axis(nb, ob);
                cross3(oa, ob, x_ab);                                   // om_a x om_b
                const Real hs = dot3(F, d2), t_ab = dot3(tq, x_ab);
                wH[2 * pp] = hs + (jb < ja ? -t_ab : 0.0);             // F_dq(a; b): tau . (om_b x om_a), b before a
                wH[2 * pp + 1] = hs + (ja < jb ? t_ab : 0.0);          // F_dq(b; a): tau . (om_a x om_b), a before b
                if (d2w) {   // second-derivative kernel: sum_o w_o F_dqdq(o; a, b) and -dt/2 sum_o w_o F_dudq(o, u; .)
                    Real acc = 0.0;
                    const int n1 = ja <= jb ? na : nb, n2 = ja <= jb ? nb : na, j2 = ja <= jb ? jb : ja;   // n1 at or before n2
                    Real o1[3], o2[3];
                    plane_axis(n1, o1); plane_axis(n2, o2);
                    for (int no = P.cu_off[c]; no < P.cu_off[c + 1]; no++) {
                        const int ko = P.dh_cfg[no];
                        if (ko >= P.nd) continue;
                        Real d3[3];
                        cdiff3(c, na, nb, no, d3);
                        Real term = dot3(F, d3);
                        if (j2 < P.dh_joint[no]) {                       // both before o: d2 om_o = om_1 x (om_2 x om_o)
                            Real oo[3], u1[3], u2[3];
                            plane_axis(no, oo);
                            cross3(o2, oo, u1); cross3(o1, u1, u2);
                            term += dot3(tq, u2);
                        }
                        acc += d2w[ko] * term;
                    }
                    S[P.e_o_wT + pp] = acc;
                    Real *Hu = S + P.e_o_Hu;
                    for (int s6 = 0; s6 < 6; s6++) {
                        const int in = P.wr_in[6 * w + s6];
                        if (in < 0) continue;
                        // F_dudq(o = b, u; a) and F_dudq(o = a, u; b): d2p component (symmetric) or the axis derivative
                        const Real to_b = s6 < 3 ? d2[s6] : (ja < jb ? x_ab[s6 - 3] : 0.0);
                        const Real to_a = s6 < 3 ? d2[s6] : (jb < ja ? -x_ab[s6 - 3] : 0.0);
                        if (kb < P.nd) lds_add(&Hu[ka * P.nu + in], -0.5 * dt * d2w[kb] * to_b);
                        if (na != nb && ka < P.nd) lds_add(&Hu[kb * P.nu + in], -0.5 * dt * d2w[ka] * to_a);
                    }
                }
            }
        }
        TG_SYNC();
    }

    // ---- plane constraints: derivatives of the world normal n = R_plane n_local w.r.t. the joints of the plane frame's
    //      path: d n/dq_a = w_a x n, d2 n/dq_a dq_b = w_a x (w_b x n) (a at or before b), ... (w: world axis of a rotary
    //      joint; prismatic joints and joints that only move the point frame contribute nothing)
    TG_HD void plane_axis(int n, Real *w) const {
        const int j = P.dh_joint[n];
        w[0] = w[1] = w[2] = 0.0;
        if (j < 0 || !(P.dh_side[n] & 1) || P.j_kind[j] < TG_RX) return;
        const Real *gj = S + P.o_G + 12 * j;
        const int ax = P.j_kind[j] - TG_RX;
        w[0] = gj[ax]; w[1] = gj[4 + ax]; w[2] = gj[8 + ax];
    }
    TG_HD static void cross3(const Real *a, const Real *b, Real *r) {
        r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
    }
    TG_HD static Real dot3(const Real *a, const Real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    // k-th derivative of the normal w.r.t. the dh items n[0..k-1] (any order)
    TG_HD void plane_dn(int c, int k, const int *n, Real *out) const {
        int o[3] = {k > 0 ? n[0] : 0, k > 1 ? n[1] : 0, k > 2 ? n[2] : 0};
        auto later = [&](int x, int y) { return P.dh_joint[x] > P.dh_joint[y]; };
        if (k > 1 && later(o[0], o[1])) { const int t_ = o[0]; o[0] = o[1]; o[1] = t_; }
        if (k > 2 && later(o[1], o[2])) { const int t_ = o[1]; o[1] = o[2]; o[2] = t_; }
        if (k > 1 && later(o[0], o[1])) { const int t_ = o[0]; o[0] = o[1]; o[1] = t_; }
        const Real *nw = S + P.o_nE + 3 * c;
        Real cur[3] = {nw[0], nw[1], nw[2]};
        for (int i = k - 1; i >= 0; i--) {   // innermost cross product belongs to the joint farthest down the path
            Real w[3], nxt[3];
            plane_axis(o[i], w);
            cross3(w, cur, nxt);
            cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        }
        out[0] = cur[0]; out[1] = cur[1]; out[2] = cur[2];
    }
    // Leibniz expansion of d^k (n . D) over the dh items n[0..k-1], D = p_plane - p_point (plane.c:28-167)
    TG_HD Real plane_dk(int c, int k, const int *n) const {
        Real acc = 0.0;
        for (int mask = 0; mask < (1 << k); mask++) {
            int a[3], b[3], na = 0, nb = 0;
            for (int i = 0; i < k; i++) { if (mask & (1 << i)) a[na++] = n[i]; else b[nb++] = n[i]; }
            Real dn[3], dd[3];
            plane_dn(c, na, a, dn);
            if (nb == 0) {
                const Real *pa = S + P.o_pE + 3 * P.c_e1[c], *pb = S + P.o_pE + 3 * P.c_e2[c];
                dd[0] = pa[0] - pb[0]; dd[1] = pa[1] - pb[1]; dd[2] = pa[2] - pb[2];
            } else if (nb == 1) cdiff1(c, b[0], dd);
            else if (nb == 2) cdiff2(c, b[0], b[1], dd);
            else cdiff3(c, b[0], b[1], b[2], dd);
            acc += dot3(dn, dd);
        }
        return acc;
    }

    // h_c,dqdq for two dependent configs given by their dh items (distance.c:65-98, point.c:40-46)
    TG_HD Real con_d2(int c, int n1, int n2) const {
        if (has_plane() && P.c_type[c] == TG_CONSTRAINT_PLANE) { const int n[2] = {n1, n2}; return plane_dk(c, 2, n); }
        Real v12[3];
        cdiff2(c, n1, n2, v12);
        if (P.c_type[c] == TG_CONSTRAINT_POINT) return v12[P.c_comp[c]];
        Real v1[3], v2[3];
        cdiff1(c, n1, v1); cdiff1(c, n2, v2);
        const Real *a = S + P.o_pE + 3 * P.c_e1[c], *b = S + P.o_pE + 3 * P.c_e2[c];
        Real h = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2] + (a[0] - b[0]) * v12[0] + (a[1] - b[1]) * v12[1] + (a[2] - b[2]) * v12[2];
        if ((P.dh_side[n1] & 4) && n1 == n2) h -= 1.0;
        return 2.0 * h;
    }
    // h_c,dqdqdq (distance.c:100-133, point.c:48-54); zero when any argument is the string-length config
    TG_HD Real con_d3(int c, int n1, int n2, int n3) const {
        if (P.dh_joint[n1] < 0 || P.dh_joint[n2] < 0 || P.dh_joint[n3] < 0) return 0.0;
        if (has_plane() && P.c_type[c] == TG_CONSTRAINT_PLANE) { const int n[3] = {n1, n2, n3}; return plane_dk(c, 3, n); }
        Real v123[3];
        cdiff3(c, n1, n2, n3, v123);
        if (P.c_type[c] == TG_CONSTRAINT_POINT) return v123[P.c_comp[c]];
        Real v1[3], v2[3], v3[3], v12[3], v13[3], v23[3];
        cdiff1(c, n1, v1); cdiff1(c, n2, v2); cdiff1(c, n3, v3);
        cdiff2(c, n1, n2, v12); cdiff2(c, n1, n3, v13); cdiff2(c, n2, n3, v23);
        const Real *a = S + P.o_pE + 3 * P.c_e1[c], *b = S + P.o_pE + 3 * P.c_e2[c];
        return 2.0 * (v1[0] * v23[0] + v1[1] * v23[1] + v1[2] * v23[2] + v2[0] * v13[0] + v2[1] * v13[1] + v2[2] * v13[2] +
                      v3[0] * v12[0] + v3[1] * v12[1] + v3[2] * v12[2] +
                      (a[0] - b[0]) * v123[0] + (a[1] - b[1]) * v123[1] + (a[2] - b[2]) * v123[2]);
    }

    // third-order Lagrangian pieces of one body for an ORDERED triple of path items (x, y, o)
    struct Third { double q, dx, dy, dO, eo, ey, ex; };
    TG_HD static int sym(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }  // packed lower triangle
    TG_HD void sort2(int &a, int &b) const { if (a > b) { const int t_ = a; a = b; b = t_; } }
    TG_HD void vel2(int x, int y, double *out) const {  // d2v/dq_x dq_y = [W_min, J_max]
        sort2(x, y);
        bracket(S + P.o_W + 6 * x, S + P.o_J + 6 * y, out);
    }
    TG_HD bool jac1(int x, int y, double *out) const {  // dJ_x/dq_y = [J_x, J_y] if x < y
        if (!(x < y)) { for (int m = 0; m < 6; m++) out[m] = 0.0; return false; }
        bracket(S + P.o_J + 6 * x, S + P.o_J + 6 * y, out);
        return true;
    }
    TG_HD double ldqq(const double *I, const double *v, int x, int y, int z, const double *v2yz) const {
        // L_ddqdqdq(dq x; q y, q z) (system.c:336-393)
        const double *Jx = S + P.o_J + 6 * x;
        double t1[6], t2[6], acc = inner6(I, Jx, v2yz);
        if (jac1(x, y, t1)) acc += inner6(I, t1, S + P.o_W + 6 * z);
        if (jac1(x, z, t1)) acc += inner6(I, t1, S + P.o_W + 6 * y);
        int a = y, b = z;
        sort2(a, b);
        if (x < a) { bracket(Jx, S + P.o_J + 6 * a, t1); bracket(t1, S + P.o_J + 6 * b, t2); acc += inner6(I, t2, v); }
        return acc;
    }
    TG_HD double lddq(const double *I, int x, int y, int z) const {  // L_ddqddqdq(dq x, dq y; q z) (system.c:491-530)
        double t1[6], acc = 0.0;
        if (jac1(x, z, t1)) acc += inner6(I, t1, S + P.o_J + 6 * y);
        if (jac1(y, z, t1)) acc += inner6(I, S + P.o_J + 6 * x, t1);
        return acc;
    }
    TG_HD Third third_order(int b, int x, int y, int o) const {
        const double *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b;
        double vxy[6], vxo[6], vyo[6], t1[6], t2[6];
        vel2(x, y, vxy); vel2(x, o, vxo); vel2(y, o, vyo);
        int a = x, bb = y, c = o;
        sort2(a, bb); sort2(bb, c); sort2(a, bb);
        bracket(S + P.o_W + 6 * a, S + P.o_J + 6 * bb, t1);
        bracket(t1, S + P.o_J + 6 * c, t2);
        const double *Ja = S + P.o_J + 6 * a, *Jb = S + P.o_J + 6 * bb, *Jc = S + P.o_J + 6 * c;
        // w_a x (w_b x v_c): third derivative of the body position, for -V_qqq (gravity.c:67-94)
        const double ux = Jb[4] * Jc[2] - Jb[5] * Jc[1], uy = Jb[5] * Jc[0] - Jb[3] * Jc[2], uz = Jb[3] * Jc[1] - Jb[4] * Jc[0];
        const double gx = Ja[4] * uz - Ja[5] * uy, gy = Ja[5] * ux - Ja[3] * uz, gz = Ja[3] * uy - Ja[4] * ux;
        Third r;
        r.q = inner6(I, S + P.o_W + 6 * x, vyo) + inner6(I, S + P.o_W + 6 * y, vxo) + inner6(I, S + P.o_W + 6 * o, vxy) +
              inner6(I, v, t2) + I[0] * (gam[0] * gx + gam[1] * gy + gam[2] * gz);   // L_dqdqdq (system.c:204-268)
        r.dx = ldqq(I, v, x, y, o, vyo); r.dy = ldqq(I, v, y, x, o, vxo); r.dO = ldqq(I, v, o, x, y, vxy);
        r.eo = lddq(I, x, y, o); r.ey = lddq(I, x, o, y); r.ex = lddq(I, y, o, x);
        return r;
    }

    TG_HD void deriv2z(bool on, CArgs &A, size_t t) {
        const int nq = P.nq, nd = P.nd, nu = P.nu, nc = P.nc, nf = P.nf;
        const int ld = P.d_aug_ld, R = P.d_nrhs, hl = nq | 1;   // odd row stride of the H tables: no LDS bank conflicts
        double *AUG = S + P.d_o_AUG, *T22 = S + P.d_o_T22;
        double *H11 = S + P.e_o_H11, *H12 = S + P.e_o_H12, *H22 = S + P.e_o_H22, *G1 = S + P.e_o_G1;
        double *w = S + P.e_o_w, *zq = S + P.e_o_zq, *zp = S + P.e_o_zp, *vec = S + P.e_o_vec;
        const int c_p1 = nf + nq, c_ex = nf + R;
        // helper waves (nw > 1): everything wave-scoped is wave 0's; the other waves join the flat pair loops and the HZ tiles
        const bool w0 = wave == 0;
        bool ok = deriv1_solve(on, true);
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        long long d2t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        long long d2last = (long long)__builtin_amdgcn_s_memtime();
#define TG_D2STAMP(i) do { long long t_ = (long long)__builtin_amdgcn_s_memtime(); d2t[i] += t_ - d2last; d2last = t_; } while (0)
#else
#define TG_D2STAMP(i) ((void)0)
#endif
        if (w0) {
        if (on) {
            TG_FOR(i, nq * nc) G1[i] = 0.0;
            TG_FOR(i, nd) { zq[i] = A.z[t * P.nX + i]; zp[i] = A.z[t * P.nX + nq + i]; }
        }
        TG_SYNC();
        // r = z_Qd + D2D2L2[:nd] z_p  (kept in vec[0..nd)), then w = Kinv^T [r; z_lambda]
        if (on) TG_FOR(i, nd) {
            double r = zq[i];
            for (int o = 0; o < nd; o++) r += T22[i * nd + o] * zp[o];
            vec[i] = r;
        }
        if (on) TG_FOR(c, nc) vec[nd + c] = A.zl ? A.zl[t * nc + c] : 0.0;   // seed on the multiplier rows: lambda1'' (trep.h:439-473)
        TG_SYNC();
        // the two D.D2L2 tables of the first-derivative solve are dead now; the H tables take their place.
        // H11 and H22 are symmetric and stored packed (lower triangle); H12 is full with an odd row stride.
        if (on) {
            TG_FOR(i, nq * (nq + 1) / 2) { H11[i] = 0.0; H22[i] = 0.0; }
            TG_FOR(i, nq * hl) H12[i] = 0.0;
        }
        TG_SYNC();
        if (on) TG_FOR(j, nf) {
            double acc = 0.0;
            for (int i = 0; i < nf; i++) {
                const double kinv = j < nd ? -AUG[i * ld + c_p1 + j] : AUG[i * ld + c_ex + (j - nd)];
                acc += vec[i] * kinv;
            }
            w[j] = acc;
        }
        TG_SYNC();
        }
        // ---- constraints at q1: G1 and the lambda-weighted third derivative (calc_h1_deriv2 :1559-1595) -----
        int cp_lo, cp_hi;
        wave_part(0, P.wc_split, P.n_cpair, cp_lo, cp_hi);
        const int *cp4 = nw > 1 ? P.wcp4 : P.cpair4;
        if (nc) {
            if (w0) {
                pose_sweep(on, 1);
                attach_points(on, false, true);
            }
            TG_D2STAMP(4);
            if (P.o_cps >= 0) {
                // The sums over the output index o are pushed into prefix / suffix sums along each end point's joint
                // path, which makes the w-contracted third derivative of a constraint O(1) per (a, b) pair instead
                // of a loop over o.  With D_t = d p_E / d q_t, Om_t = world axis of joint t (0 if prismatic) and t
                // ordered root-first:  d2 p(x<=y) = Om_x x D_y,  d3 p(x<=y<=z) = Om_x x (Om_y x D_z), so for x <= y
                //   sum_o w_o d2p(x,o)   = Om_x x SD_x + PW_x x D_x
                //   sum_o w_o d3p(x,y,o) = Om_x x (Om_y x SD_y) + Om_x x ((PW_y - PW_x) x D_y) + PW_x x (Om_x x D_y)
                // with SD_t = sum_{u >= t} w_u D_u and PW_t = sum_{u < t} w_u Om_u (w_u = 0 for kinematic configs).
                double *cps = S + P.o_cps;
                auto item = [&](int n, int E, double *D, double *Om, double &wt) {   // D, Om, weight of dh item n for end point E
                    const int *rec = P.dh_pack + 8 * (size_t)n;
                    const int k = rec[1], oj = rec[2], kind = (rec[3] >> 8) & 0xFF;
                    dpos_rec(rec[4 + E], oj, kind, D);
                    if (kind >= TG_RX) { const double *gj = S + P.o_G + oj; const int ax = kind - TG_RX; Om[0] = gj[ax]; Om[1] = gj[4 + ax]; Om[2] = gj[8 + ax]; }
                    else { Om[0] = Om[1] = Om[2] = 0.0; }
                    wt = k < nd ? w[k] : 0.0;
                };
                auto cross = [](const double *a, const double *b, double *r) {
                    r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
                };
                if (w0 && on) TG_FOR(ce, 2 * nc) {
                    const int t0 = P.cpath_off[ce], t1 = P.cpath_off[ce + 1], E = ce & 1;
                    double pw[3] = {0, 0, 0}, sd[3] = {0, 0, 0}, D[3], Om[3], wt;
                    for (int t = t0; t < t1; t++) {
                        item(P.cpath_items[t], E, D, Om, wt);
                        cps[6 * t + 3] = pw[0]; cps[6 * t + 4] = pw[1]; cps[6 * t + 5] = pw[2];
                        pw[0] += wt * Om[0]; pw[1] += wt * Om[1]; pw[2] += wt * Om[2];
                    }
                    for (int t = t1 - 1; t >= t0; t--) {
                        item(P.cpath_items[t], E, D, Om, wt);
                        sd[0] += wt * D[0]; sd[1] += wt * D[1]; sd[2] += wt * D[2];
                        cps[6 * t] = sd[0]; cps[6 * t + 1] = sd[1]; cps[6 * t + 2] = sd[2];
                    }
                    double *vw = cps + 6 * P.n_cpath + 3 * ce;       // sum_o w_o D_o over the whole path
                    vw[0] = sd[0]; vw[1] = sd[1]; vw[2] = sd[2];
                }
                TG_WSYNC();
                TG_D2STAMP(5);
                // per dh item a: v_a, V2w_a (differences end point 1 - end point 2); returns false if a has no joint
                auto first_second = [&](int c, int n, double *va, double *V2) {
                    va[0] = va[1] = va[2] = V2[0] = V2[1] = V2[2] = 0.0;
                    for (int E = 0; E < 2; E++) {
                        const int t = P.dh_pos[2 * n + E];
                        if (t < 0) continue;
                        const double *q = cps + 6 * (P.cpath_off[2 * c + E] + t);
                        double D[3], Om[3], wt, x1[3], x2[3];
                        item(n, E, D, Om, wt);
                        cross(Om, q, x1); cross(q + 3, D, x2);
                        const double sg = E ? -1.0 : 1.0;
                        for (int m = 0; m < 3; m++) { va[m] += sg * D[m]; V2[m] += sg * (x1[m] + x2[m]); }
                    }
                };
                if (on) for (int pp = cp_lo + tg_opaque(lane); pp < cp_hi; pp += TEAM) {
                    const int *pw4 = cp4 + 4 * (size_t)pp;
                    const int c = pw4[0], na = pw4[1], nb = pw4[2], ka = pw4[3] & 0xFFFF, kb = pw4[3] >> 16;
                    const int *rc = P.dh_pack + 8 * (size_t)na;
                    const int type = (rc[3] >> 16) & 0xFF, comp = rc[3] >> 24;
                    double va[3], vb[3], V2a[3], V2b[3], vab[3] = {0, 0, 0}, V3[3] = {0, 0, 0}, Vw[3] = {0, 0, 0};
                    first_second(c, na, va, V2a);
                    first_second(c, nb, vb, V2b);
                    for (int E = 0; E < 2; E++) {
                        const double sg = E ? -1.0 : 1.0;
                        const double *vw = cps + 6 * P.n_cpath + 3 * (2 * c + E);
                        for (int m = 0; m < 3; m++) Vw[m] += sg * vw[m];
                        const int ta = P.dh_pos[2 * na + E], tb = P.dh_pos[2 * nb + E];
                        if (ta < 0 || tb < 0) continue;
                        const int nx = ta <= tb ? na : nb, ny = ta <= tb ? nb : na;
                        const int base = P.cpath_off[2 * c + E];
                        const double *qx = cps + 6 * (base + (ta <= tb ? ta : tb)), *qy = cps + 6 * (base + (ta <= tb ? tb : ta));
                        double Dx[3], Ox[3], Dy[3], Oy[3], wt, d2[3], t1[3], t2[3], t3[3], dpw[3];
                        item(nx, E, Dx, Ox, wt); item(ny, E, Dy, Oy, wt);
                        cross(Ox, Dy, d2);                                   // d2 p(x, y)
                        cross(Oy, qy, t1); cross(Ox, t1, t2);                // Om_x x (Om_y x SD_y)
                        for (int m = 0; m < 3; m++) dpw[m] = qy[3 + m] - qx[3 + m];
                        cross(dpw, Dy, t1); cross(Ox, t1, t3);               // Om_x x ((PW_y - PW_x) x D_y)
                        double t4[3];
                        cross(qx + 3, d2, t4);                               // PW_x x (Om_x x D_y)
                        for (int m = 0; m < 3; m++) { vab[m] += sg * d2[m]; V3[m] += sg * (t2[m] + t3[m] + t4[m]); }
                    }
                    double acc;
                    if (type == TG_CONSTRAINT_POINT) acc = V3[comp];
                    else {
                        const double *pa = S + P.o_pE + rc[4], *pb = S + P.o_pE + rc[5];
                        const double d0 = pa[0] - pb[0], d1 = pa[1] - pb[1], d2_ = pa[2] - pb[2];
                        acc = 2.0 * (va[0] * V2b[0] + va[1] * V2b[1] + va[2] * V2b[2] + vb[0] * V2a[0] + vb[1] * V2a[1] + vb[2] * V2a[2] +
                                     Vw[0] * vab[0] + Vw[1] * vab[1] + Vw[2] * vab[2] + d0 * V3[0] + d1 * V3[1] + d2_ * V3[2]);
                    }
                    lds_add(&H11[sym(ka, kb)], S[P.o_lam + c] * acc);
                }
                TG_D2STAMP(6);
                if (on) TG_FORW(na, P.n_dh) {  // G1[ka][c] = sum_o w_o h_c,dqdq(ka, o) = 2 (v_a . Vw + d . V2w_a)
                    const int *rc = P.dh_pack + 8 * (size_t)na;
                    const int c = rc[0], type = (rc[3] >> 16) & 0xFF, comp = rc[3] >> 24;
                    double va[3], V2a[3], Vw[3];
                    first_second(c, na, va, V2a);
                    const double *v1 = cps + 6 * P.n_cpath + 3 * (2 * c), *v2 = v1 + 3;
                    for (int m = 0; m < 3; m++) Vw[m] = v1[m] - v2[m];
                    double g;
                    if (type == TG_CONSTRAINT_POINT) g = V2a[comp];
                    else {
                        const double *pa = S + P.o_pE + rc[4], *pb = S + P.o_pE + rc[5];
                        g = 2.0 * (va[0] * Vw[0] + va[1] * Vw[1] + va[2] * Vw[2] + (pa[0] - pb[0]) * V2a[0] + (pa[1] - pb[1]) * V2a[1] + (pa[2] - pb[2]) * V2a[2]);
                    }
                    G1[rc[1] * nc + c] = g;
                }
                TG_WSYNC();
                TG_D2STAMP(7);
            } else {
            // one lane per (constraint, a <= b) pair of the flat list; third derivatives are symmetric in (a, b).
                // Several constraints reach the same entry: LDS atomics (one wavefront, fixed order, deterministic).
                TG_WSYNC();
                if (on) for (int pp = cp_lo + tg_opaque(lane); pp < cp_hi; pp += TEAM) {
                    const int *pw = cp4 + 4 * (size_t)pp;
                    const int c = pw[0], na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                    const int n0 = P.cu_off[c], n1 = P.cu_off[c + 1];
                    double acc = 0.0;
                    for (int no = n0; no < n1; no++) {
                        const int ko = P.dh_cfg[no];
                        if (ko >= nd) continue;
                        acc += w[ko] * con_d3(c, na, nb, no);
                    }
                    const double val = S[P.o_lam + c] * acc;
                    lds_add(&H11[sym(ka, kb)], val);
                }
                if (on) TG_FORW(na, P.n_dh) {  // G1[ka][c] = sum_o w_o h_c,dqdq(ka, o)
                    const int c = P.dh_c[na], n0 = P.cu_off[c], n1 = P.cu_off[c + 1];
                    double g = 0.0;
                    for (int no = n0; no < n1; no++) {
                        const int ko = P.dh_cfg[no];
                        if (ko < nd) g += w[ko] * con_d2(c, na, no);
                    }
                    G1[P.dh_cfg[na] * nc + c] = g;
                }
                TG_WSYNC();
            }
            // ---- constraints at q2: H22 -= sum_c w_lambda,c DDh2[c] (calc_h2_deriv2 :1597-1622) -------------
            if (w0) {
                pose_sweep(on, 2);
                attach_points(on, false, true);
            }
            TG_WSYNC();
            TG_D2STAMP(8);
            if (on) for (int pp = cp_lo + tg_opaque(lane); pp < cp_hi; pp += TEAM) {
                const int *pw = cp4 + 4 * (size_t)pp;
                const int c = pw[0], na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                const double val = -w[nd + c] * con_d2(c, na, nb);
                lds_add(&H22[sym(ka, kb)], val);
            }
            TG_WSYNC();     // (the midpoint sweep below overwrites the poses the pair loop reads)
        }
        TG_D2STAMP(0);
        if (w0) {
        // ---- midpoint: third-order discrete-Lagrangian tables contracted on the fly --------------------------
        if (n_wrenches()) {   // the point forces' second derivatives are contracted while the midpoint poses are alive
            if (on) TG_FOR(i, nq * nu) S[P.e_o_Hu + i] = 0.0;
            TG_SYNC();
            d2w = w;
        }
        if (has_damper()) d2w = w;
        eval_midpoint(on);
        d2w = nullptr;
        if (has_damper() && on) TG_FOR(pp, n_spair()) {   // dampers: dt/4 F_dqdq + sa/2 F_ddqdq(.; a, b) + sb/2 F_ddqdq(.; b, a)
            const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
            const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
            const double Pq = 0.25 * dt * S[P.e_o_sT + 2 * pp], R = S[P.e_o_sT + 2 * pp + 1];
            lds_add(&H12[ka * hl + kb], -Pq);                    // slots (1, 2): -R/2 + R/2
            if (pw[1] != pw[2]) lds_add(&H12[kb * hl + ka], -Pq);
            lds_add(&H11[sym(ka, kb)], -(Pq - R));               // slots (1, 1): -R/2 - R/2
            lds_add(&H22[sym(ka, kb)], -(Pq + R));               // slots (2, 2)
        }
        if (on) TG_FOR(pp, n_wpair()) {   // D_a D_b fm2 = dt/4 F_dqdq in all three slot combinations (midpointvi.c:1473-1497)
            const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + n_spair() + pp);
            const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
            const double val = -0.25 * dt * S[P.e_o_wT + pp];
            lds_add(&H12[ka * hl + kb], val);
            if (pw[1] != pw[2]) lds_add(&H12[kb * hl + ka], val);
            lds_add(&H11[sym(ka, kb)], val);
            lds_add(&H22[sym(ka, kb)], val);
        }
        if (has_cs() && P.n_ncs && on) TG_FOR(i, nd) {   // -dt V(q_mid): every third derivative D_a D_b D_o L2 on config i is -dt/8 V_dqdqdq
            const double val = (zp[i] - w[i]) * (-0.125 * dt * cs_d3(i, qval(0, i)));
            lds_add(&H12[i * hl + i], val);
            lds_add(&H11[sym(i, i)], val);
            lds_add(&H22[sym(i, i)], val);
        }
        }
        TG_WSYNC();
        TG_D2STAMP(1);
        const double c8 = 0.125 * dt, c2 = 0.5 / dt;
        if (P.n_tchunk > 0) {
            // The sums over the output item o are pushed into prefix / suffix sums along every body's item path, so
            // the (w, z_p)-contracted third-order terms cost O(1) per (x, y) pair instead of a loop over o.
            // Contracting T(sa,sb,so) = q + sa dx + sb dy + sa sb eo + so (dO + sa ey + sb ex) with -w_o at so = -1 and
            // z_p,o at so = +1 leaves two weights per item, alpha_o = z_p,o - w_o on the so-free terms and
            // beta_o = z_p,o + w_o on the others (0 for kinematic configs).  Per item t (root-first):
            //   SJa_t = sum_{o >= t} alpha_o J_o,  PWa_t = sum_{o < t} alpha_o W_o,  PJb_t = sum_{o < t} beta_o J_o
            // and per body TWa, TJa, TJb.  Every term of system.c:204-530 is linear in the slot that o occupies; which
            // slot that is depends on the position of o relative to x and y, which is what the three ranges
            // (o >= hi, lo <= o < hi, o < lo) below are.
            double *tps = S + P.o_tps;
            for (int ci = 0; ci < P.n_tchunk; ci++) {
                const int b0 = P.tchunk[ci], b1 = P.tchunk[ci + 1];
                const int it0 = P.b_item_off[b0], nit = P.b_item_off[b1] - it0;
                if (on) TG_FORW(idx, 6 * (b1 - b0)) {
                    const int b = b0 + idx / 6, m = idx % 6;
                    const int first = P.b_item_off[b], last = P.b_item_off[b + 1];
                    double pwa = 0.0, pjb = 0.0, sja = 0.0;
                    for (int k = first; k < last; k++) {
                        const int cfg = P.it_pack[4 * (size_t)k + 3] & 0xFFFF;
                        const double al = cfg < nd ? zp[cfg] - w[cfg] : 0.0, be = cfg < nd ? zp[cfg] + w[cfg] : 0.0;
                        double *q = tps + 18 * (k - it0);
                        q[6 + m] = pwa; q[12 + m] = pjb;
                        pwa += al * S[P.o_W + 6 * k + m]; pjb += be * S[P.o_J + 6 * k + m];
                    }
                    for (int k = last - 1; k >= first; k--) {
                        const int cfg = P.it_pack[4 * (size_t)k + 3] & 0xFFFF;
                        const double al = cfg < nd ? zp[cfg] - w[cfg] : 0.0;
                        sja += al * S[P.o_J + 6 * k + m];
                        tps[18 * (k - it0) + m] = sja;
                    }
                    double *tb = tps + 18 * nit + 18 * (b - b0);
                    tb[m] = pwa; tb[6 + m] = sja; tb[12 + m] = pjb;      // TWa, TJa, TJb
                }
                TG_WSYNC();
                // one lane per UNORDERED pair x <= y (the flat (item, item) list of the first-derivative tables): every ingredient below is
                // either symmetric under x <-> y (q, eo, dO) or swaps with its partner (dx <-> dy, ex <-> ey), so the pair (y, x) of the
                // H12 table comes out of the same evaluation -- half the lanes' work of the ordered list (tri4) used until round 2
                int tp_lo, tp_hi;
                wave_part(P.b_pair_off[b0], nw > 1 ? P.wt_split[ci] : 0, P.b_pair_off[b1], tp_lo, tp_hi);
                const int *ta = nw > 1 ? P.wt_a : P.pair_a, *tb_ = nw > 1 ? P.wt_b : P.pair_b;
                if (on) for (int pp = tp_lo + tg_opaque(lane); pp < tp_hi; pp += TEAM) {
                    const int x = ta[pp], y = tb_[pp], kx = P.it_cfg[x], ky = P.it_cfg[y], b = P.it_body[x];
                    const double *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b;
                    const int lo = x < y ? x : y, hi = x < y ? y : x;
                    const double *Jx = S + P.o_J + 6 * x, *Jy = S + P.o_J + 6 * y, *Wx = S + P.o_W + 6 * x, *Wy = S + P.o_W + 6 * y;
                    const double *Jlo = S + P.o_J + 6 * lo, *Jhi = S + P.o_J + 6 * hi, *Wlo = S + P.o_W + 6 * lo;
                    const double *qx = tps + 18 * (x - it0), *qy = tps + 18 * (y - it0);
                    const double *qlo = tps + 18 * (lo - it0), *qhi = tps + 18 * (hi - it0);
                    const double *TWa = tps + 18 * nit + 18 * (b - b0), *TJa = TWa + 6, *TJb = TWa + 12;
                    double t1[6], t2[6], t3[6];
                    // V2a_t = sum_o alpha_o d2v(t, o) = [W_t, SJa_t] + [PWa_t, J_t]
                    double V2x[6], V2y[6];
                    bracket(Wx, qx, V2x); bracket(qx + 6, Jx, t1);
                    for (int m = 0; m < 6; m++) V2x[m] += t1[m];
                    bracket(Wy, qy, V2y); bracket(qy + 6, Jy, t1);
                    for (int m = 0; m < 6; m++) V2y[m] += t1[m];
                    double v2xy[6];
                    bracket(Wlo, Jhi, v2xy);                                   // d2v(x, y)
                    // ---- q: L_dqdqdq (system.c:204-268) ----
                    double dS[6];
                    for (int m = 0; m < 6; m++) dS[m] = qlo[m] - qhi[m];        // sum over lo <= o < hi of alpha_o J_o
                    bracket(v2xy, qhi, t1);                                    // o >= hi
                    bracket(Wlo, dS, t2); bracket(t2, Jhi, t3);                // lo <= o < hi
                    for (int m = 0; m < 6; m++) t1[m] += t3[m];
                    bracket(qlo + 6, Jlo, t2); bracket(t2, Jhi, t3);           // o < lo
                    for (int m = 0; m < 6; m++) t1[m] += t3[m];
                    double Q = inner6(I, Wx, V2y) + inner6(I, Wy, V2x) + inner6(I, TWa, v2xy) + inner6(I, v, t1);
                    {   // gravity: m gam . sum_o alpha_o w_a x (w_b x lin_c), (a, b, c) = sorted (x, y, o)
                        auto cross = [](const double *p, const double *q_, double *r) {
                            r[0] = p[1] * q_[2] - p[2] * q_[1]; r[1] = p[2] * q_[0] - p[0] * q_[2]; r[2] = p[0] * q_[1] - p[1] * q_[0];
                        };
                        double u[3], g[3], gs[3] = {0, 0, 0}, pa[3];
                        cross(Jhi + 3, qhi, u); cross(Jlo + 3, u, g);          // o >= hi: w_lo x (w_hi x lin(SJa_hi))
                        for (int m = 0; m < 3; m++) gs[m] += g[m];
                        cross(dS + 3, Jhi, u); cross(Jlo + 3, u, g);           // lo <= o < hi: w_lo x (w_o x lin_hi)
                        for (int m = 0; m < 3; m++) gs[m] += g[m];
                        for (int m = 0; m < 3; m++) pa[m] = TJa[3 + m] - qlo[3 + m];   // sum over o < lo of alpha_o w_o
                        cross(Jlo + 3, Jhi, u); cross(pa, u, g);               // o < lo: w_o x (w_lo x lin_hi)
                        for (int m = 0; m < 3; m++) gs[m] += g[m];
                        Q += I[0] * (gam[0] * gs[0] + gam[1] * gs[1] + gam[2] * gs[2]);
                    }
                    // ---- dx, dy: L_ddqdqdq with the dq slot on x resp. y (system.c:336-393), alpha-weighted ----
                    double JxSx[6], JySy[6], Jxy[6];
                    bracket(Jx, qx, JxSx); bracket(Jy, qy, JySy);             // [J_t, SJa_t] = sum_{o > t} alpha_o [J_t, J_o]
                    double Dx = inner6(I, Jx, V2y) + inner6(I, JxSx, Wy), Dy = inner6(I, Jy, V2x) + inner6(I, JySy, Wx);
                    if (x < y) {
                        bracket(Jx, Jy, Jxy);
                        bracket(Jxy, qy, t1); bracket(Jx, dS, t2); bracket(t2, Jy, t3);     // (lo, hi) = (x, y): dS = SJa_x - SJa_y
                        for (int m = 0; m < 6; m++) t1[m] += t3[m];
                        Dx += inner6(I, Jxy, TWa) + inner6(I, t1, v);
                    } else if (y < x) {
                        bracket(Jy, Jx, Jxy);
                        bracket(Jxy, qx, t1); bracket(Jy, dS, t2); bracket(t2, Jx, t3);     // (lo, hi) = (y, x): dS = SJa_y - SJa_x
                        for (int m = 0; m < 6; m++) t1[m] += t3[m];
                        Dy += inner6(I, Jxy, TWa) + inner6(I, t1, v);
                    }
                    // ---- eo: L_ddqddqdq(dq x, dq y; q o), alpha-weighted (system.c:491-530) ----
                    const double Eo = inner6(I, JxSx, Jy) + inner6(I, Jx, JySy);
                    // ---- dO, ey, ex: o in a dq slot, beta-weighted ----
                    // (by lo / hi rather than x / y: choosing between two local arrays per lane would force both into scratch memory)
                    const double *Whi = S + P.o_W + 6 * hi;
                    double PJlo[6], PJhi[6];
                    bracket(qlo + 12, Jlo, PJlo); bracket(qhi + 12, Jhi, PJhi);   // [PJb_t, J_t] = sum_{o < t} beta_o [J_o, J_t]
                    bracket(PJlo, Jhi, t1);                                    // [[PJb_lo, J_lo], J_hi]
                    const double DO = inner6(I, TJb, v2xy) + inner6(I, PJlo, Whi) + inner6(I, PJhi, Wlo) + inner6(I, t1, v);
                    const double Ea = inner6(I, Jlo, PJhi), Eb = inner6(I, Jhi, PJlo);
                    double Ey = x < y ? Ea : Eb, Ex = x < y ? Eb : Ea;         // (x == y: the same value)
                    if (x < y) Ey += inner6(I, Jxy, TJb);
                    else if (y < x) Ex += inner6(I, Jxy, TJb);
                    const double q_ = c8 * Q, dx = 0.25 * Dx, dy = 0.25 * Dy, eo = c2 * Eo, dO = 0.25 * DO, ey = c2 * Ey, ex = c2 * Ex;
                    const double h11 = q_ - dx - dy + eo + dO - ey - ex, h12 = q_ - dx + dy - eo + dO - ey + ex, h22 = q_ + dx + dy + eo + dO + ey + ex;
                    lds_add(&H12[kx * hl + ky], h12);
                    if (x != y) lds_add(&H12[ky * hl + kx], q_ + dx - dy - eo + dO + ey - ex);          // the pair (y, x): dx <-> dy, ex <-> ey
                    lds_add(&H11[sym(kx, ky)], h11); lds_add(&H22[sym(kx, ky)], h22);
                }
                TG_WSYNC();
            }
        } else {
            // one lane per ordered (item x, item y) pair of the flat list over all bodies; the output index o runs over the
            // body's items.  Bodies share configs, so the accumulation uses LDS atomics like the Newton matrix.
            if (on && w0) TG_FOR(pp, P.n_tri) {     // (fallback: one wave)
                const int *pw = P.tri4 + 4 * (size_t)pp;
                const int x = pw[0], y = pw[1], kx = pw[2] & 0xFFFF, ky = pw[2] >> 16, b = pw[3];
                const int i0 = P.b_item_off[b], i1 = P.b_item_off[b + 1];
                double h11 = 0.0, h12 = 0.0, h22 = 0.0;
                for (int o = i0; o < i1; o++) {
                    const int ko = P.it_cfg[o];
                    if (ko >= nd) continue;
                    const Third T = third_order(b, x, y, o);
                    const double q = c8 * T.q, dx = 0.25 * T.dx, dy = 0.25 * T.dy, dO = 0.25 * T.dO;
                    const double eo = c2 * T.eo, ey = c2 * T.ey, ex = c2 * T.ex;
                    // T(sa,sb,so) = q + sa dx + sb dy + so dO + sa sb eo + sa so ey + sb so ex  (midpointvi.c:1122-1453)
                    const double t111 = q - dx - dy - dO + eo + ey + ex, t112 = q - dx - dy + dO + eo - ey - ex;
                    const double t121 = q - dx + dy - dO - eo + ey - ex, t122 = q - dx + dy + dO - eo - ey + ex;
                    const double t221 = q + dx + dy - dO + eo - ey - ex, t222 = q + dx + dy + dO + eo + ey + ex;
                    h11 += -w[ko] * t111 + zp[ko] * t112;
                    h12 += -w[ko] * t121 + zp[ko] * t122;
                    h22 += -w[ko] * t221 + zp[ko] * t222;
                }
                lds_add(&H12[kx * hl + ky], h12);
                if (x <= y) { lds_add(&H11[sym(kx, ky)], h11); lds_add(&H22[sym(kx, ky)], h22); }   // (y, x) gives the same value
            }
            TG_WSYNC();
        }
        TG_D2STAMP(2);
        // ---- assemble HZ, CB columns per pass ------------------------------------------------------------------
        // tangents: y_b = (x_b, e_i for a k2 variable) with x_b = AUG[0..nd)[nf+b]; l_b = AUG[nd..nf)[nf+b].
        // HZ is symmetric; the value computed for (row a, column b) is stored at [b][a] so that the lanes of a phase
        // write contiguous rows.  CB columns share every load of H22 / H12 (tangent products) and of the
        // tangent matrix (row products) and run as 2 CB independent accumulation chains.  CB = 8 when the three [CB][nq]
        // scratch tables fit the J / W area (dead by now: 12 n_items doubles), else 4 in the `vec` area.
        const int first_k2 = nq + nd + nu;
        auto assemble = [&](auto cb_tag, double *scratch) {
            constexpr int CB = decltype(cb_tag)::value;
            double *hy = scratch, *h12y = scratch + CB * nq, *g1l = scratch + 2 * CB * nq;     // [CB][nq] each
            for (int b0 = 0; b0 < R; b0 += CB) {
                const int nb = R - b0 < CB ? R - b0 : CB;
                if (on) TG_FOR(j, nq) {
                    double a22[CB], a12[CB], ag[CB];
#pragma unroll
                    for (int c = 0; c < CB; c++) { a22[c] = 0.0; a12[c] = 0.0; ag[c] = 0.0; }
#pragma unroll 2
                    for (int i2 = 0; i2 < nd; i2++) {
                        const double h22 = H22[sym(j, i2)], h12 = H12[j * hl + i2];
                        const double *yr = AUG + i2 * ld + nf + b0;
#pragma unroll
                        for (int c = 0; c < CB; c++) { const double yb = yr[c < nb ? c : 0]; a22[c] = fma(h22, yb, a22[c]); a12[c] = fma(h12, yb, a12[c]); }
                    }
#pragma unroll
                    for (int c = 0; c < CB; c++) if (c < nb && b0 + c >= first_k2) {
                        a22[c] += H22[sym(j, nd + (b0 + c - first_k2))]; a12[c] += H12[j * hl + nd + (b0 + c - first_k2)];
                    }
                    for (int c0 = 0; c0 < nc; c0 += 4) {       // multiplier tangents, four constraints at a time: loads first
                        double g[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) g[q] = c0 + q < nc ? G1[j * nc + c0 + q] : 0.0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const double *lr = AUG + (nd + (c0 + q < nc ? c0 + q : 0)) * ld + nf + b0;
#pragma unroll
                            for (int c = 0; c < CB; c++) ag[c] = fma(g[q], lr[c < nb ? c : 0], ag[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CB; c++) { hy[c * nq + j] = a22[c]; h12y[c * nq + j] = a12[c]; g1l[c * nq + j] = ag[c]; }
                }
                TG_SYNC();
                if (on) TG_FOR(a, R) {
                    double acc[CB], s12[CB];
                    // straight-line loop body (clamped row for the columns without an H12 term, weighted out below): a
                    // branch per column would split the body into basic blocks, each waiting for its own LDS reads
                    int hrow[CB];
#pragma unroll
                    for (int c = 0; c < CB; c++) { acc[c] = 0.0; s12[c] = 0.0; hrow[c] = (c < nb && b0 + c < nq) ? (b0 + c) * hl : 0; }
#pragma unroll 2
                    for (int i2 = 0; i2 < nd; i2++) {
                        const double x = AUG[i2 * ld + nf + a];
#pragma unroll
                        for (int c = 0; c < CB; c++) {
                            acc[c] = fma(x, hy[c * nq + i2], acc[c]);
                            s12[c] = fma(H12[hrow[c] + i2], x, s12[c]);
                        }
                    }
                    // row-dependent extras without divergent branches: clamped indices and 0/1 weights, so that all the LDS
                    // reads of the epilogue can be in flight together
                    const bool k2row = a >= first_k2, qrow = a < nq;
                    const int ak = k2row ? nd + (a - first_k2) : 0, aq = qrow ? a : 0;
                    const double wk = k2row ? 1.0 : 0.0, wq = qrow ? 1.0 : 0.0;
                    double xl[8];                                  // multiplier tangents of this row (constraints in groups of 8)
#pragma unroll
                    for (int q = 0; q < 8; q++) xl[q] = q < nc ? AUG[(nd + q) * ld + nf + a] : 0.0;
#pragma unroll
                    for (int c = 0; c < CB; c++) {
                        const int bcol = b0 + c;
                        if (c >= nb) break;
                        double v = acc[c] + wk * hy[c * nq + ak] + wq * (h12y[c * nq + aq] + g1l[c * nq + aq]);
                        if (bcol < nq) {                           // uniform over the wavefront
                            double sg = 0.0;
#pragma unroll
                            for (int q = 0; q < 8; q++) sg += (q < nc ? G1[bcol * nc + q] : 0.0) * xl[q];
                            for (int cc = 8; cc < nc; cc++) sg += G1[bcol * nc + cc] * AUG[(nd + cc) * ld + nf + a];
                            v += s12[c] + wk * H12[bcol * hl + ak] + sg + wq * H11[sym(aq, bcol)];
                        }
                        A.hz[(t * R + bcol) * R + a] = ok ? v : NAN;
                    }
                }
                TG_SYNC();
            }
        };
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_HZ_MFMA)
        // ---- the same assembly on the matrix cores (full-wave teams) --------------------------------------------------
        // HZ = Y2' H22 Y2 + N + N' + D with Y2 = [X; e_k2] the q2 tangents, N = [H12 Y2 + G1 L; 0] the rows of the q1 variables and
        // D the H11 block.  Per block of 16 columns b:  TB = H22 Y2[:, b] (staged in LDS), then every 16 x 16 output tile (a, b) is ONE
        // accumulator fed by chains of v_mfma_f64_16x16x4:  X[:, a]' (TB + H12[b, :]')  +  L[:, a]' G1[b, :]'  (S and N'), and for the tiles
        // with q1 rows  H12[a, :] X[:, b] + G1[a, :] L[:, b]  (N);  the unit rows / columns of the k2 variables and H11 are added per
        // element.  Operand layout of the instruction: A: lane holds A[lane & 15][lane >> 4], B: B[lane >> 4][lane & 15], C / D: column
        // lane & 15, rows (lane >> 4) + 4 r.  Up to GA tiles are accumulated side by side: a dependent MFMA issues every ~200 cycles,
        // independent ones every ~64 (tools/micro/mfma_f64_rate.hip).  Rows of a tile are stored as HZ[a][b] (= HZ[b][a]): 16 lanes
        // write 128 contiguous bytes.
        typedef double hz4 __attribute__((ext_vector_type(4)));
        auto assemble_mfma = [&](double *TB) {
            constexpr int GA = 3;
            const int l15 = lane & 15, l4 = lane >> 4;
            const int n_jt = (nq + 15) >> 4, n_bt = (R + 15) >> 4, n_k = (nd + 3) >> 2, n_kc = (nc + 3) >> 2;
            // HZ is symmetric: block column bt only computes the tiles on and above the diagonal (row tiles 0 .. bt) and stores the
            // off-diagonal ones a second time, transposed.  Blocks are dealt to the waves largest first in snake order (helper waves:
            // 8 and 7 of the 15 tiles of an 80 x 80 matrix), each wave with its own TB.
            for (int bi = 0; bi < n_bt; bi++) {
                const int ph = bi % (2 * nw);
                if ((ph < nw ? ph : 2 * nw - 1 - ph) != wave) continue;
                const int bt = n_bt - 1 - bi, b0 = 16 * bt;
                const int bcol = b0 + l15;
                const bool bin = bcol < R, bq = bcol < nq;
                if (on) for (int jt0 = 0; jt0 < n_jt; jt0 += GA) {
                    hz4 acc[GA];
#pragma unroll
                    for (int g = 0; g < GA; g++) acc[g] = hz4{0.0, 0.0, 0.0, 0.0};
                    for (int ks = 0; ks < n_k; ks++) {
                        const int i2 = 4 * ks + l4;
                        const double bw = (i2 < nd && bin) ? AUG[i2 * ld + nf + bcol] : 0.0;
                        double av[GA];
#pragma unroll
                        for (int g = 0; g < GA; g++) { const int j = 16 * (jt0 + g) + l15; av[g] = (j < nq && i2 < nd) ? H22[sym(j, i2)] : 0.0; }
#pragma unroll
                        for (int g = 0; g < GA; g++) if (jt0 + g < n_jt) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bw, acc[g], 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < GA; g++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int j = 16 * (jt0 + g) + l4 + 4 * r;
                            if (j < nq) {
                                double v = acc[g][r];
                                if (bin && bcol >= first_k2) v += H22[sym(j, nd + bcol - first_k2)];
                                TB[j * 16 + l15] = v;
                            }
                        }
                    }
                }
                TG_SYNC();
                TG_D2STAMP(9);
                if (on) for (int at0 = 0; at0 <= bt; at0 += GA) {
                    const int n_at = bt + 1;                      // row tiles of this block column
                    hz4 acc[GA];
#pragma unroll
                    for (int g = 0; g < GA; g++) acc[g] = hz4{0.0, 0.0, 0.0, 0.0};
                    for (int ks = 0; ks < n_k; ks++) {            // S and N': X[:, a]' (TB + H12[b, :]')
                        const int i2 = 4 * ks + l4;
                        const bool kin = i2 < nd;
                        const double tb = kin ? TB[i2 * 16 + l15] : 0.0, hb = (kin && bq) ? H12[bcol * hl + i2] : 0.0;
                        const double bw = tb + hb;
                        double av[GA];
#pragma unroll
                        for (int g = 0; g < GA; g++) { const int a = 16 * (at0 + g) + l15; av[g] = (kin && a < R) ? AUG[i2 * ld + nf + a] : 0.0; }
#pragma unroll
                        for (int g = 0; g < GA; g++) if (at0 + g < n_at) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bw, acc[g], 0, 0, 0);
                    }
                    if (b0 < nq) for (int kc = 0; kc < n_kc; kc++) {   // N': L[:, a]' G1[b, :]'
                        const int c = 4 * kc + l4;
                        const bool cin = c < nc;
                        const double bw = (cin && bq) ? G1[bcol * nc + c] : 0.0;
                        double av[GA];
#pragma unroll
                        for (int g = 0; g < GA; g++) { const int a = 16 * (at0 + g) + l15; av[g] = (cin && a < R) ? AUG[(nd + c) * ld + nf + a] : 0.0; }
#pragma unroll
                        for (int g = 0; g < GA; g++) if (at0 + g < n_at) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bw, acc[g], 0, 0, 0);
                    }
                    if (16 * at0 < nq) {                              // N: H12[a, :] X[:, b] + G1[a, :] L[:, b] for the tiles with q1 rows
                        for (int ks = 0; ks < n_k; ks++) {
                            const int i2 = 4 * ks + l4;
                            const bool kin = i2 < nd;
                            const double bw = (kin && bin) ? AUG[i2 * ld + nf + bcol] : 0.0;
                            double av[GA];
#pragma unroll
                            for (int g = 0; g < GA; g++) { const int a = 16 * (at0 + g) + l15; av[g] = (kin && a < nq) ? H12[a * hl + i2] : 0.0; }
#pragma unroll
                            for (int g = 0; g < GA; g++) if (16 * (at0 + g) < nq && at0 + g < n_at) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bw, acc[g], 0, 0, 0);
                        }
                        for (int kc = 0; kc < n_kc; kc++) {
                            const int c = 4 * kc + l4;
                            const bool cin = c < nc;
                            const double bw = (cin && bin) ? AUG[(nd + c) * ld + nf + bcol] : 0.0;
                            double av[GA];
#pragma unroll
                            for (int g = 0; g < GA; g++) { const int a = 16 * (at0 + g) + l15; av[g] = (cin && a < nq) ? G1[a * nc + c] : 0.0; }
#pragma unroll
                            for (int g = 0; g < GA; g++) if (16 * (at0 + g) < nq && at0 + g < n_at) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bw, acc[g], 0, 0, 0);
                        }
                    }
                    TG_D2STAMP(10);
#pragma unroll
                    for (int g = 0; g < GA; g++) {
                        if (at0 + g >= n_at) break;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int a = 16 * (at0 + g) + l4 + 4 * r;
                            if (a < R && bin) {
                                double v = acc[g][r];
                                if (a >= first_k2) {                  // unit row of a k2 variable
                                    const int kk = nd + (a - first_k2);
                                    v += TB[kk * 16 + l15] + (bq ? H12[bcol * hl + kk] : 0.0);
                                }
                                if (a < nq) {
                                    if (bcol >= first_k2) v += H12[a * hl + nd + (bcol - first_k2)];   // unit column of a k2 variable
                                    if (bq) v += H11[sym(a, bcol)];
                                }
                                A.hz[(t * R + a) * R + bcol] = ok ? v : NAN;
                                if (at0 + g < bt) A.hz[(t * R + bcol) * R + a] = ok ? v : NAN;      // the transposed tile
                            }
                        }
                    }
                    TG_D2STAMP(11);
                }
                TG_SYNC();
            }
        };
        // staging tiles of the helper waves: the prefix-sum scratch of the body terms (dead pose area), 16 nq doubles each
        const bool tb_fits = nw == 1 || (P.o_dqi + P.n_items) - P.o_tps >= (nw - 1) * 16 * nq;
        if (TEAM == 64 && 12 * P.n_items >= 16 * nq && P.o_W == P.o_J + 6 * P.n_items && tb_fits) assemble_mfma(w0 ? S + P.o_J : S + P.o_tps + (wave - 1) * 16 * nq);
        else
#endif
        if (!w0) { }
        else if (TEAM == 64 && 12 * P.n_items >= 24 * nq && P.o_W == P.o_J + 6 * P.n_items) assemble(IntTag<8>{}, S + P.o_J);
        else assemble(IntTag<4>{}, vec);
        if (n_wrenches() && nu > 0) {
            // input blocks of the point forces: D1D3fm2 = D2D3fm2 = dt/2 F_dudq (midpointvi.c:1500-1512) couple an input
            // column with the total configuration tangent of the other variable:  HZ[a][u] += sum_i (dq1_i/da + dq2_i/da) Hu[i][u]
            // and symmetrically.  Added to the finished matrix (rare path: read-modify-write of this trajectory's HZ).
#if defined(__HIP_DEVICE_COMPILE__)
            __threadfence();
            __syncthreads();
#endif
            const double *Hu = S + P.e_o_Hu;
            const int c_u = nq + nd;
            if (on && ok && w0) TG_FOR(idx, R * nu) {
                const int a = idx / nu, m = idx % nu;
                double corr = 0.0;
                for (int i = 0; i < nd; i++) corr += AUG[i * ld + nf + a] * Hu[i * nu + m];
                if (a < nq) corr += Hu[a * nu + m];
                if (a >= first_k2) corr += Hu[(nd + (a - first_k2)) * nu + m];
                double *hz = A.hz + t * (size_t)R * R;
#if defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&hz[(size_t)a * R + (c_u + m)], corr);
                atomicAdd(&hz[(size_t)(c_u + m) * R + a], corr);
#else
                hz[(size_t)a * R + (c_u + m)] += corr;
                hz[(size_t)(c_u + m) * R + a] += corr;
#endif
            }
        }
        TG_D2STAMP(3);
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if (A.prof_out && t == 0 && lane == 0 && w0) for (int i = 0; i < 12; i++) A.prof_out[i] = d2t[i];
#endif
    }

    // team-uniform convergence test (midpointvi.c:672-689)
    TG_HD bool solved(double tolerance) const {
        PROG &P = tg_fresh(this->P);
        // four partial sums: a single accumulator is a chain of nd dependent fp64 FMAs (~30 cycles each on this part)
        double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
        int i = 0;
        for (; i + 3 < P.nd; i += 4) {
            const double a = S[P.o_f + i], b = S[P.o_f + i + 1], c = S[P.o_f + i + 2], d = S[P.o_f + i + 3];
            n0 = fma(a, a, n0); n1 = fma(b, b, n1); n2 = fma(c, c, n2); n3 = fma(d, d, n3);
        }
        for (; i < P.nd; i++) n0 = fma(S[P.o_f + i], S[P.o_f + i], n0);
        const double norm = (n0 + n1) + (n2 + n3);
#if defined(__HIP_DEVICE_COMPILE__)
        if (TEAM == 64 && P.nc <= 64) {
            // one constraint per lane and a wave vote: as a loop with an early exit this was a cascade of nc basic blocks, each waiting out
            // its own LDS read
            const int c = lane < P.nc ? lane : 0;
            const bool off = lane < P.nc && fabs(S[P.o_f + P.nd + c]) > S[P.o_ctol + c];
            return !(sqrt(norm) > tolerance) && !__any(off ? 1 : 0);
        }
#endif
        if (sqrt(norm) > tolerance) return false;
        for (int c = 0; c < P.nc; c++) if (fabs(S[P.o_f + P.nd + c]) > S[P.o_ctol + c]) return false;
        return true;
    }

    // [Df | f] -> [.. | Df^-1 f]: the register-resident row-per-lane solver when the matrix fits the team
    TG_HD bool solve_kkt(bool on) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int nb4 = (P.nf + 3) >> 2;   // matrix size in blocks of 4 rows
        if constexpr (std::is_same<Real, double>::value) if (TEAM >= 4 && 4 * nb4 <= TEAM && nb4 <= 8) {
            double *Ad = S + P.o_Df;
            switch (nb4) {
            case 1: return Core<TEAM>::template gj_rows<4>(on, Ad, P.nf, P.df_ld, lane);
            case 2: return Core<TEAM>::template gj_rows<(TEAM >= 8 ? 8 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            case 3: return Core<TEAM>::template gj_rows<(TEAM >= 12 ? 12 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            case 4: return Core<TEAM>::template gj_rows<(TEAM >= 16 ? 16 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            case 5: return Core<TEAM>::template gj_rows<(TEAM >= 20 ? 20 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            case 6: return Core<TEAM>::template gj_rows<(TEAM >= 24 ? 24 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            case 7: return Core<TEAM>::template gj_rows<(TEAM >= 28 ? 28 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            default: return Core<TEAM>::template gj_rows<(TEAM >= 32 ? 32 : 4)>(on, Ad, P.nf, P.df_ld, lane);
            }
        }
#endif
        return gauss_jordan(on, S + P.o_Df, P.nf, 1, P.df_ld, S + P.o_scal);
    }

    // =====================================================================================================
    // Continuous dynamics (reference calc_dynamics, system.c:749-893): accelerations of the dynamic configs and the
    // constraint forces at a state (q, dq, u, ddq_k).  The reference factors M, forms A_proj = -Ad M^-1 Ad^T and back-
    // substitutes; here the same equations are one KKT solve
    //     [ M   -Ad^T ] [ddq_d ]   [ D                                   ]
    //     [ Ad    0   ] [lambda] = [ -(Ak ddq_k + sum_ij h_dqdq dq_i dq_j) ]
    // and the bias D = L_dq - L_ddqdq dq - M_dk ddq_k + F is summed per (body, joint) item without forming the nq^2
    // Coriolis table: with W_k = dv/dq_k and S_F = sum_k (W_k dq_k + J_k ddq_k [k kinematic]) of body F,
    //     D_i = sum_{items a of config i} ( m gam.J_a - <J_a, S_F> - <[J_a, v_F], v_F> ) + F_i
    // (the <W_a, v> parts of L_dq and L_ddqdq dq cancel).  q must be loaded in both q1 and q2.
    // =====================================================================================================
    TG_HD bool dynamics(bool on, CArgs &A, size_t t) {
        const int nq = P.nq, nd = P.nd, nk = P.nk, nc = P.nc, nf = P.nf, ld = P.df_ld;
        // The KKT matrix shares its storage with the poses (program.hpp, LDS layout): the right-hand side is
        // accumulated in f while the poses are alive, the matrix is assembled afterwards from J and the Dh items.
        Real *K = S + P.o_Df, *rhs = S + P.o_f, *ddk = S + P.o_nu + P.nu;
        if (on) {
            TG_FOR(i, nq) S[P.o_dq + i] = seeded(A.dq_in[t * nq + i], nq + i);
            TG_FOR(i, nk) ddk[i] = seeded(A.ddqk_in[t * nk + i], 2 * nq + i);
            TG_FOR(i, nf) rhs[i] = 0.0;
        }
        TG_SYNC();
        pose_sweep(on, 2);
        attach_points(on, true, true);
        spring_terms(on);
        wrench_terms(on);
        if (nc) {
            constraints(on, 2, false, S + P.o_Dh2, 0);
            if (on) {
                TG_FOR(n, P.n_dh) {
                    const int c = P.dh_pack[8 * (size_t)n], k = P.dh_pack[8 * (size_t)n + 1];
                    if (k >= nd) lds_add(&rhs[nd + c], -S[P.o_Dh2 + n] * ddk[k - nd]);
                }
                TG_FOR(pp, P.n_cpair) {
                    const int *pw = P.cpair4 + 4 * (size_t)pp;
                    const int c = pw[0], na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                    const Real h = con_d2(c, na, nb) * S[P.o_dq + ka] * S[P.o_dq + kb];
                    lds_add(&rhs[nd + c], na != nb ? -2.0 * h : -h);
                }
            }
            TG_SYNC();
        }
        jacobians(on);
        velocities(on);
        // S_F per body, in the (now dead) joint pose area
        Real *SF = S + P.o_G;
        if (on) TG_FOR(idx, 6 * P.n_bodies) {
            const int b = idx / 6, m = idx % 6;
            Real acc = 0.0;
            for (int k = P.b_item_off[b]; k < P.b_item_off[b + 1]; k++) {
                const int cfg = P.it_pack[4 * (size_t)k + 3] & 0xFFFF;
                acc += S[P.o_W + 6 * k + m] * S[P.o_dq + cfg] + (cfg >= nd ? S[P.o_J + 6 * k + m] * ddk[cfg - nd] : 0.0);
            }
            SF[idx] = acc;
        }
        TG_SYNC();
        if (on) {
            TG_FOR(it, P.n_items) {
                const int b = P.it_pack[4 * (size_t)it], cfg = P.it_pack[4 * (size_t)it + 3] & 0xFFFF;
                const Real *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b;
                const Real *J = S + P.o_J + 6 * it;
                Real jv[6];
                bracket(J, v, jv);
                const Real term = I[0] * (gam[0] * J[0] + gam[1] * J[1] + gam[2] * J[2]) - inner6(I, J, SF + 6 * b) - inner6(I, jv, v);
                if (cfg < nd) lds_add(&rhs[cfg], term);
            }
            TG_FOR(i, nd) {
                Real force = -P.damp[i] * S[P.o_dq + i];
                for (int k = 0; k < P.n_cf; k++) if (P.cf_cfg[k] == i) force += S[P.o_u + P.cf_in[k]];
                if (has_cs()) force -= cs_d1(i, S[P.o_q2 + i]);
                if (n_springs()) force -= S[P.o_sV + i];
                if (n_wrenches()) force += S[P.o_wF + i];
                if (has_damper()) force += S[P.o_sF + i];
                lds_add(&rhs[i], force);
            }
        }
        TG_SYNC();
        if (on) TG_FOR(i, nf * ld) K[i] = 0.0;
        TG_SYNC();
        if (on) {
            TG_FOR(r, nf) K[r * ld + nf] = rhs[r];
            TG_FOR(n, P.n_dh) {
                const int c = P.dh_pack[8 * (size_t)n], k = P.dh_pack[8 * (size_t)n + 1];
                if (k < nd) { K[k * ld + nd + c] = -S[P.o_Dh2 + n]; K[(nd + c) * ld + k] = S[P.o_Dh2 + n]; }
            }
            TG_FOR(pp, P.n_npairs) {   // M = [L_ddqddq] (system.c:459-489)
                const int *pw = P.pair4 + 4 * (size_t)pp;
                const int ia = pw[0], ib = pw[1], ca = pw[2] & 0xFFFF, cb = pw[2] >> 16, b = pw[3];
                const Real mab = inner6(S + P.o_I + 4 * b, S + P.o_J + 6 * ia, S + P.o_J + 6 * ib);
                lds_add(&K[ca * ld + cb], mab);
                if (ia != ib) lds_add(&K[cb * ld + ca], mab);
            }
        }
        TG_SYNC();
        const bool ok = solve_kkt(on);
        if (on && ok && A.ddq_out) TG_FOR(i, nd) A.ddq_out[t * nd + i] = tgdual::top(K[i * ld + nf]);
        if (on && ok && A.lam_out) TG_FOR(c, nc) A.lam_out[t * nc + c] = tgdual::top(K[(nd + c) * ld + nf]);
        return ok;
    }

    // Kinetic and potential energy at (q, dq) (System_total_energy / System_L, system.c:78-127): T = sum 1/2 <v_F, I v_F>,
    // V = -sum m g.p_F + config springs + two-point springs.  One lane per term, summed through an LDS atomic.
    TG_HD void energy(bool on, CArgs &A, size_t t) {
        Real *acc = S + P.o_f;   // [0] = T, [1] = V
        if (on) {
            TG_FOR(i, P.nq) S[P.o_dq + i] = seeded(A.dq_in[t * P.nq + i], P.nq + i);
            if (lane == 0) { acc[0] = 0.0; acc[1] = 0.0; }
        }
        TG_SYNC();
        pose_sweep(on, 2);
        attach_points(on, true, n_springs() > 0);
        jacobians(on);
        velocities(on);
        if (on) {
            TG_FOR(b, P.n_bodies) {
                const Real *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gb = S + P.o_gB + 12 * b;
                lds_add(&acc[0], 0.5 * inner6(I, v, v));
                lds_add(&acc[1], -I[0] * (P.grav[0] * gb[3] + P.grav[1] * gb[7] + P.grav[2] * gb[11]));
            }
            if (has_cs()) TG_FOR(i, P.nq) {
                const Real q = S[P.o_q2 + i];
                lds_add(&acc[1], 0.5 * P.cs_k[i] * q * q - P.cs_kq0[i] * q + P.cs_c0[i]);   // (a NonlinearConfigSpring's V() is 0 in the reference, :15-22)
            }
            TG_FOR(sp, n_springs()) {
                const int c = P.nc + sp;
                const Real *a = S + P.o_pE + 3 * P.c_e1[c], *b = S + P.o_pE + 3 * P.c_e2[c];
                const Real x = sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
                lds_add(&acc[1], 0.5 * P.s_k[sp] * (x - P.s_x0[sp]) * (x - P.s_x0[sp]));
            }
        }
        TG_SYNC();
        if (on && lane == 0) { A.energy_out[2 * t] = tgdual::top(acc[0]); A.energy_out[2 * t + 1] = tgdual::top(acc[1]); }
    }

    // First and second derivatives of the Lagrangian for every config / pair of configs at (q, dq) (System_L_dq ...
    // System_L_ddqddq, system.c:129-489): the per-item and per-pair quantities of the integrator, summed into the caller's
    // (zeroed) output arrays.  lag1 = [L_dq | L_ddq], lag2 = [L_dqdq | L_ddqdq (dq row, q column) | L_ddqddq].
    TG_HD void lagrangian(bool on, CArgs &A, size_t t) {
        const int nq = P.nq;
        if (on) TG_FOR(i, nq) S[P.o_dq + i] = seeded(A.dq_in[t * nq + i], nq + i);
        TG_SYNC();
        pose_sweep(on, 2);
        attach_points(on, true, n_springs() > 0);
        spring_terms(on);
        jacobians(on);
        velocities(on);
        double *o1 = A.lag1_out + t * 2 * (size_t)nq, *o2 = A.lag2_out + t * 3 * (size_t)nq * nq;
        if (on) {
            TG_FOR(it, P.n_items) {
                const int b = P.it_pack[4 * (size_t)it], cfg = P.it_pack[4 * (size_t)it + 3] & 0xFFFF;
                const Real *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b;
                const Real *J = S + P.o_J + 6 * it, *W = S + P.o_W + 6 * it;
                gl_add(&o1[cfg], tgdual::top(inner6(I, W, v) + I[0] * (gam[0] * J[0] + gam[1] * J[1] + gam[2] * J[2])));
                gl_add(&o1[nq + cfg], tgdual::top(inner6(I, J, v)));
            }
            TG_FOR(pp, P.n_pairs) {
                const int ia = P.pair_a[pp], ib = P.pair_b[pp];
                const int b = P.it_pack[4 * (size_t)ia], ca = P.it_pack[4 * (size_t)ia + 3] & 0xFFFF, cb = P.it_pack[4 * (size_t)ib + 3] & 0xFFFF;
                const Real *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b;
                const Real *Ja = S + P.o_J + 6 * ia, *Jb = S + P.o_J + 6 * ib, *Wa = S + P.o_W + 6 * ia, *Wb = S + P.o_W + 6 * ib;
                Real tb[6];
                bracket(Wa, Jb, tb);
                const Real lqq = inner6(I, tb, v) + inner6(I, Wa, Wb) +
                                   I[0] * (gam[0] * (Ja[4] * Jb[2] - Ja[5] * Jb[1]) + gam[1] * (Ja[5] * Jb[0] - Ja[3] * Jb[2]) +
                                           gam[2] * (Ja[3] * Jb[1] - Ja[4] * Jb[0]));
                const Real mab = inner6(I, Ja, Jb);
                bracket(Ja, Jb, tb);
                const Real c_ab = inner6(I, tb, v) + inner6(I, Ja, Wb), c_ba = inner6(I, Jb, Wa);
                gl_add(&o2[(size_t)ca * nq + cb], tgdual::top(lqq));
                gl_add(&o2[((size_t)nq + ca) * nq + cb], tgdual::top(c_ab));
                gl_add(&o2[((size_t)2 * nq + ca) * nq + cb], tgdual::top(mab));
                if (ia != ib) {
                    gl_add(&o2[(size_t)cb * nq + ca], tgdual::top(lqq));
                    gl_add(&o2[((size_t)nq + cb) * nq + ca], tgdual::top(c_ba));
                    gl_add(&o2[((size_t)2 * nq + cb) * nq + ca], tgdual::top(mab));
                }
            }
            if (has_cs()) TG_FOR(i, nq) {
                Real d1_, d2_, d3_;
                cs_eval(i, S[P.o_q2 + i], d1_, d2_, d3_);
                gl_add(&o1[i], tgdual::top(-d1_));
                gl_add(&o2[(size_t)i * nq + i], tgdual::top(-d2_));
            }
            if (n_springs()) TG_FOR(i, nq) gl_add(&o1[i], tgdual::top(-S[P.o_sV + i]));
            TG_FOR(pp, n_spair()) {
                const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
                const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                gl_add(&o2[(size_t)ka * nq + kb], tgdual::top(-S[P.o_sH + pp]));
                if (pw[1] != pw[2]) gl_add(&o2[(size_t)kb * nq + ka], tgdual::top(-S[P.o_sH + pp]));
            }
        }
    }

    // =====================================================================================================
    // First derivatives of the continuous dynamics (reference calc_dynamics_deriv1, system.c:912-1299): d(ddq_d, lambda)
    // / d(q, dq, ddq_k, u).  With r = D - M ddq_d + Ad^T lambda = 0 and g = A ddq + dq^T H dq = 0 solved by `dynamics`,
    // the implicit-function theorem gives, with the SAME KKT matrix, one right-hand side per derivative variable:
    //     [ M  -Ad^T ] [d ddq_d ]   [  dr/dtheta ]
    //     [ Ad   0   ] [d lambda] = [ -dg/dtheta ]      (partials at fixed ddq_d, lambda)
    // and all columns are eliminated together.  The reference's O(nq^3) tables (M_dq, L_ddqdqdq dq, ...) never appear:
    // per (body, joint) item k the derivative of the body's acceleration-like vector a_F = sum_j (W_j dq_j + J_j ddq_j),
    //     da_F/dq_k  = [PX_k, J_k] + [W_k, v - P_k]      (PX_k, P_k: prefix sums of W_j dq_j + J_j ddq_j and of J_j dq_j)
    //     da_F/ddq_k = 2 W_k + [J_k, v]
    // is O(1), and the torque tau_a = m gam.J_a - <J_a, a_F> - <[J_a, v], v> of item a is differentiated pair by pair.
    // =====================================================================================================
    TG_HD bool dyn_deriv1(bool on, CArgs &A, size_t t) {
        const int nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc, nf = P.nf, ld = P.g_ld;
        Real *AUG = S + P.g_o_AUG, *X = S + P.g_o_X, *aF = S + P.g_o_aF, *xs = S + P.g_o_x, *acc = xs + nf;
        const Real *dq = S + P.o_dq;
        const int c_q = nf, c_dq = nf + nq, c_k = nf + 2 * nq, c_u = c_k + nk;
        const bool ok0 = dynamics(on, A, t);
        const bool path_sums = nc > 0 && 12 * nc <= 6 * P.n_items;      // (the constraint path sums below live in the X area)
        if (on) TG_FOR(r, nf) xs[r] = S[P.o_Df + r * P.df_ld + nf];
        TG_SYNC();
        if (on) {
            TG_FOR(i, nq) acc[i] = i < nd ? xs[i] : S[P.o_nu + nu + (i - nd)];   // ddq of every config
            TG_FOR(i, nf * ld) AUG[i] = 0.0;
        }
        TG_SYNC();
        // the poses again (the solve above overwrote them); J, W, v, gam of `dynamics` are still valid
        pose_sweep(on, 2);
        attach_points(on, true, true);
        if (nc && on) {
            TG_FOR(n, P.n_dh) {
                const int c = P.dh_pack[8 * (size_t)n], k = P.dh_pack[8 * (size_t)n + 1];
                const Real a = S[P.o_Dh2 + n];
                if (k < nd) { AUG[k * ld + nd + c] = -a; AUG[(nd + c) * ld + k] = a; }
                else AUG[(nd + c) * ld + c_k + (k - nd)] = -a;                       // -dg/d(ddq_k)
            }
            TG_FOR(pp, P.n_cpair) {
                const int *pw = P.cpair4 + 4 * (size_t)pp;
                const int c = pw[0], na = pw[1], nb = pw[2], ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                const Real h2 = con_d2(c, na, nb), lam = xs[nd + c];
                const int row = (nd + c) * ld;
                if (ka < nd) lds_add(&AUG[ka * ld + c_q + kb], lam * h2);           // d(Ad^T lambda)/dq
                lds_add(&AUG[row + c_q + kb], -h2 * acc[ka]);                        // -d(A ddq)/dq
                lds_add(&AUG[row + c_dq + ka], -2.0 * h2 * dq[kb]);                  // -d(dq^T H dq)/d(dq)
                if (na != nb) {
                    if (kb < nd) lds_add(&AUG[kb * ld + c_q + ka], lam * h2);
                    lds_add(&AUG[row + c_q + ka], -h2 * acc[kb]);
                    lds_add(&AUG[row + c_dq + kb], -2.0 * h2 * dq[ka]);
                }
            }
            // -d(dq^T H dq)/dq_k = -sum_ij h_c,dqdqdq(i, j, k) dq_i dq_j, plane constraints (and systems whose scratch is too small for the
            // path sums below): term by term, one lane per (item k, item i), j >= i
            TG_FOR(idx, P.n_dh * P.g_max_cu) {
                const int n = idx / P.g_max_cu, c = P.dh_c[n], k = P.dh_cfg[n];
                if (path_sums && !(has_plane() && P.c_type[c] == TG_CONSTRAINT_PLANE)) continue;
                const int ni = P.cu_off[c] + idx % P.g_max_cu;
                if (ni >= P.cu_off[c + 1]) continue;
                const Real dqi = dq[P.dh_cfg[ni]];
                Real sum = 0.0;
                for (int nj = ni; nj < P.cu_off[c + 1]; nj++)
                    sum += (nj == ni ? 1.0 : 2.0) * con_d3(c, ni, nj, n) * dq[P.dh_cfg[nj]];
                lds_add(&AUG[(nd + c) * ld + c_q + k], -sum * dqi);
            }
        }
        TG_SYNC();
        if (path_sums) {
            // The same contraction for distance and point constraints in O(path length) per end point instead of O(n^3) third derivatives per
            // constraint (76 % of this kernel on the puppet, where a string depends on up to 14 configs).  For an end point p with the joints t of
            // its path ordered root-first, D_t = dp/dq_t and Om_t the world axis of joint t (0 if prismatic): d2p(x <= y) = Om_x x D_y,
            // d3p(x <= y <= z) = Om_x x (Om_y x D_z).  With the running sums along the path
            //   W_k = sum_{t<=k} Om_t dq_t,   al_k = sum_{t<=k} W_{t-1} x Om_t dq_t,   T_k = sum_{t>k} D_t dq_t,   C_k = sum_{t>k} dq_t Om_t x (D_t dq_t + 2 T_t)
            // the contracted derivatives of the point are
            //   a = sum_t D_t dq_t,   c = sum_ij d2p(i, j) dq_i dq_j = C_0,
            //   a_k = sum_i d2p(i, k) dq_i = W_k x D_k + Om_k x T_k,
            //   c_k = sum_ij d3p(i, j, k) dq_i dq_j = W_k x (W_k x D_k) + al_k x D_k + 2 W_k x (Om_k x T_k) + Om_k x C_k
            // (both i, j at or before k: Jacobi's identity turns the ordered double sum into the first two terms) and, v = p_1 - p_2 and the
            // differences of a, c, a_k, c_k over the two end points,  sum_ij h_ijk dq_i dq_j = 2 (2 a_k . a + v_k . c + v . c_k)  for a distance
            // constraint (distance.c:100-133), (c_k)_comp for a point constraint (point.c:48-54).  Pass 1: a and c per (constraint, end point);
            // pass 2: each end point's lane walks its path forwards (W, al), then backwards (W, al by subtraction) and adds its part of every item's entry.
            Real *tot = X;      // [2 nc][6]: a, c of the whole path (X is written after this phase)
            auto item = [&](int n, int E, Real *D, Real *Om, Real &rate, int &cfg) {
                const int *rec = P.dh_pack + 8 * (size_t)n;
                const int oj = rec[2], kind = (rec[3] >> 8) & 0xFF;
                cfg = rec[1];
                dpos_rec(rec[4 + E], oj, kind, D);
                if (kind >= TG_RX) { const Real *gj = S + P.o_G + oj; const int ax = kind - TG_RX; Om[0] = gj[ax]; Om[1] = gj[4 + ax]; Om[2] = gj[8 + ax]; }
                else { Om[0] = Om[1] = Om[2] = 0.0; }
                rate = dq[cfg];
            };
            if (on) TG_FOR(ce, 2 * nc) {
                const int t0 = P.cpath_off[ce], t1 = P.cpath_off[ce + 1], E = ce & 1;
                Real T[3] = {0, 0, 0}, C[3] = {0, 0, 0}, D[3], Om[3], r, x[3], y[3];
                int cfg;
                for (int t = t1 - 1; t >= t0; t--) {
                    item(P.cpath_items[t], E, D, Om, r, cfg);
                    for (int m = 0; m < 3; m++) y[m] = r * D[m] + 2.0 * T[m];
                    cross3(Om, y, x);
                    for (int m = 0; m < 3; m++) { C[m] += r * x[m]; T[m] += r * D[m]; }
                }
                Real *o = tot + 6 * ce;
                for (int m = 0; m < 3; m++) { o[m] = T[m]; o[3 + m] = C[m]; }
            }
            TG_SYNC();
            if (on) TG_FOR(ce, 2 * nc) {
                const int c = ce >> 1, E = ce & 1, t0 = P.cpath_off[ce], t1 = P.cpath_off[ce + 1];
                if (has_plane() && P.c_type[c] == TG_CONSTRAINT_PLANE) continue;
                const bool point = P.c_type[c] == TG_CONSTRAINT_POINT;
                const Real *o1 = tot + 12 * c, *o2 = o1 + 6;
                const Real *p1 = S + P.o_pE + 3 * P.c_e1[c], *p2 = S + P.o_pE + 3 * P.c_e2[c];
                Real a[3], cc[3], v[3], W[3] = {0, 0, 0}, al[3] = {0, 0, 0}, T[3] = {0, 0, 0}, C[3] = {0, 0, 0}, D[3], Om[3], r, x[3], y[3], z[3], ak[3], ck[3];
                for (int m = 0; m < 3; m++) { a[m] = o1[m] - o2[m]; cc[m] = o1[3 + m] - o2[3 + m]; v[m] = p1[m] - p2[m]; }
                const Real sg = E ? -1.0 : 1.0;
                int cfg;
                for (int t = t0; t < t1; t++) {      // W and al of the whole path
                    item(P.cpath_items[t], E, D, Om, r, cfg);
                    cross3(W, Om, x);
                    for (int m = 0; m < 3; m++) { al[m] += r * x[m]; W[m] += r * Om[m]; }
                }
                for (int t = t1 - 1; t >= t0; t--) {
                    item(P.cpath_items[t], E, D, Om, r, cfg);
                    cross3(W, D, x);                 // W_k x D_k
                    cross3(Om, T, y);                // Om_k x T_k
                    for (int m = 0; m < 3; m++) ak[m] = x[m] + y[m];
                    cross3(W, x, ck);                // W_k x (W_k x D_k)
                    cross3(al, D, z);
                    for (int m = 0; m < 3; m++) ck[m] += z[m];
                    cross3(W, y, z);
                    for (int m = 0; m < 3; m++) ck[m] += 2.0 * z[m];
                    cross3(Om, C, z);
                    for (int m = 0; m < 3; m++) ck[m] += z[m];
                    const Real h3 = point ? ck[P.c_comp[c]] : 2.0 * (2.0 * dot3(ak, a) + dot3(D, cc) + dot3(v, ck));
                    lds_add(&AUG[(nd + c) * ld + c_q + cfg], -(sg * h3));
                    for (int m = 0; m < 3; m++) z[m] = r * D[m] + 2.0 * T[m];
                    cross3(Om, z, x);
                    for (int m = 0; m < 3; m++) { C[m] += r * x[m]; T[m] += r * D[m]; W[m] -= r * Om[m]; }
                    cross3(W, Om, x);                // W_{k-1} x Om_k
                    for (int m = 0; m < 3; m++) al[m] -= r * x[m];
                }
            }
            TG_SYNC();
        }
        // One lane per body walks its path with the two running prefixes P_k = sum_{j<k} J_j dq_j and
        // PX_k = sum_{j<k} (W_j dq_j + J_j ddq_j) in registers and leaves X_k = da_F/dq_k = [PX_k, J_k] + [W_k, v - P_k]
        // (only X is stored: 6 doubles per item instead of 18 keeps the kernel at two wavefronts per CU)
        if (on) TG_FOR(b, P.n_bodies) {
            const Real *v = S + P.o_vB + 6 * b;
            Real pr[6] = {0, 0, 0, 0, 0, 0}, px[6] = {0, 0, 0, 0, 0, 0};
            for (int k = P.b_item_off[b]; k < P.b_item_off[b + 1]; k++) {
                const int cfg = P.it_pack[4 * (size_t)k + 3] & 0xFFFF;
                const Real *J = S + P.o_J + 6 * k, *W = S + P.o_W + 6 * k;
                const Real dqk = dq[cfg], ak = acc[cfg];
                Real vm[6], t1[6], t2[6];
                for (int m = 0; m < 6; m++) vm[m] = v[m] - pr[m];
                bracket(px, J, t1);
                bracket(W, vm, t2);
                for (int m = 0; m < 6; m++) {
                    X[6 * k + m] = t1[m] + t2[m];
                    pr[m] += J[m] * dqk;
                    px[m] += W[m] * dqk + J[m] * ak;
                }
            }
            for (int m = 0; m < 6; m++) aF[6 * b + m] = px[m];
        }
        TG_SYNC();
        if (on) {
            TG_FOR(pp, P.n_pairs) {
                const int x = P.pair_a[pp], y = P.pair_b[pp];   // x at or before y on the path
                const int b = P.it_pack[4 * (size_t)x];
                const Real *I = S + P.o_I + 4 * b, *v = S + P.o_vB + 6 * b, *gam = S + P.o_gam + 3 * b, *aFb = aF + 6 * b;
                const Real *Jx = S + P.o_J + 6 * x, *Jy = S + P.o_J + 6 * y;
                // gravity: d(m gam.J_a)/dq_k = m gam.(w_x x v_y), symmetric in (a, k)
                const Real g2 = I[0] * (gam[0] * (Jx[4] * Jy[2] - Jx[5] * Jy[1]) + gam[1] * (Jx[5] * Jy[0] - Jx[3] * Jy[2]) +
                                          gam[2] * (Jx[3] * Jy[1] - Jx[4] * Jy[0]));
                auto emit = [&](int a, int k, bool k_later) {
                    const int ca = P.it_pack[4 * (size_t)a + 3] & 0xFFFF, ck = P.it_pack[4 * (size_t)k + 3] & 0xFFFF;
                    if (ca >= nd) return;
                    const Real *Ja = S + P.o_J + 6 * a, *Jk = S + P.o_J + 6 * k, *Wk = S + P.o_W + 6 * k, *dAk = X + 6 * k;
                    Real br[6], jav[6], jaw[6], jkv[6];
                    bracket(Ja, Jk, br); bracket(Ja, v, jav); bracket(Ja, Wk, jaw); bracket(Jk, v, jkv);
                    Real tq = g2 - inner6(I, Ja, dAk) - inner6(I, jaw, v) - inner6(I, jav, Wk);
                    if (k_later) {   // dJ_a/dq_k = [J_a, J_k] only for joints after a
                        Real djv[6];
                        bracket(br, v, djv);
                        tq -= inner6(I, br, aFb) + inner6(I, djv, v);
                    }
                    const Real td = -(2.0 * inner6(I, Ja, Wk) + inner6(I, Ja, jkv)) - inner6(I, br, v) - inner6(I, jav, Jk);
                    const Real mak = inner6(I, Ja, Jk);
                    lds_add(&AUG[ca * ld + c_q + ck], tq);
                    lds_add(&AUG[ca * ld + c_dq + ck], td);
                    if (ck < nd) lds_add(&AUG[ca * ld + ck], mak);
                    else lds_add(&AUG[ca * ld + c_k + (ck - nd)], -mak);
                };
                emit(x, y, y != x);
                if (x != y) emit(y, x, false);
            }
            TG_FOR(i, nd) {
                lds_add(&AUG[i * ld + c_dq + i], -P.damp[i]);
                for (int k = 0; k < P.n_cf; k++) if (P.cf_cfg[k] == i) lds_add(&AUG[i * ld + c_u + P.cf_in[k]], 1.0);
                if (has_cs()) lds_add(&AUG[i * ld + c_q + i], -cs_d2(i, S[P.o_q2 + i]));
            }
            TG_FOR(pp, n_wpair()) {
                const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + n_spair() + pp);
                const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                if (ka < nd) lds_add(&AUG[ka * ld + c_q + kb], S[P.o_wH + 2 * pp]);
                if (pw[1] != pw[2] && kb < nd) lds_add(&AUG[kb * ld + c_q + ka], S[P.o_wH + 2 * pp + 1]);
            }
            TG_FOR(n, n_wdh()) {
                const int m = P.n_dh + n_sdh() + n, o = P.dh_cfg[m], w = P.dh_c[m] - (nc + n_springs());
                if (o >= nd) continue;
                for (int s6 = 0; s6 < 6; s6++) {
                    const int in = P.wr_in[6 * w + s6];
                    if (in >= 0) lds_add(&AUG[o * ld + c_u + in], S[P.o_wD + 6 * n + s6]);
                }
            }
            TG_FOR(pp, n_spair()) {
                const int *pw = P.cpair4 + 4 * (size_t)(P.n_cpair + pp);
                const int ka = pw[3] & 0xFFFF, kb = pw[3] >> 16;
                const Real h = S[P.o_sH + pp];
                if (ka < nd) lds_add(&AUG[ka * ld + c_q + kb], -h);
                if (pw[1] != pw[2] && kb < nd) lds_add(&AUG[kb * ld + c_q + ka], -h);
                if (has_damper()) {
                    Real fab, fba, fdd;
                    damper_pair(pp, fab, fba, fdd);
                    if (ka < nd) { lds_add(&AUG[ka * ld + c_q + kb], fab); lds_add(&AUG[ka * ld + c_dq + kb], fdd); }
                    if (pw[1] != pw[2] && kb < nd) { lds_add(&AUG[kb * ld + c_q + ka], fba); lds_add(&AUG[kb * ld + c_dq + ka], fdd); }
                }
            }
        }
        TG_SYNC();
        bool ok;
        const int R = P.g_nrhs;
#if defined(__HIP_DEVICE_COMPILE__)
        const int w = nf + R, nb4 = (nf + 3) >> 2;
        if constexpr (!std::is_same<Real, double>::value) ok = gauss_jordan(on, AUG, nf, R, ld, S + P.o_scal);   // (forward-mode scalars: the generic solver)
        else
        if (TEAM == 64 && nf > 16 && nf <= 31 && w <= 128 && 12 * P.n_joints >= 200) {
            // 17..31 unknowns, up to 128 columns: panels of four columns, every right-hand side in the rank-4 matrix-core update
            // (gj_panel_rhs: what the first-derivative kernel of the discrete path uses; gj_cols below spends 28 x 56 lane-wide FMAs
            // plus the pivot-column traffic per pivot step on the same system)
            Real *sc = S + P.o_G;   // the joint poses are dead during the solve
            switch (nb4) {
            case 5: ok = Core<64, SPRINGS, PROG>::template gj_panel_rhs<20, 8>(on, AUG, nf, w, ld, lane, sc); break;
            case 6: ok = Core<64, SPRINGS, PROG>::template gj_panel_rhs<24, 8>(on, AUG, nf, w, ld, lane, sc); break;
            case 7: ok = Core<64, SPRINGS, PROG>::template gj_panel_rhs<28, 8>(on, AUG, nf, w, ld, lane, sc); break;
            default: ok = Core<64, SPRINGS, PROG>::template gj_panel_rhs<32, 8>(on, AUG, nf, w, ld, lane, sc); break;
            }
        } else
        if (TEAM == 64 && w <= 128 && nb4 <= 8 && P.gjc_ok) {
            Real *sc = S + P.o_G;   // the joint poses are dead during the solve
            switch (nb4) {
            case 1: ok = Core<TEAM>::template gj_cols<4>(on, AUG, nf, w, ld, sc, lane); break;
            case 2: ok = Core<TEAM>::template gj_cols<8>(on, AUG, nf, w, ld, sc, lane); break;
            case 3: ok = Core<TEAM>::template gj_cols<12>(on, AUG, nf, w, ld, sc, lane); break;
            case 4: ok = Core<TEAM>::template gj_cols<16>(on, AUG, nf, w, ld, sc, lane); break;
            case 5: ok = Core<TEAM>::template gj_cols<20>(on, AUG, nf, w, ld, sc, lane); break;
            case 6: ok = Core<TEAM>::template gj_cols<24>(on, AUG, nf, w, ld, sc, lane); break;
            case 7: ok = Core<TEAM>::template gj_cols<28>(on, AUG, nf, w, ld, sc, lane); break;
            default: ok = Core<TEAM>::template gj_cols<32>(on, AUG, nf, w, ld, sc, lane); break;
            }
        } else
#endif
            ok = gauss_jordan(on, AUG, nf, R, ld, S + P.o_scal);
        if (on && ok0 && ok) {
            const int col0[4] = {c_q, c_dq, c_k, c_u}, rows[4] = {nq, nq, nk, nu};
            for (int g = 0; g < 8; g++) {
                double *dst = A.g1[g];
                if (!dst) continue;
                const int v = g & 3, width = g < 4 ? nd : nc, r0 = g < 4 ? 0 : nd;
                dst += t * (size_t)rows[v] * width;
                TG_FOR(idx, rows[v] * width) {
                    const int k = idx / width, o = idx % width;
                    dst[idx] = tgdual::top(AUG[(r0 + o) * ld + col0[v] + k]);
                }
            }
        }
        return ok0 && ok;
    }

    // midpoint evaluation shared by every mode: rates, poses, Jacobians, velocities, residual
    TG_HD void eval_midpoint(bool on) {
        PROG &P = tg_fresh(this->P);
        if (on) TG_FOR(i, P.nq) S[P.o_dq + i] = (S[P.o_q2 + i] - S[P.o_q1 + i]) / dt;
        TG_SYNC();
        TG_STAMP(0);
        pose_sweep(on, 0);
        TG_STAMP(1);
        attach_points(on, true, n_springs() > 0 || n_wrenches() > 0);
        spring_terms(on);
        wrench_terms(on);
        jacobians(on);
        TG_STAMP(2);
        velocities(on);
        TG_STAMP(3);
        residual_dyn(on);
        TG_STAMP(4);
    }
    TG_HD void eval_constraints(bool on, int sel, bool want_h, double *Dh, int ld = -1) {
        if (P.nc == 0) return;
        if (ld < 0) ld = 0;
        pose_sweep(on, sel);
        TG_STAMP(5);
        attach_points(on, false, true);
        constraints(on, sel, want_h, Dh, ld);
        TG_STAMP(6);
    }
};

// One trajectory of a FORWARD-MODE launch of a continuous-dynamics mode (dual.hpp): the state (q, dq, ddq_k, u) is loaded with unit
// directions on the variables RunArgs::seed1 / seed2 name, the mode's own code runs on Real = Dual<double> (one direction) or
// Dual<Dual<double>> (two), and its outputs receive the highest-order coefficient: with MODE_DYN_DERIV1 the second derivatives of the
// continuous dynamics (reference calc_dynamics_deriv2, system.c:1301-2029), one input variable per trajectory; with MODE_LAGRANGIAN the
// third- and fourth-order derivatives of the Lagrangian (System_L_dqdqdq ... System_L_ddqddqdqdq, system.c:204-622).  S: the team's LDS
// slice in units of Real (the same layout as the mode's double kernel).
template <int TEAM, int MODE, bool SPRINGS, class PROG, class ARGS, class Real>
TG_HD void run_forward(PROG &P, ARGS &A, Real *S, int lane, int traj) {
    static_assert(MODE == MODE_DYNAMICS || MODE == MODE_DYN_DERIV1 || MODE == MODE_LAGRANGIAN || MODE == MODE_ENERGY, "run_forward: continuous-dynamics modes only");
    const int nq = P.nq, nk = P.nk, nu = P.nu;
    const bool live = traj < A.batch;
    const size_t t = (size_t)(live ? traj : 0);
    Core<TEAM, SPRINGS, PROG, Real> core(P, S, lane, A.t2 - A.t1);
    core.seed1 = A.seed1 ? A.seed1[t] : -1;
    core.seed2 = A.seed2 ? A.seed2[t] : -1;
    core.init_sweep_schedule(false);
    if (live) {
        TG_FOR(i, nq) { const Real q = core.seeded(A.q2[t * nq + i], i); S[P.o_q1 + i] = q; S[P.o_q2 + i] = q; }
        TG_FOR(i, P.nd) S[P.o_p1 + i] = 0.0;
        TG_FOR(i, P.nc) S[P.o_lam + i] = 0.0;
        TG_FOR(i, nu) S[P.o_u + i] = core.seeded(A.u1[t * nu + i], 2 * nq + nk + i);
        TG_FOR(i, P.n_dh) { S[P.o_Dh1 + i] = 0.0; S[P.o_Dh2 + i] = 0.0; }
    }
    TG_SYNC();
    if constexpr (MODE == MODE_LAGRANGIAN) core.lagrangian(live, A, t);
    if constexpr (MODE == MODE_ENERGY) core.energy(live, A, t);
    if constexpr (MODE == MODE_DYN_DERIV1) {
        const bool ok = core.dyn_deriv1(live, A, t);
        if (live && lane == 0) { A.iters[t] = 0; A.status[t] = ok ? TG_OK : TG_SINGULAR; }
    }
    if constexpr (MODE == MODE_DYNAMICS) {
        const bool ok = core.dynamics(live, A, t);
        if (live && lane == 0) { A.iters[t] = 0; A.status[t] = ok ? TG_OK : TG_SINGULAR; }
    }
}

// One trajectory (team) of a launch.  `traj` may be >= batch (idle team): it still takes part in
// every TG_SYNC.
// MODE is a compile-time parameter so that every kernel mode gets its own register allocation (the
// derivative modes are far larger than the rollout loop).
// PIVOT: -1 the pivot rule is read from the arguments at run time (generic kernels); 0 / 1 compile the single-precision ranking /
// the reference's exact rule in alone (specialised kernels: the variant not asked for is not in the kernel's call graph at all)
template <int TEAM, int MODE, bool SPRINGS = false, class PROG = CProg, class ARGS = CArgs, int PIVOT = -1>
TG_HD void run_trajectory(PROG &P0, ARGS &A0, double *S, int lane, int traj, int wave = 0, int nw = 1) {
    PROG &P = tg_fresh(P0);
    ARGS &A = A0;
    const int nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc;
    const bool live = traj < A.batch;
    const size_t t = (size_t)(live ? traj : 0);
    double dt = MODE == MODE_ROLLOUT ? A.dt : (A.t2 - A.t1);
    if (A.dt_steps && A.dt_period > 0) dt = A.dt_steps[t % (size_t)A.dt_period];   // per-trajectory step size (k-parallel linearisation)
    Core<TEAM, SPRINGS, PROG> core(P, S, lane, dt);
    core.wave = wave; core.nw = nw;
    core.d1_compact = MODE == MODE_DERIV1 && P.a_ok != 0;
    if (wave == 0) core.init_sweep_schedule(MODE == MODE_ROLLOUT);
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    core.prof_last = (long long)__builtin_amdgcn_s_memtime();
    const long long prof_rt0 = (long long)__builtin_amdgcn_s_memrealtime();     // 100 MHz wall clock: total cycles / these ticks = the shader clock
#endif

    // ---- load state ----------------------------------------------------------------------------------
    if (live && wave == 0) {
        TG_FOR(i, nq) { S[P.o_q1 + i] = A.q1[t * nq + i]; S[P.o_q2 + i] = A.q2[t * nq + i]; }
        TG_FOR(i, nd) S[P.o_p1 + i] = (MODE == MODE_ROLLOUT) ? A.p2[t * nd + i] : A.p1[t * nd + i];
        TG_FOR(i, nc) S[P.o_lam + i] = A.lam[t * nc + i];
        TG_FOR(i, nu) S[P.o_u + i] = A.u1[t * nu + i];
        TG_FOR(i, P.n_dh) { S[P.o_Dh1 + i] = 0.0; S[P.o_Dh2 + i] = 0.0; }
    }
    TG_SYNC();

    if constexpr (MODE == MODE_CALC_P2) {  // MidpointVI.calc_p2: midpointvi.c:491-504
        core.eval_midpoint(live);
        if (live) TG_FOR(i, nd) A.p2[t * nd + i] = 0.5 * dt * S[P.o_Ldq + i] + S[P.o_Lddq + i];
        return;
    }
    if constexpr (MODE == MODE_DERIV1) {
        core.deriv1(live, A, t);
        return;
    }
    if constexpr (MODE == MODE_DERIV2Z) {
        core.deriv2z(live, A, t);
        return;
    }
    if constexpr (MODE == MODE_LAGRANGIAN) {
        core.lagrangian(live, A, t);
        return;
    }
    if constexpr (MODE == MODE_ENERGY) {
        core.energy(live, A, t);
        return;
    }
    if constexpr (MODE == MODE_DYN_DERIV1) {
        const bool ok = core.dyn_deriv1(live, A, t);
        if (live && lane == 0) { A.iters[t] = 0; A.status[t] = ok ? TG_OK : TG_SINGULAR; }
        return;
    }
    if constexpr (MODE == MODE_DYNAMICS) {
        const bool ok = core.dynamics(live, A, t);
        if (live && lane == 0) { A.iters[t] = 0; A.status[t] = ok ? TG_OK : TG_SINGULAR; }
        return;
    }
    if constexpr (MODE == MODE_CALC_F) {  // MidpointVI.calc_f: midpointvi.c:567-575
        core.eval_constraints(live, 1, false, S + P.o_Dh1);
        core.eval_midpoint(live);
        core.eval_constraints(live, 2, true, S + P.o_Dh2);
        if (live) TG_FOR(i, P.nf) A.f_out[t * P.nf + i] = S[P.o_f + i];
        return;
    }

    // ---- rollout ----------------------------------------------------------------------------------------
    if constexpr (MODE == MODE_ROLLOUT) {
    const int nX = P.nX;
    if (live && A.X) {  // X_0 = [q2; p2; v2] of the incoming state (dsystem.py:276-281, midpointvi.py:325-332)
        double *x = A.X + t * (size_t)(A.n_steps + 1) * nX;
        TG_FOR(i, nq) x[i] = S[P.o_q2 + i];
        TG_FOR(i, nd) x[nq + i] = S[P.o_p1 + i];
        TG_FOR(i, nk) x[nq + nd + i] = (A.t2 != A.t1) ? (S[P.o_q2 + nd + i] - S[P.o_q1 + nd + i]) / (A.t2 - A.t1) : 0.0;
    }
    if (live && A.X && A.Kproj) {  // projection: X_0 = bX_0 by definition (dsystem.py:441), including its v part
        const size_t o0 = t * (size_t)(A.n_steps + 1) * nX;
        TG_FOR(i, nX) A.X[o0 + i] = A.bX[o0 + i];
    }
    bool failed = false;
    int status = TG_OK, total_iters = 0, n_fallback = 0;
    // open-loop inputs and kinematic targets of the NEXT step, requested while this step's Newton loop runs (one value per lane when they
    // fit a team): read at the step's start they are two HBM round trips in a row on every step's critical path
    double pre_u = 0.0, pre_k = 0.0;
    bool pre_ok = false;
    for (int step = 0; step < A.n_steps; step++) {
        PROG &P = tg_fresh_step(P0);      // per step: nothing of the schedule / the arguments stays in SGPRs across steps
        ARGS &A = tg_fresh_args(A0);
        const int nq = P.nq, nd = P.nd, nk = P.nk, nu = P.nu, nc = P.nc, nX = P.nX;
        const double dt_prev = dt;             // step size of the step that produced the incoming state (feedback: v = dq_k / dt_prev)
        if (A.dt_steps && A.dt_period == 0) { dt = A.dt_steps[step]; core.dt = dt; core.inv_dt = 1.0 / dt; }   // non-uniform time base (dsystem.py:229-274 takes any t)
        const bool on = live && !failed;
        bool fb_done = false;
#if defined(__HIP_DEVICE_COMPILE__)
        if (TEAM == 64 && A.Kproj && 2 * (nu + nk) <= TEAM && nX <= 32 * (TEAM / (nu + nk) > 4 ? 4 : TEAM / (nu + nk)) && nX + 4 * (nu + nk) <= 6 * P.n_items) {
            // feedback inputs u = bU - K (x - bX) with the gain rows spread over the wavefront: `parts` lanes per row, each with its slice of
            // the row in flight at once and four partial sums; the state error is formed once (one lane per entry, coalesced reference
            // read) in the dead Jacobian area.  One lane per row walks 80 entries in ten dependent batches of gain-row loads.
            const int nU = nu + nk, parts = TEAM / nU > 4 ? 4 : TEAM / nU, chunk = (nX + parts - 1) / parts;
            double *dx = S + P.o_J, *part = dx + nX;
            const int l = tg_opaque(lane);
            const int j = l / parts, pt = l - j * parts, i0 = pt * chunk;
            const bool mine = l < parts * nU;
            const size_t grp = A.group_map ? (size_t)A.group_map[t / A.group_size] : (size_t)(t / A.group_size);
            const double *Kr = A.Kproj + ((grp * A.n_steps + step) * nU + (mine ? j : 0)) * nX;
            if (on && step > 0) {
                const double *bx = A.bX + (t * (size_t)(A.n_steps + 1) + step) * nX;
                const double dtp = dt_prev;
                TG_FOR(i, nX) {
                    double x;
                    if (i < nq) x = S[P.o_q2 + i];
                    else if (i < nq + nd) x = S[P.o_p1 + i - nq];
                    else x = dtp != 0.0 ? (S[P.o_q2 + nd + (i - nq - nd)] - S[P.o_q1 + nd + (i - nq - nd)]) / dtp : 0.0;
                    dx[i] = x - bx[i];
                }
            }
            TG_SYNC();
            if (on && step > 0 && mine) {
                double kv[32];
#pragma unroll
                for (int u = 0; u < 32; u++) if (u < chunk) kv[u] = Kr[i0 + u < nX ? i0 + u : 0];
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int u = 0; u < 32; u += 4) {
                    if (u < chunk) a0 = fma(kv[u], i0 + u < nX ? dx[i0 + u < nX ? i0 + u : 0] : 0.0, a0);
                    if (u + 1 < chunk) a1 = fma(kv[u + 1], i0 + u + 1 < nX ? dx[i0 + u + 1 < nX ? i0 + u + 1 : 0] : 0.0, a1);
                    if (u + 2 < chunk) a2 = fma(kv[u + 2], i0 + u + 2 < nX ? dx[i0 + u + 2 < nX ? i0 + u + 2 : 0] : 0.0, a2);
                    if (u + 3 < chunk) a3 = fma(kv[u + 3], i0 + u + 3 < nX ? dx[i0 + u + 3 < nX ? i0 + u + 3 : 0] : 0.0, a3);
                }
                part[pt * nU + j] = (a0 + a1) + (a2 + a3);
            }
            TG_SYNC();
            if (on && l < nU) {
                double acc = A.bU[(t * A.n_steps + step) * nU + l];
                if (step > 0) for (int q = 0; q < parts; q++) acc -= part[q * nU + l];   // X_0 = bX_0 by definition of the projection: no correction at k = 0
                S[P.o_nu + l] = acc;
                if (A.Uout) A.Uout[(t * A.n_steps + step) * nU + l] = acc;
            }
            TG_SYNC();
            fb_done = true;
        }
#endif
        if (A.Kproj && !fb_done) {  // feedback inputs from the state entering this step (before the shift: v needs q1)
            const int nU = nu + nk;
            if (on) TG_FOR(j, nU) {
                const size_t grp = A.group_map ? (size_t)A.group_map[t / A.group_size] : (size_t)(t / A.group_size);
                const double *Kr = A.Kproj + ((grp * A.n_steps + step) * nU + j) * nX;
                const double *bx = A.bX + (t * (size_t)(A.n_steps + 1) + step) * nX;
                const double dtp = step == 0 ? (A.t2 - A.t1) : dt_prev;
                double acc = A.bU[(t * A.n_steps + step) * nU + j];
                if (step > 0)   // X_0 = bX_0 by definition of the projection: no correction at k = 0
                for (int i = 0; i < nq; i++) acc -= Kr[i] * (S[P.o_q2 + i] - bx[i]);
                if (step > 0)
                for (int i = 0; i < nd; i++) acc -= Kr[nq + i] * (S[P.o_p1 + i] - bx[nq + i]);
                if (step > 0)
                for (int i = 0; i < nk; i++) {
                    const double v = dtp != 0.0 ? (S[P.o_q2 + nd + i] - S[P.o_q1 + nd + i]) / dtp : 0.0;
                    acc -= Kr[nq + nd + i] * (v - bx[nq + nd + i]);
                }
                S[P.o_nu + j] = acc;
                if (A.Uout) A.Uout[(t * A.n_steps + step) * nU + j] = acc;
            }
            TG_SYNC();
        }
        // advance: q1 <- q2, (p1 already holds p2), inputs, kinematic targets, hints (midpointvi.py:188-197)
        if (on) {
            if (A.predictor) {   // opt-in warm start: constant-velocity extrapolation of the dynamic configs
                TG_FOR(i, nq) {
                    const double prev = S[P.o_q1 + i], cur = S[P.o_q2 + i];
                    S[P.o_q1 + i] = cur;
                    if (i < nd) S[P.o_q2 + i] = 2.0 * cur - prev;
                }
            } else {
                TG_FOR(i, nq) S[P.o_q1 + i] = S[P.o_q2 + i];
            }
            if (pre_ok) { if (lane < nu) S[P.o_u + lane] = pre_u; }
            else TG_FOR(i, nu) S[P.o_u + i] = A.Kproj ? S[P.o_nu + i] : A.U[(t * A.n_steps + step) * nu + i];
            // the momentum entering the last step is the state's p1 afterwards (midpointvi.py:189)
            if (step == A.n_steps - 1) TG_FOR(i, nd) A.p1[t * nd + i] = S[P.o_p1 + i];
        }
        TG_SYNC();
        if (on) {
            if (pre_ok) { if (lane < nk) S[P.o_q2 + nd + lane] = pre_k; }
            else TG_FOR(i, nk) S[P.o_q2 + nd + i] = A.Kproj ? S[P.o_nu + nu + i] : A.K[(t * A.n_steps + step) * nk + i];
            if (A.q2_hint && step == 0) TG_FOR(i, nd) S[P.o_q2 + i] = A.q2_hint[t * nd + i];
            if (A.lam_hint && step == 0) TG_FOR(i, nc) S[P.o_lam + i] = A.lam_hint[t * nc + i];
        }
        TG_SYNC();
        // Dh at q1, held fixed during the solve (midpointvi.c:705-707).  After the first step it is the
        // Dh2 of the previous step's converged q2 (same point, same inputs), so only step 0 sweeps.
        // rates dq = (q2 - q1) / dt of the first evaluation: here (no reader before the barriers below) and, for the later evaluations, in the
        // Newton update itself -- a phase of its own at the head of every evaluation otherwise (eval_both_tab)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_DUAL_SWEEP)
        const bool fuse_rates = core.dual_ok() && P.tab_ok;
#else
        const bool fuse_rates = false;
#endif
        core.rates_ready = fuse_rates;
        if (fuse_rates && on) TG_FOR(i, nq) S[P.o_dq + i] = core.over_dt(S[P.o_q2 + i] - S[P.o_q1 + i]);
        if (step == 0) core.eval_constraints(on, 1, false, S + P.o_Dh1);
        else if (nc) {
            if (on) TG_FOR(i, P.n_dh) S[P.o_Dh1 + i] = S[P.o_Dh2 + i];
            TG_SYNC();
        }
        int iterations = 0;
        bool done = !on;
        pre_ok = false;
        if (TEAM == 64 && on && !A.Kproj && nu <= TEAM && nk <= TEAM && step + 1 < A.n_steps) {
            if (lane < nu) pre_u = A.U[(t * A.n_steps + step + 1) * nu + lane];
            if (lane < nk) pre_k = A.K[(t * A.n_steps + step + 1) * nk + lane];
            pre_ok = true;
        }
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        { long long t_ = (long long)__builtin_amdgcn_s_memtime(); core.prof[9] += t_ - core.prof_last; core.prof_last = t_; }
#endif
        for (;;) {
            PROG &P = tg_fresh(P0);
            const int nd = P.nd, nc = P.nc;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_DUAL_SWEEP)
            if constexpr (TEAM == 64 && !SPRINGS && tg_static_wev<typename std::remove_cv<PROG>::type>::value) { core.wev_on = true; core.pk_image = PIVOT == 0; core.eval_world(!done); }
            else if (core.dual_ok()) { if (P.tab_ok) core.eval_both_tab(!done); else core.eval_both(!done); }
            else
#endif
            {
                core.eval_midpoint(!done);
                core.eval_constraints(!done, 2, true, S + P.o_Dh2);
            }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_DUAL_SWEEP)
            if (core.dual_ok() && P.tab_ok) { if (!done && core.solved_fused(A.tolerance)) done = true; }
            else
#endif
            if (!done && core.solved(A.tolerance)) done = true;
#if defined(TG_MOCK_TIMING)
            // TIMING MOCK (tools/mock_third_wave.sh; never loadable by the package): the same instruction stream on numbers that may be
            // garbage -- exactly three Newton iterations per step whatever the residual says, the structured solve's guards ignored, the
            // iterate frozen (the update's stores go to a dead word).  Only for builds whose LDS areas are deliberately aliased.
            done = live && !failed ? iterations >= 3 : true;
#endif
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            { long long t_ = (long long)__builtin_amdgcn_s_memtime(); core.prof[12] += t_ - core.prof_last; core.prof_last = t_; }
#endif
            if (!done && iterations > A.max_iterations) { done = true; failed = true; status = TG_NOT_CONVERGED; }
#if defined(__HIP_DEVICE_COMPILE__)
            // the workgroup is one wavefront: a wave vote replaces the LDS-based block-wide AND
            if (TEAM == 64 ? done : (__all(done ? 1 : 0) != 0)) break;
#else
            if (done) break;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_BBD)
            // the structured solve's table rows of this lane, requested ahead of the matrix assembly
            BbdRows<tg_static_bbd_cols<TEAM == 64 && PIVOT == 0, typename std::remove_cv<PROG>::type>::value> bbd_tab_rows;
            if constexpr (TEAM == 64 && PIVOT == 0 && tg_static_bbd<typename std::remove_cv<PROG>::type>::value)
                bbd_tab_rows = bbd_rows<tg_static_bbd_cols<true, typename std::remove_cv<PROG>::type>::value>((const int *)(S + P.o_bbd), lane);
#endif
            core.newton_matrix(!done);
            bool ok;
#if defined(__HIP_DEVICE_COMPILE__)
            const int nb4 = (P.nf + 3) >> 2;   // matrix size in blocks of 4 rows
            bool bbd_done = false, bbd_updated = false;
#if !defined(TG_NO_BBD)
            if constexpr (TEAM == 64 && PIVOT == 0 && tg_static_bbd<typename std::remove_cv<PROG>::type>::value) {
                // structured solve along the system's bordered-block-diagonal plan (bbd.hpp): no pivot search; a failed pivot guard
                // leaves the image untouched and the pivoting solver below takes over (a full-wave team: done is false here and
                // the branch is uniform).  Scratch: the Jacobian columns, dead between the matrix's assembly and the next evaluation.
                typedef typename std::remove_cv<PROG>::type SP;
                __builtin_amdgcn_s_setprio(TG_CHAIN_PRIO);
                // the solve's scratch: the Jacobian columns (dead between the matrix's assembly and the next evaluation) -- or, with the packed
                // image, the per-body entries behind the q2 poses: the twists and per-config vectors in the J / W area then survive the solve, and
                // a failed guard can re-assemble the dense image for the pivoting solver from them
                constexpr bool PKI = tg_static_pk<SP>::value;
                typedef typename std::conditional<PKI, BbdPackedImage<SP::bbd_pk_nr, SP::bbd_pk_nc2, SP::bbd_pk_tb, SP::bbd_pk_tc2, SP::bbd_pk_xs>, BbdDenseImage>::type Img;
                double *bscr = PKI ? S + P.o_W + 12 * P.n_joints : S + P.o_J;
#if !defined(TG_BBD_FUSED_UPDATE) || defined(TG_MOCK_TIMING)
                bbd_done = gj_bbd<SP::nf, SP::df_ld, SP::bbd_ng, SP::bbd_nb, SP::bbd_t, BbdNoUpdate, Img>(S + P.o_Df, bbd_tab_rows, bscr, lane, P.bbd_tvar);
#else
                // -DTG_BBD_FUSED_UPDATE (measured: 30.03 against 30.01 ms, i.e. nothing, for 2 spilled registers; off by default): the Newton update
                // rides on the solve's last stage in the kernels whose update also forms the rates (implied by the world-frame evaluation's conditions)
                typedef BbdUpd<SP::nd, SP::o_q2 - SP::o_Df, SP::o_q1 - SP::o_Df, SP::o_dq - SP::o_Df, SP::o_lam - SP::o_Df> Upd;
                if constexpr (tg_static_wev<SP>::value) {
                    bbd_done = gj_bbd<SP::nf, SP::df_ld, SP::bbd_ng, SP::bbd_nb, SP::bbd_t, Upd, Img>(S + P.o_Df, bbd_tab_rows, bscr, lane, P.bbd_tvar, core.dt, core.inv_dt);
                    bbd_updated = bbd_done;
                } else bbd_done = gj_bbd<SP::nf, SP::df_ld, SP::bbd_ng, SP::bbd_nb, SP::bbd_t, BbdNoUpdate, Img>(S + P.o_Df, bbd_tab_rows, bscr, lane, P.bbd_tvar);
#endif
                if (!bbd_done) n_fallback++;
                if constexpr (PKI) {
                    // guard failed (rare): the pivoting solver wants the dense image -- the packed one (untouched by the failed solve) unpacked
                    // through the plan's map; the two overlap, so every lane first reads its share, then writes it
                    if (!bbd_done) {
                        constexpr int NE = SP::nf * (SP::nf + 1), PER = (NE + TEAM - 1) / TEAM;
                        double v[PER];
#pragma unroll
                        for (int u = 0; u < PER; u++) {
                            const int e = lane + u * TEAM, at = P.bbd_map[e < NE ? e : 0];
                            const double x = S[P.o_Df + (at >= 0 ? at : 0)];
                            v[u] = at >= 0 ? x : 0.0;
                        }
                        TG_SYNC();
                        if constexpr (SP::df_ld > SP::nf + 1) { TG_FOR(i, SP::nf) S[P.o_Df + i * SP::df_ld + SP::nf + 1] = 0.0; }      // (the padding column of an odd row stride)
#pragma unroll
                        for (int u = 0; u < PER; u++) {
                            const int e = lane + u * TEAM;
                            if (e < NE) S[P.o_Df + (e / (SP::nf + 1)) * SP::df_ld + e % (SP::nf + 1)] = v[u];
                        }
                        TG_SYNC();
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                ok = true;
#if defined(TG_MOCK_TIMING)
                bbd_done = true;
#endif
            }
#endif
            if (bbd_done) { }
            else if (TEAM >= 4 && 4 * nb4 <= TEAM && nb4 <= 8) {
                double *Ad = S + P.o_Df;
                if (PIVOT < 0 ? A.exact_pivot != 0 : PIVOT == 1) {
                switch (nb4) {
                case 1: ok = Core<TEAM>::template gj_rows_exact<4>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 2: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 8 ? 8 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 3: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 12 ? 12 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 4: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 16 ? 16 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 5: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 20 ? 20 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 6: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 24 ? 24 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 7: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 28 ? 28 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                default: ok = Core<TEAM>::template gj_rows_exact<(TEAM >= 32 ? 32 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                }
                } else
#if defined(TG_GJ_PANEL_DEFAULT) && !defined(TG_NO_GJ_PANEL)
                // full-wave teams, 17..31 unknowns: panels of four columns + matrix-core trailing update (scratch: the Jacobian
                // columns, dead between the Newton matrix's assembly and the next evaluation).  System-specialised kernels only
                // (spec_kernel.hip defines TG_GJ_PANEL_DEFAULT): with run-time sizes the panel code carries guards and address
                // arithmetic that make the generic rollout kernel slower (88.5 vs 85.5 ms per benchmark launch), so it keeps gj_rows
                if (TEAM == 64 && nb4 >= 5 && P.nf <= 31 && 12 * P.n_items >= 128) {
                    double *scr = S + P.o_J;
                    switch (nb4) {
                    case 5: ok = Core<64>::template gj_panel<20>(!done, Ad, P.nf, P.df_ld, lane, scr); break;
                    case 6: ok = Core<64>::template gj_panel<24>(!done, Ad, P.nf, P.df_ld, lane, scr); break;
                    case 7: ok = Core<64>::template gj_panel<28>(!done, Ad, P.nf, P.df_ld, lane, scr); break;
                    default: ok = Core<64>::template gj_panel<32>(!done, Ad, P.nf, P.df_ld, lane, scr); break;
                    }
                } else
#endif
                switch (nb4) {
                case 1: ok = Core<TEAM>::template gj_rows<4>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 2: ok = Core<TEAM>::template gj_rows<(TEAM >= 8 ? 8 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 3: ok = Core<TEAM>::template gj_rows<(TEAM >= 12 ? 12 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 4: ok = Core<TEAM>::template gj_rows<(TEAM >= 16 ? 16 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 5: ok = Core<TEAM>::template gj_rows<(TEAM >= 20 ? 20 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 6: ok = Core<TEAM>::template gj_rows<(TEAM >= 24 ? 24 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                case 7: ok = Core<TEAM>::template gj_rows<(TEAM >= 28 ? 28 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                default: ok = Core<TEAM>::template gj_rows<(TEAM >= 32 ? 32 : 4)>(!done, Ad, P.nf, P.df_ld, lane); break;
                }
            } else
#endif
                ok = core.gauss_jordan(!done, S + P.o_Df, P.nf, 1, P.df_ld, S + P.o_scal);
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            { long long t_ = (long long)__builtin_amdgcn_s_memtime(); core.prof[14] += t_ - core.prof_last; core.prof_last = t_; }
#endif
#if !defined(TG_MOCK_TIMING)
            if (!done && !ok) { done = true; failed = true; status = TG_SINGULAR; }
#endif
#if defined(TG_MOCK_TIMING)
            if (!done) {      // (the same loads, operations and stores; the stores hit dead words of the scale / closed-loop input areas)
                TG_FOR(i, nd) { const double v = S[P.o_q2 + i] - S[P.o_Df + i * P.df_ld + P.nf]; S[P.o_scal + (i & 15)] = v; S[P.o_nu + (i & 15)] = core.over_dt(v - S[P.o_q1 + i]); }
                TG_FOR(c, nc) S[P.o_scal + 16 + c] = S[P.o_lam + c] - S[P.o_Df + (nd + c) * P.df_ld + P.nf];
                iterations++;
            }
#else
            if (!done) {
#if defined(__HIP_DEVICE_COMPILE__)
                if (bbd_updated) { }      // (the structured solve applied the update itself)
                else
#endif
                {
                // the solution: the image's right-hand-side column, or the packed image's solution vector (the structured solve's own order)
                auto sol = [&](int i) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TG_NO_BBD)
                    if constexpr (TEAM == 64 && PIVOT == 0 && tg_static_pk<typename std::remove_cv<PROG>::type>::value) {
                        typedef typename std::remove_cv<PROG>::type SP;
                        if (bbd_done) return S[P.o_Df + SP::bbd_pk_xs + i];
                    }
#endif
                    return S[P.o_Df + i * P.df_ld + P.nf];
                };
                if (fuse_rates) TG_FOR(i, nd) { const double v = S[P.o_q2 + i] - sol(i); S[P.o_q2 + i] = v; S[P.o_dq + i] = core.over_dt(v - S[P.o_q1 + i]); }
                else
                TG_FOR(i, nd) S[P.o_q2 + i] -= sol(i);
                TG_FOR(c, nc) S[P.o_lam + c] -= sol(nd + c);
                }
                iterations++;
            }
#endif
            TG_SYNC();
        }
        if (on && !failed) {
            total_iters += iterations;
            // p2 = D2L2 at the converged midpoint (midpointvi.c:742-743); it becomes p1 of the next step
#if defined(TG_MOCK_TIMING)
            TG_FOR(i, nd) S[P.o_scal + (i & 15)] = 0.5 * dt * S[P.o_Ldq + i] + S[P.o_Lddq + i];
#else
            TG_FOR(i, nd) S[P.o_p1 + i] = 0.5 * dt * S[P.o_Ldq + i] + S[P.o_Lddq + i];
#endif
        }
        TG_SYNC();
        if (on && !failed && A.X) {
            double *x = A.X + (t * (size_t)(A.n_steps + 1) + step + 1) * nX;
            TG_FOR(i, nq) x[i] = S[P.o_q2 + i];
            TG_FOR(i, nd) x[nq + i] = S[P.o_p1 + i];
            TG_FOR(i, nk) x[nq + nd + i] = (S[P.o_q2 + nd + i] - S[P.o_q1 + nd + i]) / dt;
        }
    }
    // ---- write back q1, q2, p2, lambda1, u1 (p1 was stored when the last step started) ----------------------
#if defined(TG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (A.prof_out && traj == TG_PROF_TRAJ && lane == 0) {
        core.prof[13] = (long long)__builtin_amdgcn_s_memrealtime() - prof_rt0;    // (rollout: slot 13 carries the wall-clock ticks instead of the few tail cycles)
        for (int i = 0; i < 16; i++) A.prof_out[i] = core.prof[i];
    }
#endif
    if (live) {
        TG_FOR(i, nq) { A.q1[t * nq + i] = S[P.o_q1 + i]; A.q2[t * nq + i] = S[P.o_q2 + i]; }
        TG_FOR(i, nd) A.p2[t * nd + i] = S[P.o_p1 + i];
        TG_FOR(i, nc) A.lam[t * nc + i] = S[P.o_lam + i];
        TG_FOR(i, nu) A.u1[t * nu + i] = S[P.o_u + i];
        if (lane == 0) { A.iters[t] = total_iters; A.status[t] = status; if (A.fallbacks) A.fallbacks[t] = n_fallback; }
        if (A.mirror) {
            const size_t Bn = (size_t)A.batch;
            double *m = A.mirror;
            TG_FOR(i, nq) m[t * nq + i] = S[P.o_q2 + i];
            TG_FOR(i, nd) m[Bn * nq + t * nd + i] = S[P.o_p1 + i];
            TG_FOR(i, nc) m[Bn * (nq + nd) + t * nc + i] = S[P.o_lam + i];
            if (lane == 0) { int *o = reinterpret_cast<int *>(m + Bn * (nq + nd + nc)); o[t] = total_iters; o[Bn + t] = status; }
        }
    }
    }
}

}  // namespace tg
