// bbd_solve.hpp -- device part of the structured Newton solve (plan and rationale: bbd.hpp).  Included by mvi_core.hpp after
// tg_rcp; device pass only.
#pragma once
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
namespace tg {

// diagnostic builds (tools/micro/bbd_bench.hip -DTG_BBD_STAMPS): cycle stamps between the stages of gj_bbd
#if defined(TG_BBD_STAMPS)      // (the including file defines  __device__ long long tg::tg_bbd_stamps[8])
#define BBD_STAMP(i) do { long long t_ = (long long)__builtin_amdgcn_s_memtime(); if (lane == 0 && blockIdx.x == 0) tg_bbd_stamps[i] += t_ - bbd_t0; bbd_t0 = t_; } while (0)
#else
#define BBD_STAMP(i) ((void)0)
#endif

// lane k of every 16-lane row broadcast to the row (DPP row_newbcast; K is a compile-time constant)
template <int K>
__device__ __forceinline__ double bbd_bcast(double v) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + K, 0xf, 0xf, true));
}
// a += l * (lane K's b)
template <int K>
__device__ __forceinline__ void bbd_fmac(double &a, double b, double l) {
#if defined(TG_BBD_ASM)
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(l), "n"(K));
#else
    a = fma(l, bbd_bcast<K>(b), a);
#endif
}

// One Gauss-Jordan step on column K of a row-per-lane block held in registers a[0 .. NCOL): every lane of a 16-lane row except
// lane K subtracts its multiple of lane K's row.  Columns < K are already eliminated (only the pivot lanes hold them) and column
// K itself is not touched (the pivot lane keeps its pivot, the others' entries are dead).  `first` .. `last`: the live columns.
// myrp: 1 / pivot of the lane's own row, kept by the pivot lane of each step (every lane computes the reciprocal).  The pivot guard is
// evaluated from it after the last step -- |pivot| > guard  <=>  |1 / pivot| guard < 1 -- instead of a compare and two mask operations
// in every step.
template <int K, int NCOL>
__device__ __forceinline__ void bbd_step(double (&a)[NCOL], int r, double &myrp) {
    const double p = bbd_bcast<K>(a[K]);
    const double rp = tg_rcp(p);
    const bool is_piv = r == K;
    myrp = is_piv ? rp : myrp;
    const double l = is_piv ? 0.0 : -a[K] * rp;
#pragma unroll
    for (int j = K + 1; j < NCOL; j++) bbd_fmac<K>(a[j], a[j], l);
}
template <int K, int KEND, int NCOL>
__device__ __forceinline__ void bbd_steps(double (&a)[NCOL], int r, double &myrp) {
    if constexpr (K < KEND) {
        bbd_step<K, NCOL>(a, r, myrp);
        bbd_steps<K + 1, KEND, NCOL>(a, r, myrp);
    }
}

// Solve the dense image A [NF][LD] (right-hand side in column NF) along the plan whose rows for this lane are `rows` (bbd_rows).
// scratch: T * (T + 1) + T doubles of LDS outside the image.  Returns true and leaves x_i in A[i][NF] if every guard held;
// otherwise returns false with the image untouched.
// The lane's rows of the plan tables (BbdPlan::tab staged in LDS): constants of the lane, so a caller can request them long before the
// solve (the rollout does, ahead of the matrix assembly) and the solve starts with the row loads instead of a table round trip.
template <int NGB> struct BbdRows { int wl; int wc[NGB]; };
template <int NGB>
__device__ __forceinline__ BbdRows<NGB> bbd_rows(const int *tab_generic, int lane) {
    typedef __attribute__((address_space(3))) const int lds_int;
    lds_int *tab = (lds_int *)tab_generic;
    BbdRows<NGB> t;
    t.wl = tab[lane];
#pragma unroll
    for (int j = 0; j < NGB; j++) t.wc[j] = tab[64 + 16 * (lane >> 4) + j];
    return t;
}

// The Newton update applied by the solver's own lanes (rollout kernels): the lane that forms x_i also forms q2_i - x_i and the rate
// (q2_i - q1_i) / dt, or lambda_c - x_{nd + c} -- as a phase of its own the update is an LDS round trip and a barrier behind the solve.
// nd = 0: no update (the solutions only go to the image's right-hand-side column, where every caller can read them).
// The LDS offsets of the operands relative to the image are compile-time (BbdUpd: q2, q1, dq, lambda1 minus the image's offset; ND = 0: no
// update -- the solutions only go to the image's right-hand-side column, where every caller can read them); only dt and 1 / dt travel.
#if defined(TG_BBD_INLINE)      // (A/B switch: the structured solve inlined into its caller instead of an out-of-line call with its own register allocation)
#define TG_BBD_ATTR __forceinline__
#else
#define TG_BBD_ATTR __noinline__
#endif
template <int ND_, int Q2_, int Q1_, int DQ_, int LAM_> struct BbdUpd { static constexpr int nd = ND_, q2 = Q2_, q1 = Q1_, dq = DQ_, lam = LAM_; };
typedef BbdUpd<0, 0, 0, 0, 0> BbdNoUpdate;
// Where the solver finds its rows: the dense image [NF][LD] (gathered by the plan tables), or the image in its own order (bbd.hpp,
// BbdPacked): row r of group g at (g * NR + r) * NC2, trailing row i at TB + i * TC2, solutions to XS + variable
struct BbdDenseImage { static constexpr bool packed = false; static constexpr int nr = 0, nc2 = 0, tb = 0, tc2 = 0, xs = 0; };
template <int NR_, int NC2_, int TB_, int TC2_, int XS_> struct BbdPackedImage { static constexpr bool packed = true; static constexpr int nr = NR_, nc2 = NC2_, tb = TB_, tc2 = TC2_, xs = XS_; };
template <int NF, int LD, int NG, int NB, int T, class UP = BbdNoUpdate, class IMG = BbdDenseImage, class TVAR = const int *>
__device__ TG_BBD_ATTR bool gj_bbd(double *A_generic, BbdRows<NG + NB> rows, double *scratch_generic, int lane, TVAR tvar, double up_dt = 0.0, double up_inv_dt = 0.0) {
    typedef __attribute__((address_space(3))) double lds_double;
    lds_double *A = (lds_double *)A_generic, *U = (lds_double *)scratch_generic, *XT = U + T * (T + 1);
    constexpr int NCOL = NG + NB + 1, UL = T + 1;
#if defined(TG_BBD_STAMPS)
    long long bbd_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    constexpr double GUARD = 9.5367431640625e-07;   // 2^-20
    const int g = lane >> 4, r = lane & 15;
    const int wl = rows.wl;
    const int row = (wl & 0xFF) - 1, trow = ((wl >> 8) & 0xFF) - 1, timg = ((wl >> 16) & 0xFF) - 1;
    int wc[NG + NB];
#pragma unroll
    for (int j = 0; j < NG + NB; j++) wc[j] = rows.wc[j];
    // ---- stage 0: registers.  own rows: everything; border rows: the own columns only (the rest accumulates the Schur update)
    // operands of the fused update, requested with the rows (the variable of this lane: an own row's, or -- lanes < T -- trailing variable `lane`)
    // (a lane can hold two variables: an own row of its group, always a config, and -- lanes < T -- trailing variable `lane`)
    const bool ut_cfg = lane < T && timg < UP::nd, ut_lam = lane < T && timg >= UP::nd, uo = r < NG && row >= 0;
    double ut_q2 = 0.0, ut_q1 = 0.0, ut_lam_v = 0.0, uo_q2 = 0.0, uo_q1 = 0.0;
    if constexpr (UP::nd > 0) {
        const int ci = ut_cfg ? timg : 0, li = ut_lam ? timg - UP::nd : 0, oi = uo ? row : 0;
        ut_q2 = A[UP::q2 + ci]; ut_q1 = A[UP::q1 + ci]; ut_lam_v = A[UP::lam + li];
        uo_q2 = A[UP::q2 + oi]; uo_q1 = A[UP::q1 + oi];
    }
    double a[NCOL];
    const bool own = r < NG, have = row >= 0;
    const int ro = (have ? row : 0) * LD;
    double amax = 0.0;
    double tr[T + 1];
    const bool tl = lane < T;
    const int to = (tl ? timg : 0) * LD;
    double tmax = 0.0;
    if constexpr (IMG::packed) {
        // the lane's row is one run of doubles (16-byte loads); a lane past the group's rows re-reads row 0 (finite numbers nobody uses), the
        // padding rows' identity entries and the border rows' zero columns are in the image
        typedef double bbd_d2 __attribute__((ext_vector_type(2)));
        static_assert(NCOL <= IMG::nc2 && T + 1 <= IMG::tc2 && (IMG::nc2 & 1) == 0 && (IMG::tc2 & 1) == 0, "gj_bbd: packed row strides");
        const __attribute__((address_space(3))) bbd_d2 *pr = (const __attribute__((address_space(3))) bbd_d2 *)(A + (g * IMG::nr + (r < IMG::nr ? r : 0)) * IMG::nc2);
#pragma unroll
        for (int j = 0; j < NCOL; j += 2) {
            const bbd_d2 v = pr[j >> 1];
            a[j] = v.x;
            if (j + 1 < NCOL) a[j + 1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < NG + NB; j++) amax = fmax(amax, fabs(a[j]));
        const __attribute__((address_space(3))) bbd_d2 *pt = (const __attribute__((address_space(3))) bbd_d2 *)(A + IMG::tb + (lane < T ? lane : T) * IMG::tc2);      // (lanes past the system: the zero row)
#pragma unroll
        for (int j = 0; j <= T; j += 2) {
            const bbd_d2 v = pt[j >> 1];
            tr[j] = v.x;
            if (j + 1 <= T) tr[j + 1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < T; j++) tmax = fmax(tmax, fabs(tr[j]));
    } else {
#pragma unroll
    for (int j = 0; j < NCOL; j++) {
        const int col = j < NG + NB ? (wc[j] & 0xFF) - 1 : NF;
        const bool ld_ = have && col >= 0 && (own || j < NG);
        const double v = A[ro + (col >= 0 ? col : 0)];
        a[j] = ld_ ? v : ((own && !have && j == r) ? 1.0 : 0.0);
        if (j < NG + NB) amax = fmax(amax, fabs(a[j]));
    }
    // the trailing system: lane i < T of the first row holds row tvar[i]
#pragma unroll
    for (int j = 0; j <= T; j++) {
        const double v = A[to + (j < T ? tvar[j] : NF)];
        tr[j] = tl ? v : ((lane < 16 && j == lane) ? 1.0 : 0.0);
        if (j < T) tmax = fmax(tmax, fabs(tr[j]));
    }
    }
    for (int e = lane; e < T * UL; e += 64) U[e] = 0.0;
    BBD_STAMP(0);
    // ---- stage 1: the groups' own columns
    double myrp = 0.0;
    bbd_steps<0, NG, NCOL>(a, r, myrp);
    // a lane that never was a pivot lane keeps 1 / pivot = 0 and passes; a zero pivot gives inf (or NaN further down): fails
    bool bad = !(fabs(myrp) * (GUARD * amax) < 1.0);
    BBD_STAMP(1);
    // ---- stage 2: Schur updates of the border rows into U, then the trailing system
    if (trow >= 0) {
#pragma unroll
        for (int j = NG; j < NCOL; j++) {
            const int tc = j < NG + NB ? ((wc[j] >> 8) & 0xFF) - 1 : T;
            if (tc >= 0) __hip_atomic_fetch_add((double *)(U + trow * UL + tc), a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    asm volatile("" ::: "memory");
    BBD_STAMP(2);
    double trp = 0.0;
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j <= T; j++) { const double u = U[(tl ? lane : 0) * UL + j]; tr[j] += tl ? u : 0.0; }
        BBD_STAMP(3);
        bbd_steps<0, T, T + 1>(tr, r, trp);
        BBD_STAMP(4);
        bad = bad || !(fabs(trp) * (GUARD * tmax) < 1.0);
        if (tl) XT[lane] = tr[T] * trp;
    }
#if !defined(TG_MOCK_TIMING)      // (timing mock, mvi_core.hpp: the guards are evaluated and ignored)
    if (__any(bad ? 1 : 0)) return false;
#else
    asm volatile("" :: "v"(bad ? 1 : 0));
#endif
    asm volatile("" ::: "memory");
    // ---- stage 3: solutions.  trailing variables straight, own variables by back-substitution from the border's
    const double xt = tr[T] * trp;
    if (tl) A[IMG::packed ? IMG::xs + timg : to + NF] = xt;
    double xo = 0.0;
    if (own && have) {
        double s = a[NG + NB];
#pragma unroll
        for (int j = NG; j < NG + NB; j++) {
            const int tc = ((wc[j] >> 8) & 0xFF) - 1;
            s = fma(-a[j], XT[tc >= 0 ? tc : 0], s);     // (a border column that does not exist holds zeros)
        }
        xo = s * myrp;
        A[IMG::packed ? IMG::xs + row : ro + NF] = xo;
    }
    if constexpr (UP::nd > 0) {
        auto rate = [&](double v, double q1v) { const double d = v - q1v, q = d * up_inv_dt; return fma(fma(-q, up_dt, d), up_inv_dt, q); };   // Core::over_dt
        if (ut_cfg) { const double v = ut_q2 - xt; A[UP::q2 + timg] = v; A[UP::dq + timg] = rate(v, ut_q1); }
        if (ut_lam) A[UP::lam + timg - UP::nd] = ut_lam_v - xt;
        if (uo) { const double v = uo_q2 - xo; A[UP::q2 + row] = v; A[UP::dq + row] = rate(v, uo_q1); }
    }
    __syncthreads();
    BBD_STAMP(5);
    return true;
}

}  // namespace tg
#endif
