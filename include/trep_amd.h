/*
 * trep_amd.h -- C ABI of libtrepamd.so, the MI355X (gfx950) batched MidpointVI engine.
 *
 * This is the drop-in boundary for the MidpointVI hot path of MurpheyLab/trep.
 * It replaces what the reference reaches through the CPython type
 * `_trep._MidpointVI` (reference trep/_trep/midpointvi.c:2756-2921: _solve_DEL,
 * calc_p2, _calc_f, _calc_deriv1, _calc_deriv2 and the numpy work arrays they
 * mutate) and through the exported C-API slot `MidpointVI_solve_DEL`
 * (trep/_trep/c_api.h:88,586-590,693; trep/_trep/trep.h:781-783), for a BATCH of
 * independent trajectories of one mechanical system.
 *
 * Plain C, plain pointers and sizes; no Python, numpy or torch types.  All
 * floating point is IEEE fp64.  Host arrays are caller-owned and row-major
 * [batch][n]; device buffers are owned by the library.  Handles are not
 * thread-safe; one HIP stream per batch.
 */
#ifndef TREP_AMD_H
#define TREP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frame transform kinds (reference trep/_trep/trep.h:160-167 TREP_WORLD..TREP_CONST_SE3). */
enum { TG_WORLD = 0, TG_TX = 1, TG_TY = 2, TG_TZ = 3, TG_RX = 4, TG_RY = 5, TG_RZ = 6, TG_CONST_SE3 = 7 };
/* Constraint kinds (reference constraints/distance.c, constraints/point.c). */
enum { TG_CONSTRAINT_DISTANCE = 0, TG_CONSTRAINT_POINT = 1, TG_CONSTRAINT_PLANE = 2 };

/*
 * Flattened mechanical system: the integer topology tables the reference keeps in
 * Frame_s / Config_s / System_s (trep/_trep/trep.h:105-273, 541-633), as computed by
 * System._structure_changed (trep/system.py:672-771) and Frame._structure_changed
 * (trep/frame.py:658-691), plus the parameters of the supported potential / force /
 * constraint types.  All index arrays are int32; "none" is -1.
 */
typedef struct tg_system_desc {
    int32_t n_frames, n_configs, n_dyn, n_kin, n_inputs, n_constraints, n_masses;
    int32_t n_gravity, n_damping, n_config_forces;
    /* frames in System.frames order (depth-first, parent before child) */
    const int32_t *frame_transform;   /* [n_frames]  TG_WORLD..TG_CONST_SE3 */
    const int32_t *frame_parent;      /* [n_frames]  -1 for the world frame */
    const int32_t *frame_config;      /* [n_frames]  driving config index, -1 if fixed */
    const double  *frame_value;       /* [n_frames]  constant parameter of a fixed TX..RZ frame */
    const double  *frame_lg;          /* [n_frames*12] row-major 3x4 constant transform (CONST_SE3) */
    const double  *frame_inertia;     /* [n_frames*4] mass, Ixx, Iyy, Izz */
    const int32_t *frame_cache_size;  /* [n_frames]  number of configs on the path world->frame */
    const int32_t *frame_cache_index; /* [n_frames*(n_configs+1)] path configs, padded with -1 */
    /* configs: the n_dyn dynamic ones first, then the n_kin kinematic ones */
    const int32_t *config_kinematic;  /* [n_configs] 0/1 */
    const int32_t *config_k_index;    /* [n_configs] index among kinematic configs, -1 if dynamic */
    const int32_t *config_gen;        /* [n_configs] position in every dependent frame's path; n_configs if none */
    const int32_t *config_masses_off; /* [n_configs+1] CSR offsets into config_masses */
    const int32_t *config_masses;     /* frame indices of the massive frames depending on each config */
    const int32_t *masses;            /* [n_masses] frame indices with non-zero inertia */
    /* potentials / forces */
    const double  *gravity;           /* [n_gravity*3] */
    const double  *damping;           /* [n_damping*n_dyn] */
    const int32_t *config_force_config; /* [n_config_forces] config index */
    const int32_t *config_force_input;  /* [n_config_forces] input index */
    /* holonomic constraints */
    const int32_t *constraint_type;      /* [n_constraints] */
    const int32_t *constraint_frame1;    /* [n_constraints] */
    const int32_t *constraint_frame2;    /* [n_constraints] */
    const int32_t *constraint_config;    /* [n_constraints] distance: kinematic length config or -1 */
    const int32_t *constraint_component; /* [n_constraints] point: 0,1,2 */
    const double  *constraint_distance;  /* [n_constraints] distance: constant length if no config */
    const double  *constraint_tolerance; /* [n_constraints] */
    /* potentials on single configs: V = 1/2 k (q - q0)^2 (potentials/configspring.c:15-45) */
    int32_t n_config_springs;
    const int32_t *config_spring_config; /* [n_config_springs] config index */
    const double  *config_spring_k;      /* [n_config_springs] */
    const double  *config_spring_q0;     /* [n_config_springs] */
    /* springs between the origins of two frames: V = 1/2 k (|p1 - p2| - x0)^2 (potentials/linearspring.c:16-78).
     * Like the reference, which defines V, V_dq and V_dqdq only, systems with such springs have no second
     * derivatives of the step map (tg_batch_deriv2_* return TG_ERR_UNSUPPORTED). */
    int32_t n_linear_springs;
    const int32_t *linear_spring_frame1; /* [n_linear_springs] */
    const int32_t *linear_spring_frame2; /* [n_linear_springs] */
    const double  *linear_spring_k;      /* [n_linear_springs] */
    const double  *linear_spring_x0;     /* [n_linear_springs] */
    /* TG_CONSTRAINT_PLANE: h = (R(frame1) n) . (p(frame1) - p(frame2)), frame1 = plane frame, frame2 = point frame
     * (constraints/plane.c:13-26) */
    const double  *constraint_normal;    /* [n_constraints*3] plane normal n in plane-frame coordinates (others: zeros) */
    /* forces/hybridwrench.c: a wrench applied at the origin of a frame, force and torque both given by their world-frame
     * components; each of the six components (fx, fy, fz, tx, ty, tz) is an input or a constant. */
    int32_t n_hybrid_wrenches;
    const int32_t *hybrid_wrench_frame;  /* [n_hybrid_wrenches] */
    const int32_t *hybrid_wrench_input;  /* [n_hybrid_wrenches*6] input index of each component, or -1 for a constant */
    const double  *hybrid_wrench_const;  /* [n_hybrid_wrenches*6] the constant components */
    const int32_t *hybrid_wrench_kind;   /* [n_hybrid_wrenches] 0: HybridWrench; 1: SpatialWrench (forces/spatialwrench.c: all six
                                          * coefficients from g_dq g^-1, the joint's spatial twist); 2: BodyWrench
                                          * (forces/bodywrench.c: from g^-1 g_dq, the body twist) */
    /* forces/lineardamper.c: f = -c (d/dt |p1 - p2|) d|p1 - p2|/dq between the origins of two frames. */
    int32_t n_linear_dampers;
    const int32_t *linear_damper_frame1; /* [n_linear_dampers] */
    const int32_t *linear_damper_frame2; /* [n_linear_dampers] */
    const double  *linear_damper_c;      /* [n_linear_dampers] */
    /* potentials/nonlinear_config_spring.c:24-61: dV/dq = -y(m q + b) on one config, y a piecewise-quintic Spline
     * (trep/spline.py:6-259, evaluation _trep/spline.c:8-57).  Spring i uses the table rows first[i] .. first[i+1]-1, one row per
     * polynomial piece in x order: (left knot, a, b, c, d, e, f), y = a t^5 + b t^4 + c t^3 + d t^2 + e t + f with t = x - left knot;
     * the first / last piece also serve x below / above the knots. */
    int32_t n_nonlinear_springs;
    const int32_t *nonlinear_spring_config; /* [n_nonlinear_springs] config index */
    const double  *nonlinear_spring_m;      /* [n_nonlinear_springs] */
    const double  *nonlinear_spring_b;      /* [n_nonlinear_springs] */
    const int32_t *nonlinear_spring_first;  /* [n_nonlinear_springs + 1] CSR offsets into the rows of nonlinear_spring_pieces */
    const double  *nonlinear_spring_pieces; /* [rows * 7] */
} tg_system_desc;

/* Per-trajectory status written by every solve (reference: ConvergenceError / ValueError("singular")
 * raised from MidpointVI_solve_DEL, midpointvi.c:715-718, math-code.c:393-398). */
enum { TG_OK = 0, TG_NOT_CONVERGED = 1, TG_SINGULAR = 2 };

/* Library-level error codes returned by the functions below (0 = success). */
enum { TG_SUCCESS = 0, TG_ERR_INVALID = -1, TG_ERR_HIP = -2, TG_ERR_UNSUPPORTED = -3, TG_ERR_STATE = -4 };

/* Batch state fields for tg_batch_set / tg_batch_get (row-major [batch][width]). */
enum {
    TG_F_Q1 = 0,      /* width nq */
    TG_F_Q2 = 1,      /* width nq */
    TG_F_P1 = 2,      /* width nd */
    TG_F_P2 = 3,      /* width nd */
    TG_F_U1 = 4,      /* width nu */
    TG_F_LAMBDA1 = 5, /* width nc */
    /* first derivatives (valid after tg_batch_deriv1); [deriv var][output] like the reference
     * (trep.h:425-437): */
    TG_F_Q2_DQ1 = 10, /* nq x nd */
    TG_F_Q2_DP1 = 11, /* nd x nd */
    TG_F_Q2_DU1 = 12, /* nu x nd */
    TG_F_Q2_DK2 = 13, /* nk x nd */
    TG_F_P2_DQ1 = 14, TG_F_P2_DP1 = 15, TG_F_P2_DU1 = 16, TG_F_P2_DK2 = 17,
    TG_F_L1_DQ1 = 18, /* nq x nc */
    TG_F_L1_DP1 = 19, TG_F_L1_DU1 = 20, TG_F_L1_DK2 = 21
};

typedef struct tg_system tg_system;
typedef struct tg_batch tg_batch;

const char *tg_version(void);
const char *tg_last_error(void);
/* Number of visible HIP devices (<=0: none; the library then refuses to create batches). */
int tg_device_count(void);
/* out[0..3] = compute units, LDS bytes a workgroup may use (opt-in maximum), wavefront size, 0 */
int tg_device_info(int32_t device, int32_t out[4]);

/* Compile the flattened system into the device schedule.  The descriptor is copied. */
tg_system *tg_system_create(const tg_system_desc *desc);
void tg_system_destroy(tg_system *sys);
/* sizes: out[0..5] = nq, nd, nk, nu, nc, nX(=nq+nd+nk) */
int tg_system_sizes(const tg_system *sys, int32_t out[6]);

/* Introspection: out[0..7] = team size (lanes per trajectory), LDS bytes per trajectory, joints,
 * levels, bodies, (body,config) items, (item,item) pairs, constraint-Jacobian items. */
int tg_system_info(const tg_system *sys, int32_t out[8]);

/* A batch of `batch` independent trajectories of `sys` resident on HIP device `device`. */
tg_batch *tg_batch_create(tg_system *sys, int32_t batch, int32_t device);
void tg_batch_destroy(tg_batch *b);
/* Newton tolerance on |DEL| (reference MidpointVI.tolerance, default 1e-10). */
int tg_batch_set_tolerance(tg_batch *b, double tolerance);
/* Integrator times, shared by the whole batch (reference MidpointVI.t1/.t2). */
int tg_batch_set_times(tg_batch *b, double t1, double t2);
int tg_batch_get_times(const tg_batch *b, double *t1, double *t2);
/* Copy a state field host->device / device->host. */
int tg_batch_set(tg_batch *b, int32_t field, const double *host);
int tg_batch_get(tg_batch *b, int32_t field, double *host);
int tg_batch_field_width(const tg_batch *b, int32_t field);

/* p2 = D2 L_d(q1, q2) at the current (q1,q2,t1,t2): reference MidpointVI.calc_p2
 * (midpointvi.c:491-504 via :2691-2700). */
int tg_batch_calc_p2(tg_batch *b);
/* DEL residual f (width nd+nc) at the current state: reference MidpointVI.calc_f
 * (midpointvi.c:567-575). */
int tg_batch_calc_f(tg_batch *b, double *f_host);

/*
 * One MidpointVI.step for every trajectory (reference trep/midpointvi.py:174-201 +
 * MidpointVI_solve_DEL, midpointvi.c:691-747): q1<-q2, p1<-p2, t1<-t2, t2<-t2_new,
 * u1<-u1_host, q2[nd:]<-k2_host, optional q2 / lambda1 hints, Newton solve, p2.
 * u1_host [batch][nu], k2_host [batch][nk] (NULL allowed when the width is 0);
 * q2_hint_host [batch][nd] or NULL; lambda_hint_host [batch][nc] or NULL.
 * iterations_out / status_out: [batch] or NULL.  Returns TG_SUCCESS even when some
 * trajectories fail: inspect status_out.
 */
int tg_batch_step(tg_batch *b, double t2_new, const double *u1_host, const double *k2_host,
                  const double *q2_hint_host, const double *lambda_hint_host,
                  int32_t max_iterations, int32_t *iterations_out, int32_t *status_out);

/*
 * Device-resident rollout: n_steps consecutive steps of size dt in ONE kernel launch,
 * starting from the batch's current (t2, q2, p2, lambda1).  U_dev [batch][n_steps][nu]
 * and K_dev [batch][n_steps][nk] are DEVICE pointers (NULL when the width is 0);
 * X_dev, if not NULL, is a DEVICE buffer [batch][n_steps+1][nX] receiving the DSystem-style
 * state X_k = [q2 ; p2 ; v2] (reference trep/discopt/dsystem.py:11-63) for k = 0..n_steps.
 * iterations are accumulated per trajectory (tg_batch_rollout_stats).  Asynchronous on the
 * batch's stream; tg_batch_synchronize waits.
 */
int tg_batch_rollout(tg_batch *b, int32_t n_steps, double dt, const double *U_dev,
                     const double *K_dev, double *X_dev, int32_t max_iterations);
/*
 * Closed-loop rollout (the projection operator of trep.discopt: DSystem.project, dsystem.py:426-451, and
 * DOptimizer.armijo_simulate, doptimizer.py:405-428): the inputs are computed in the kernel,
 *   U_k = bU_k - Kproj_k (X_k - bX_k),   X_k = [q2; p2; v2],   U_k = [u1; k2],
 * Kproj_dev [groups][n_steps][nU][nX] with one gain schedule per `group_size` consecutive trajectories
 * (e.g. all Armijo candidates of one seed), bX_dev [batch][n_steps+1][nX], bU_dev [batch][n_steps][nU].
 * X_dev [batch][n_steps+1][nX] and U_dev [batch][n_steps][nU] (either may be NULL) receive the projected
 * trajectory.  Starts from the batch's current (t2, q2, p2, lambda1) like tg_batch_rollout.
 */
int tg_batch_rollout_closed_loop(tg_batch *b, int32_t n_steps, double dt, const double *Kproj_dev, int32_t group_size,
                                 const double *bX_dev, const double *bU_dev, double *X_dev, double *U_dev,
                                 int32_t max_iterations);
int tg_batch_rollout_stats(tg_batch *b, int64_t *total_iterations, int32_t *n_failed);
int tg_batch_status(tg_batch *b, int32_t *iterations_out, int32_t *status_out);
/* Per trajectory, for the last rollout / step launch: how many of its Newton systems (reference: one LU_decomp per Newton iteration,
 * midpointvi.c:720-733) the structured solve of a specialised kernel handed to the pivoting solver because a pivot guard failed.
 * Results are the same either way; a batch that reports fallbacks for most systems (time steps <= 1e-3, very heavy bodies) would run
 * faster with tg_batch_set_pivot_rule(b, 1) or without a specialised kernel.  Zeros from kernels without a structured solve. */
int tg_batch_solver_fallbacks(tg_batch *b, int32_t *fallbacks_out);

/* Device-side snapshot / restore of the whole integrator state (q1,q2,p1,p2,lambda1,u1,t1,t2):
 * asynchronous device-to-device copies on the batch's stream.  Lets a caller replay rollouts from
 * the same state (line-search candidates, benchmarks) without touching the host. */
int tg_batch_snapshot(tg_batch *b);
int tg_batch_restore(tg_batch *b);

/* First derivatives of the last solved step (reference MidpointVI_calc_deriv1,
 * midpointvi.c:1100-1120); results are read with tg_batch_get(TG_F_Q2_DQ1 ...). */
int tg_batch_deriv1(tg_batch *b);

/*
 * Second derivatives of the last solved step, contracted with z over the output index
 * (reference MidpointVI_calc_deriv2, midpointvi.c:2516-2545, as consumed by DSystem.fdxdx / fdxdu /
 * fdudu, trep/discopt/dsystem.py:320-386):
 *   hz[b][A][B] = sum_o z[b][o] * q2_dAdB[A][B][o] + z[b][nq+o] * p2_dAdB[A][B][o]
 * with the derivative variables ordered A,B in (q1[nq], p1[nd], u1[nu], k2[nk]); R = nq+nd+nu+nk.
 * z_host [batch][nX] (only the Qd and p parts are read, like the reference), hz_host [batch][R][R].
 * The full [A][B][out] tensors are never materialised.
 */
int tg_batch_deriv2_contract(tg_batch *b, const double *z_host, double *hz_host);
/* Same with an additional weight vector on the multiplier outputs (either vector may be NULL = zeros):
 *   hz[b][A][B] += sum_c zlambda[b][c] * l1_dAdB[A][B][c]      (the reference's _l1_dAdB tensors, trep.h:439-473;
 * in the implicit-function solve this is a seed on the constraint rows of the adjoint vector). */
int tg_batch_deriv2_contract_lambda(tg_batch *b, const double *z_host, const double *zlambda_host, double *hz_host);

/* Device memory helpers so a host language without a HIP binding can stage inputs. */
void *tg_device_alloc(int32_t device, uint64_t bytes);
int tg_device_free(int32_t device, void *ptr);
int tg_memcpy_h2d(int32_t device, void *dst_dev, const void *src_host, uint64_t bytes);
int tg_memcpy_d2h(int32_t device, void *dst_host, const void *src_dev, uint64_t bytes);

int tg_batch_synchronize(tg_batch *b);
/* Use an externally created hipStream_t (e.g. torch's current stream); NULL = own stream. */
int tg_batch_set_stream(tg_batch *b, void *hip_stream);
/* Non-uniform time base.  The reference's DSystem takes an arbitrary time vector (trep/discopt/dsystem.py:229-274: every
 * set / step uses t[k+1] - t[k]).  A list of `count` step sizes on the batch replaces the scalar dt of the entry points
 * below it: by_trajectory = 0: step k of a rollout / closed-loop rollout uses dt[k] (n_steps <= count); by_trajectory = 1:
 * trajectory t of a one-step batch (tg_batch_step, tg_batch_set_from_trajectories and the derivative kernels that follow:
 * batch = seeds x horizon) uses dt[t % count].  count = 0 removes the list. */
int tg_batch_set_step_sizes(tg_batch *b, int32_t count, const double *dt_host, int32_t by_trajectory);

/* Pivot rule of the Newton-system solve (replaces LU_decomp + LU_solve_vec, math-code.c:337-461, as used by
 * MidpointVI_solve_DEL, midpointvi.c:720-733).  Both settings are Gauss-Jordan with implicit row scaling and partial
 * pivoting over the rows not used yet, singular if the scaled pivot is <= 1e-20:
 *   exact = 0 (default): candidates are ranked in single precision (one 32-bit wave max per step); candidates within
 *       2^-17 relative of each other are taken in row order.  Fast; the solution agrees with the reference's to rounding.
 *   exact = 1: the reference's rule bit for bit -- fp64 comparison, ties to the first row of its (swapped) row order,
 *       exact singular test -- i.e. the same pivot ROW for every column, also where candidates differ by one ulp
 *       (every row's largest entry scales to 1 +- 1 ulp, so that happens in most puppet solves).  About 9 % slower.
 * Full-wave teams with 17..31 unknowns (the puppet: 28) run the default rule as gj_panel -- panels of four columns eliminated one
 * row per lane, the trailing matrix kept in the accumulator layout of v_mfma_f64_16x16x4_f64 and updated on the matrix cores --
 * which ranks the candidates exactly as above and differs in rounding only (block update instead of four rank-1 updates).
 * tg_debug_solve (test hook) solves [A | b] (row-major [n][n+1], n <= 32) with either rule (exact = 0 / 1; exact = 2: the default
 * rule through gj_panel, 16 < n < 32) and reports which original row served as pivot of each column; status TG_OK or TG_SINGULAR. */
int tg_batch_set_pivot_rule(tg_batch *b, int32_t exact);
int tg_debug_solve(int32_t device, int32_t n, int32_t exact, const double *A_aug_host, double *x_host, int32_t *pivot_rows_host, int32_t *status_host);
/* Structured Newton solve (csrc/bbd.hpp; system-specialised rollout kernels, default pivot rule).  The reference factors the dense
 * matrix with a pivot search per column (midpointvi.c:720-733 -> math-code.c:337-461).  For a tree mechanism the matrix
 * [[Df11, -Dh1^T], [Dh2, 0]] is bordered block diagonal -- the limbs couple only through the trunk's configs and the constraints --
 * and Df11 = -M/dt + O(dt) needs no pivot search: the schedule compiler derives the blocks from the matrix's structural pattern,
 * the kernel eliminates the blocks side by side on 16-lane DPP rows, then the dense border system (configs first, constraints
 * last), and back-substitutes.  Every pivot is guarded (|pivot| > 2^-20 of its row's largest entry); a failed guard hands the
 * untouched matrix to the pivoting solver above, so singular systems are reported exactly as before.
 * tg_batch_debug_newton_solve (test hook) runs that solve -- as compiled into the batch's loaded specialised library -- on
 * n_mats caller-supplied systems [nf][nf + 1]; path[m] = 1 structured, 2 pivoting solver (guard failed, no plan, or
 * skip_structured != 0), -1 singular. */
/* tg_system_newton_plan (host only): out[0..7] = plan found, groups, largest own block, largest border list, trailing size, nf, nd, 0;
 * pattern (optional, [nf * nf]): structural non-zeros of the Newton matrix (symmetrised); tab (optional, [128]): packed plan tables. */
int tg_system_newton_plan(const tg_system *sys, int32_t out[8], uint8_t *pattern, int32_t *tab);
int tg_batch_debug_newton_solve(tg_batch *b, int32_t n_mats, int32_t skip_structured, const double *A_aug_host, double *x_host, int32_t *path_host);

/* System-specialised rollout kernel.  The reference interprets the frame tree at run time; the generic kernels here
 * interpret a flat schedule.  For long rollouts of one system the schedule can instead be compiled into the kernel:
 * tg_system_spec_header returns a generated C++ header (every size, count and LDS offset a constant, the index tables
 * constant arrays); trep_amd/specialize.py compiles csrc/spec_kernel.hip against it with hipcc and
 * tg_batch_load_specialized makes the rollouts of a batch use that kernel (same template source as the generic
 * kernel: identical Newton iteration counts, states equal up to the compiler's FMA contraction choices, <= 1e-12
 * relative).  Returns the text length incl. terminator (or -1).
 * tg_system_spec_key is the 64-bit FNV-1a hash of that text; a specialised library exports the key of the header it
 * was compiled against (tg_spec_key) and tg_batch_load_specialized refuses a library whose key or sizes differ.
 * tg_batch_info: out[0] bit m = kernel mode m has a specialised kernel loaded; out[1] / out[2] bit m = a mode-m launch
 * went through a specialised / generic kernel since the batch was created; out[3] / out[4] = number of such launches;
 * out[5] pivot rule; out[6] team size; out[7] wavefronts per trajectory in the loaded library's derivative kernels (1, or 2 with
 * helper waves).  (Modes: 0 rollout/step, 1 calc_p2, 2 calc_f, 3 deriv1, 4 deriv2z, 5.. dynamics.) */
int64_t tg_system_spec_header(const tg_system *sys, char *buf, uint64_t capacity);
uint64_t tg_system_spec_key(const tg_system *sys);
int tg_batch_load_specialized(tg_batch *b, const char *library_path);
int tg_batch_info(const tg_batch *b, int32_t out[8]);
/* The HIP stream (hipStream_t) the batch launches on, for ordering foreign work after it (tg_comm_wait_stream). */
void *tg_batch_stream(tg_batch *b);

/* HIP-event timing of the kernels launched on the batch's stream since the last reset (opt-in: the first call switches
 * the per-launch events on and returns zeros; launches before it are not timed):
 * number of launches and the sum of their durations in milliseconds. */
int tg_batch_timing(tg_batch *b, int32_t reset, int32_t *n_launches, double *total_ms);

/* ------------------------------------------------------------------------------------------------------
 * Device-side discopt primitives: the direct caller of the MidpointVI path.
 * Everything below takes DEVICE pointers (tg_device_alloc) and is asynchronous; tg_device_synchronize or any
 * tg_memcpy_* waits.  "Seeds" are independent optimisation problems of the same system and horizon N:
 *   X [S][N+1][nX], U [S][N][nU], A [S][N][nX][nX], B [S][N][nX][nU], K [S][N][nU][nX] ...
 * `select_dev` (optional, int32 [n_problems]) restricts a call to a subset of the seeds.
 * ------------------------------------------------------------------------------------------------------ */

/* ---- continuous dynamics -------------------------------------------------------------
 * Replaces calc_dynamics (system.c:749-893) behind System.f() / System.lambda_() (system.py:951-959, 1018-1024),
 * for every trajectory of the batch at once: accelerations of the dynamic configs and constraint forces at the state
 * (q [B][nq], dq [B][nq], u [B][nu], ddq of the kinematic configs [B][nk]).  Outputs ddq [B][nd], lambda [B][nc],
 * status [B] (TG_OK or TG_SINGULAR; may be NULL).  Arrays of zero width may be NULL.  The integrator state of the
 * batch (q1, q2, p, lambda1, caches) is not touched.  The _device variant takes device pointers, launches on the
 * batch's stream and does not synchronise. */
int tg_batch_dynamics(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host,
                      const double *ddqk_host, double *ddq_host, double *lambda_host, int32_t *status_host);
int tg_batch_dynamics_device(tg_batch *b, const double *q_dev, const double *dq_dev, const double *u_dev,
                             const double *ddqk_dev, double *ddq_dev, double *lambda_dev, int32_t *status_dev);
/* First derivatives of the same map: calc_dynamics_deriv1 (system.c:912-1299) behind System.f_dq(), f_ddq(), f_dddk(),
 * f_du(), lambda_dq(), lambda_ddq(), lambda_dddk(), lambda_du() (system.py:961-980, 1026-1044).  Arrays are laid out
 * like the reference's internal ones, derivative variable first: f_dq, f_ddq [B][nq][nd]; f_dddk [B][nk][nd];
 * f_du [B][nu][nd]; lambda_* the same with nc outputs.  Any output may be NULL (not wanted).  The _device variant takes
 * the eight device pointers in that order. */
int tg_batch_dynamics_deriv1(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host,
                             const double *ddqk_host, double *f_dq, double *f_ddq, double *f_dddk, double *f_du,
                             double *lambda_dq, double *lambda_ddq, double *lambda_dddk, double *lambda_du,
                             int32_t *status_host);
int tg_batch_dynamics_deriv1_device(tg_batch *b, const double *q_dev, const double *dq_dev, const double *u_dev,
                                    const double *ddqk_dev, double *const out_dev[8], int32_t *status_dev);
/* First and second derivatives of the Lagrangian of every state of the batch, for every config / pair of configs
 * (System_L_dq, L_ddq, L_dqdq, L_ddqdq, L_ddqddq, system.c:129-489 behind System.L_dq() ... System.L_ddqddq(),
 * system.py:852-925).  first [B][2][nq] = (L_dq, L_ddq); second [B][3][nq][nq] = (L_dqdq, L_ddqdq with the velocity
 * config as the row and the configuration config as the column, L_ddqddq). */
int tg_batch_lagrangian(tg_batch *b, const double *q_host, const double *dq_host, double *first_host, double *second_host);
/* Forward-mode (dual-number) runs of the two calls above: every elementary operation of the analytic kernel carries its exact
 * derivative along one input variable per trajectory (csrc/dual.hpp) -- no step size, no truncation error.  seed [B]: the
 * variable, numbered q [nq] | dq [nq] | ddq_k [nk] | u [nu] (-1: none, the outputs are then zero).  The outputs have the layout
 * of the plain call and hold the DERIVATIVE of each entry along that variable:
 *   tg_batch_dynamics_deriv1_forward -- the second derivatives of the continuous dynamics: replaces calc_dynamics_deriv2
 *     (system.c:1301-2029; M_dqdq, D_dqdq ... lambda_dudu, f_dudu) behind System.f_dqdq() ... lambda_dudu() (system.py:982-1078):
 *     d f_dq / d q_j = f_dqdq[., j], d f_ddq / d q_j = f_ddqdq[., j], d f_ddq / d dq_j = f_ddqddq[., j], ... one variable per trajectory;
 *   tg_batch_lagrangian_forward -- the third derivatives of the Lagrangian (one seed: System_L_dqdqdq, L_ddqdqdq, L_ddqddqdq,
 *     system.c:204-268, 336-393, 491-530) and the fourth (seed2 not NULL: L_ddqdqdqdq, L_ddqddqdqdq, system.c:395-457, 532-622),
 *     behind System.L_dqdqdq() ... L_ddqddqdqdq() (system.py:869-949).
 * One wavefront per trajectory whatever the system's team size; TG_ERR_UNSUPPORTED if the dual-number slice exceeds the LDS. */
int tg_batch_dynamics_deriv1_forward(tg_batch *b, const double *q_host, const double *dq_host, const double *u_host,
                                     const double *ddqk_host, const int32_t *seed_host, double *f_dq, double *f_ddq,
                                     double *f_dddk, double *f_du, double *lambda_dq, double *lambda_ddq,
                                     double *lambda_dddk, double *lambda_du, int32_t *status_host);
int tg_batch_lagrangian_forward(tg_batch *b, const double *q_host, const double *dq_host, const int32_t *seed1_host,
                                const int32_t *seed2_host, double *first_host, double *second_host);

/* Initial guess of the Newton iteration in the device-resident rollouts.  0 (default): the reference's, q2 <- the previous
 * q2 (midpointvi.py:188-197) -- iteration counts then match the reference's.  1: constant-velocity extrapolation
 * q2 + (q2 - q1); same root to within the solver tolerance, about one Newton iteration fewer per step.  An opt-in
 * that departs from the reference's iteration-by-iteration behaviour; the headline benchmark uses 0. */
int tg_batch_set_predictor(tg_batch *b, int32_t mode);

/* Kinetic and potential energy of every state of the batch: energy[b] = {T, V} with T = sum over the massive frames of
 * 1/2 <v_b, I v_b> and V the sum of the potentials, so that System.L() = T - V and System.total_energy() = T + V
 * (System_L / System_total_energy, system.c:78-127).  q, dq [B][nq]; energy [B][2]. */
int tg_batch_energy(tg_batch *b, const double *q_host, const double *dq_host, double *energy_host);

/* DSystem.set(X[s][k], U[s][k], k, xk_hint = X[s][k+1]) for every (s, k) at once (reference
 * trep/discopt/dsystem.py:229-251 as used by linearize_trajectory, :406-423, and calc_newton_model,
 * doptimizer.py:333-335): the batch must hold seeds*horizon trajectories, trajectory t = s*horizon + k;
 * lambda1 is reset to 0 and the dynamic part of X[s][k+1] is the Newton start, like the reference. */
int tg_batch_set_from_trajectories(tg_batch *b, int32_t seeds, int32_t horizon, double t0, double dt,
                                   const double *X_dev, const double *U_dev, int32_t max_iterations);
/* initialize_from_state(t, Q, p) (midpointvi.py:145-153) with trajectory b's (Q, p) read from the head of the
 * DSystem state vector X_dev[b*row_stride_doubles ...] = [Q; p; v]; lambda1 <- 0. */
int tg_batch_initialize_from_state_device(tg_batch *b, double t, const double *X_dev, uint64_t row_stride_doubles);
/* tg_batch_rollout_closed_loop for the first n_trajectories trajectories of the batch only, with an optional
 * indirection for the gain schedules: group g (= trajectory / group_size) uses Kproj_dev[group_select_dev[g]].
 * Used for the later rounds of a batched Armijo search, when only some seeds are still searching. */
int tg_batch_rollout_closed_loop_subset(tg_batch *b, int32_t n_trajectories, int32_t n_steps, double dt,
                                        const double *Kproj_dev, int32_t group_size, const int32_t *group_select_dev,
                                        const double *bX_dev, const double *bU_dev, double *X_dev, double *U_dev,
                                        int32_t max_iterations);
/* DSystem.fdx / fdu of every solved step (dsystem.py:284-317): A_dev [batch][nX][nX], B_dev [batch][nX][nU],
 * written directly by the first-derivative kernel (the twelve reference-layout arrays are not produced). */
int tg_batch_linearize(tg_batch *b, double *A_dev, double *B_dev);
/* tg_batch_deriv2_contract with device-resident z [batch][nX] and hz [batch][R][R]. */
int tg_batch_deriv2_contract_device(tg_batch *b, const double *z_dev, double *hz_dev);
/* ... for the trajectories s * horizon + k, k_begin <= k < k_end, of a batch of seeds * horizon trajectories only (same arrays, same layout) */
int tg_batch_deriv2_contract_device_range(tg_batch *b, const double *z_dev, double *hz_dev, int32_t horizon, int32_t k_begin, int32_t k_end);

/* Time-varying LQ problem (reference trep/discopt/dlqr.py:41-81; with q_dev = r_dev = NULL and no curvature it
 * is solve_tv_lqr, dlqr.py:9-38).  Weights: Q_k = Q_dev[s*Q_seed_stride + k*Q_step_stride + ...] (strides in
 * doubles, 0 = shared), terminal Qf, R_k likewise.  If hz_dev != NULL the z-contracted second derivative of the
 * dynamics HZ [S][N][hz_R][hz_R] (variables ordered x-part[hz_nx], u-part[nU]) is added on the fly:
 * Q_k += HZ[:hz_nx,:hz_nx], S_k = HZ[:hz_nx, hz_nx:], R_k += HZ[hz_nx:, hz_nx:]  (the Newton model of
 * doptimizer.py:319-345).  Outputs: K [S][N][nU][nX], C [S][N][nU] (affine only), P0 [S][nX][nX], b0 [S][nX],
 * status [S] (TG_OK / TG_SINGULAR).  Q_k and Qf must be symmetric (every P_k then is: the recursion symmetrises
 * it like the reference does). */
typedef struct tg_lq_problem {
    int32_t n_problems, horizon, nX, nU;
    const int32_t *select_dev;
    const double *A_dev, *B_dev;
    const double *Q_dev;  int64_t Q_seed_stride, Q_step_stride;
    const double *Qf_dev; int64_t Qf_seed_stride;
    const double *R_dev;  int64_t R_seed_stride, R_step_stride;
    const double *hz_dev; int32_t hz_R, hz_nx;
    const double *q_dev, *r_dev;          /* [S][N+1][nX], [S][N][nU] or both NULL */
    double *K_dev, *C_dev, *P0_dev, *b0_dev;
    int32_t *status_dev;
    double *b_next_dev;                   /* optional (with q, r): [S][N][nX], row k = b_{k+1}, the affine term entering step k.  With
                                           * the projection weights (Q = R = I) and the cost gradients as q, r this is the adjoint
                                           * z_{k+1} of the Newton model (doptimizer.py:340-343): tg_adjoint_sweep for free */
    int32_t ds_nd, ds_nk, ds_nu;          /* optional: A_k, B_k have the block structure of DSystem.fdx / fdu (dsystem.py:284-317) for a system with
                                           * ds_nd dynamic and ds_nk kinematic configs and ds_nu force inputs -- states [Qd | Qk | p | v], inputs
                                           * [u | rho]: A's Qk rows and v columns are zero and a v row holds one entry (its Qk column); B's Qk and v
                                           * rows hold one entry each (their rho column).  The sweep then skips those blocks in its matrix products
                                           * (about half of the matrix-core work at the puppet's sizes); they must hold exact zeros.  ds_nd = 0: dense.
                                           * Requires nX = 2 (ds_nd + ds_nk), nU = ds_nu + ds_nk. */
    int32_t k_begin, k_end;               /* optional: sweep only the steps k_end - 1 ... k_begin of the horizon (k_end = 0: all of it).  Every array
                                           * keeps its full-horizon layout.  With k_end < horizon the sweep starts from ... */
    const double *Pt_dev, *bt_dev;        /* ... (P, b) at step k_end: [S][nX][nX], [S][nX] -- the P0_dev / b0_dev an earlier launch over the steps
                                           * [k_end, ...) wrote (which, with k_begin > 0, are (P, b) at step k_begin).  A horizon swept in chunks
                                           * this way gives bit for bit the gains of one sweep: the pipelined Newton step of BatchDOptimizer runs the
                                           * projection sweep, the second derivatives and the Newton-model sweep chunk by chunk in three stream lanes */
} tg_lq_problem;
int tg_tv_lq(int32_t device, const tg_lq_problem *problem);

/* Backward adjoint of the Newton model (doptimizer.py:319-345): Z[s][k] = z_{k+1}, the vector the second
 * derivatives of step k are contracted with; z_k = q_k - K_k' r_k + (A_k - B_k K_k)' z_{k+1}, z_N = q_N. */
int tg_adjoint_sweep(int32_t device, int32_t n_problems, int32_t horizon, int32_t nX, int32_t nU,
                     const int32_t *select_dev, const double *A_dev, const double *B_dev, const double *K_dev,
                     const double *q_dev, const double *r_dev, double *Z_dev);
/* Descent direction from the LQ solution (doptimizer.py:391-402): dU_k = -K_k dX_k - C_k,
 * dX_{k+1} = A_k dX_k + B_k dU_k, dX_0 = 0; dcost[s] = sum_k q_k.dX_k + r_k.dU_k (calc_dcost, :262-270). */
int tg_tangent_rollout(int32_t device, int32_t n_problems, int32_t horizon, int32_t nX, int32_t nU,
                       const int32_t *select_dev, const double *A_dev, const double *B_dev, const double *K_dev,
                       const double *C_dev, const double *q_dev, const double *r_dev, double *dX_dev, double *dU_dev,
                       double *dcost_dev);
/* DCost (trep/discopt/dcost.py:5-118): cost[t] = sum_k 1/2 (x-xd)'Q(x-xd) + 1/2 (u-ud)'R(u-ud) + terminal
 * with Qf, for n_trajectories trajectories of which `group` consecutive ones share the reference of one seed:
 * trajectory t is compared with Xd/Ud of seed select_dev[t/group] (or t/group if select_dev is NULL). */
int tg_quadratic_cost(int32_t device, int32_t n_trajectories, int32_t group, const int32_t *select_dev, int32_t horizon,
                      int32_t nX, int32_t nU, const double *X_dev, const double *U_dev, const double *Xd_dev, const double *Ud_dev,
                      const double *Q_dev, const double *R_dev, const double *Qf_dev, double *cost_dev);
/* Its gradients q [S][N+1][nX] (row N = terminal), r [S][N][nU] (dcost.py:62-84). */
int tg_quadratic_cost_gradients(int32_t device, int32_t n_problems, int32_t horizon, int32_t nX, int32_t nU,
                                const int32_t *select_dev, const double *X_dev, const double *U_dev,
                                const double *Xd_dev, const double *Ud_dev, const double *Q_dev, const double *R_dev,
                                const double *Qf_dev, double *q_dev, double *r_dev);
/* Armijo candidates (doptimizer.py:436-446): row c = i*n_lambdas + m of bX/bU is X[s_i] + lambda_m dX[s_i]. */
int tg_armijo_candidates(int32_t device, int32_t n_problems, int32_t n_lambdas, int32_t horizon, int32_t nX,
                         int32_t nU, const int32_t *select_dev, const double *lambdas_dev, const double *X_dev,
                         const double *U_dev, const double *dX_dev, const double *dU_dev, double *bX_dev,
                         double *bU_dev);
/* dst[dst_rows[i]] = src[src_rows[i]], rows of row_doubles doubles (NULL index = identity). */
int tg_copy_rows(int32_t device, int32_t n_rows, uint64_t row_doubles, const int32_t *dst_rows_dev,
                 const int32_t *src_rows_dev, const double *src_dev, double *dst_dev);
/* Stream lane of the calling thread for every discrete-optimisation launch above (tg_tv_lq ... tg_copy_rows): 0 = the
 * device's default stream (initial state), 1 / 2 = two ordinary streams per device.  Streams created without flags order
 * themselves against the default stream in both directions, so kernels launched on lanes 1 and 2 run side by side and
 * anything launched after returning to lane 0 waits for both -- BatchDOptimizer runs the projection-gain sweep and the
 * quasi-Newton LQ sweep of a step this way when the GPU has idle CUs (the reference runs them one after the other,
 * doptimizer.py:462-480; the results do not depend on it). */
int tg_dopt_use_stream(int32_t device, int32_t lane);
/* lanes 1 .. 4: the lane's HIP stream (hipStream_t as void *) -- e.g. for tg_batch_set_stream --, and "lane `waiter` continues after
 * everything enqueued so far in lane `signal`" (an event, no host synchronisation) */
void *tg_dopt_lane_stream(int32_t device, int32_t lane);
int tg_dopt_lane_wait(int32_t device, int32_t waiter, int32_t signal);
int tg_device_synchronize(int32_t device);

/* ---- multi-GPU: one process per GPU, batch sharded, one collective ----------------------
 * The reference has no counterpart (one trajectory per System object, one process).  The only exchange of the path
 * is an all-gather of per-trajectory results (terminal states, costs) after a rollout -- what the discopt line search
 * (trep/discopt/doptimizer.py:405-459) looks at -- plus scalar reductions for barriers / timings.  Implemented on
 * RCCL directly (librccl is dlopen'ed on first use; xGMI on the GPU box).  Rank 0 makes the 128-byte id with
 * tg_comm_unique_id and passes it to the other ranks out of band BEFORE they call tg_comm_create. */
#define TG_COMM_ID_BYTES 128
enum { TG_REDUCE_SUM = 0, TG_REDUCE_MAX = 1, TG_REDUCE_MIN = 2 };
typedef struct tg_comm tg_comm;
int tg_comm_unique_id(uint8_t id_out[TG_COMM_ID_BYTES]);
tg_comm *tg_comm_create(int32_t device, int32_t world, int32_t rank, const uint8_t id_in[TG_COMM_ID_BYTES]);
void tg_comm_destroy(tg_comm *comm);
int tg_comm_info(const tg_comm *comm, int32_t out[3]);   /* ranks and this rank as RCCL reports them (ncclCommCount / ncclCommUserRank), device */
/* recv_dev [world][bytes_per_rank] <- every rank's send_dev [bytes_per_rank]; asynchronous on the communicator's own
 * stream.  That stream is ordered after work on the device's NULL stream only: a buffer produced on another stream (a
 * tg_batch's: tg_batch_stream) must be handed over with tg_comm_wait_stream(comm, producer_stream) first -- it records an
 * event on the producer and makes the communicator's stream wait for it (no host synchronisation).
 * tg_comm_all_gather_after does both in one call.  tg_comm_stream_wait_comm is the other direction (a consumer stream waits for the collective). */
int tg_comm_all_gather(tg_comm *comm, const void *send_dev, void *recv_dev, uint64_t bytes_per_rank);
int tg_comm_wait_stream(tg_comm *comm, void *producer_hip_stream);
int tg_comm_all_gather_after(tg_comm *comm, void *producer_hip_stream, const void *send_dev, void *recv_dev, uint64_t bytes_per_rank);
int tg_comm_stream_wait_comm(tg_comm *comm, void *consumer_hip_stream);
int tg_comm_synchronize(tg_comm *comm);
/* In-place reduction of n host doubles over all ranks (blocking); tg_comm_barrier is a 1-element sum. */
int tg_comm_all_reduce_host(tg_comm *comm, double *values, int32_t n, int32_t op);
int tg_comm_barrier(tg_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* TREP_AMD_H */
