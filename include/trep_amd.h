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
enum { TG_CONSTRAINT_DISTANCE = 0, TG_CONSTRAINT_POINT = 1 };

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

/* Device memory helpers so a host language without a HIP binding can stage inputs. */
void *tg_device_alloc(int32_t device, uint64_t bytes);
int tg_device_free(int32_t device, void *ptr);
int tg_memcpy_h2d(int32_t device, void *dst_dev, const void *src_host, uint64_t bytes);
int tg_memcpy_d2h(int32_t device, void *dst_host, const void *src_dev, uint64_t bytes);

int tg_batch_synchronize(tg_batch *b);
/* Use an externally created hipStream_t (e.g. torch's current stream); NULL = own stream. */
int tg_batch_set_stream(tg_batch *b, void *hip_stream);
/* HIP-event timing of the kernels launched on the batch's stream since the last reset:
 * number of launches and the sum of their durations in milliseconds. */
int tg_batch_timing(tg_batch *b, int32_t reset, int32_t *n_launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* TREP_AMD_H */
