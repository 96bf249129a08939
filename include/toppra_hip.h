/*
 * toppra_hip.h -- C-ABI of the MI355X-native batched TOPP-RA "seidel" path.
 *
 * This is the drop-in boundary: a plain-C shared library (libtoppra_hip.so) with no torch or
 * Python types in its signatures.  Each entry point names the reference interface it replaces
 * (paths relative to the reference root, hungpham2511/toppra v0.6.2).  The reference calls its
 * solver once per stage from Python (3N calls per trajectory); this library takes over at the
 * *pass* level and adds the batch dimension, so one call solves B independent trajectories.
 *
 * Layouts are trajectory-major, C-contiguous, fp64 (int32 for status / active sets):
 *   coef   [B][4][nseg][d]   scipy CubicSpline.c of each path (interpolator.py:419), highest
 *                            power first
 *   breaks [nseg+1]          spline breakpoints (CubicSpline.x); [B][nseg+1] with
 *                            TPR_BREAKS_PER_TRAJ
 *   grid   [N+1]             path discretisation; [B][N+1] with TPR_GRID_PER_TRAJ
 *   vlim   [B][d][2]         JointVelocityConstraint.vlim      (linear_joint_velocity.py:19-28)
 *   alim   [B][d][2]         JointAccelerationConstraint.alim  (linear_joint_acceleration.py:46-52)
 *   sd_start, sd_end [B]     boundary path velocities (NULL = 0)
 *
 * Buffers belong to the caller.  Pointers are host pointers unless TPR_DEVICE_PTRS is set, in
 * which case every pointer in tpr_problem/tpr_result is a device pointer on the current device
 * and the call is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default).
 *
 * Numerical failure is data, not an error code, exactly as in the reference: the per-trajectory
 * status is TPR_STATUS_* and failed outputs are NaN-filled.  The int return value reports API
 * misuse only (0 = ok, negative = error; text via tpr_last_error()).
 */
#ifndef TOPPRA_HIP_H
#define TOPPRA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPR_MAX_DOF 32      /* generic lane-per-trajectory kernel (rows per LP: nC = 2 + 4*d <= 130)          */
#define TPR_MAX_DOF_FAST 16 /* rows-across-lanes kernels (auto at 14..16 dof and for mid-size batches); 17..32 dof: family 4 / 1 */

/* tpr_problem.flags */
#define TPR_HAS_VELOCITY 1      /* JointVelocityConstraint present                             */
#define TPR_HAS_ACCELERATION 2  /* JointAccelerationConstraint present                         */
#define TPR_ACC_INTERPOLATION 4 /* DiscretizationType.Interpolation (reference default), else Collocation */
#define TPR_DEVICE_PTRS 8
#define TPR_BREAKS_PER_TRAJ 16
#define TPR_GRID_PER_TRAJ 32
/* bit 64 is reserved (it selected an approximate mode in round 1, retired: every path is bit-exact now) */
/* sd_start / sd_end (and sdmin / sdmax of tpr_controllable_sets_batch) hold the SQUARED boundary velocities
 * x = sd^2, squared by the caller.  The reference squares them with Python's `**` (reachability_algorithm.py:226,
 * :262): libm pow() for Python floats, which is not correctly rounded (one ulp off sd * sd for ~0.08 % of the
 * values), numpy's sd * sd for arrays.  Without the flag the device computes sd * sd; a caller that must reproduce
 * the Python-float case squares in that very expression and sets the flag (toppra_amd.algorithm.TOPPRA does).
 * Honoured by tpr_solve_batch, tpr_controllable_sets_batch, tpr_solve_desired_duration_batch.                  */
#define TPR_BOUNDARY_SQUARED 256
/* Force every stage LP through the full Seidel iteration (served by the rows-across-lanes kernels).
 * By default the throughput kernels answer a backward LP from a certificate -- the reference's own pivot trace followed
 * with margins far above the solver's tolerances, the final vertex verified against every row, and the reference's
 * last-pivot formulas evaluated for it; what is not predictable runs the full iteration.  Both give the same bits
 * (tests/test_gpu_fullsize.py on the GPU, tests/test_host_cert.py on the CPU: the certificate source against the
 * restatement of the reference, stage LP by stage LP); this flag exists for A/B testing and for callers who want the
 * iteration itself replicated.                                                                     */
#define TPR_STRICT_SEIDEL 128
/* Accepted and ignored since round 4: every kernel family certifies a stage LP only where the reference's WHOLE pivot
 * sequence is predictable (rounds 2-3 had a faster default that bounded the reference's last pivot only and could return
 * an LP's optimum where an earlier pivot of the reference ends its run with "infeasible" -- a 1e-14 sliver, 1 in 1.5e5
 * trajectories of an adversarial family).  Family 3 follows the trace at lane level (prologue pivots, the slide along the
 * limiting row as prefix records: DESIGN.md section 3.1), family 2 certifies a moved pair after one predictable pivot,
 * family 4 the kept pair only; everything else runs the reference's iteration.                                       */
#define TPR_SOUND_CERTIFICATES 512

/* per-trajectory status == ParameterizationReturnCode (algorithm/algorithm.py:49-62) */
#define TPR_STATUS_OK 0
#define TPR_STATUS_FAIL_UNCONTROLLABLE 1
#define TPR_STATUS_ERR_UNKNOWN 2

/* API error codes */
#define TPR_E_OK 0
#define TPR_E_BADARG (-1)
#define TPR_E_HIP (-2)
#define TPR_E_UNSUPPORTED (-3)

typedef struct tpr_problem {
    int32_t B, d, nseg, N;
    int32_t flags;
    int32_t variant; /* kernel selection: 0 = auto, 1 = generic lane-per-trajectory, 2 = rows-across-lanes,
                        3 = lane-per-trajectory certificates (d <= 15; sd2, u, status required),
                        4 = one trajectory per wave (the latency kernel: any dof, N <= 1480; auto for
                            small batches),
                        5 = two trajectories per wave, 32 lanes each (the fused solve for 1..7 dof, N <= ~800;
                            auto between 1536 and 9215 trajectories) */
    const double *coef;
    const double *breaks;
    const double *grid;
    const double *vlim;
    const double *alim;
    const double *sd_start;
    const double *sd_end;
    int32_t *active; /* [B][4] = active_c_up[2], active_c_down[2] (may be NULL = a fresh object): the warm-start
                        state of the reference's seidelWrapper OBJECT (cy_seidel_solverwrapper.pyx:526-527), which
                        persists across the passes run on one instance and is also written by the forward pass's
                        1-D path (:646-649).  Read at the start and updated at the end of tpr_solve_batch,
                        tpr_controllable_sets_batch and tpr_feasible_sets_batch, so that a sequence of passes on one
                        object (examples/plot_kinematics.py:48,72) returns the reference's bits.  Maintained by kernel
                        family 4 (auto-selected when set) and, where that family cannot take the problem (N > 1480),
                        by the generic lane kernel (family 1, auto-selected then); forcing variant 2 or 3 together
                        with a non-NULL state is refused (TPR_E_UNSUPPORTED): those families neither read nor update
                        it.  The other entries ignore it (tpr_solve_stagewise_batch takes its own argument). */
} tpr_problem;

typedef struct tpr_result {
    double *sd2;     /* [B][N+1]    x_i = sd_i^2           (may be NULL)                          */
    double *sd;      /* [B][N+1]    sd_vec = sqrt(x)       (may be NULL)                          */
    double *u;       /* [B][N]      sdd_vec                (may be NULL; tpr_solve_batch keeps it in a
                                                            workspace then, like K)                 */
    double *K;       /* [B][N+1][2] controllable sets      (may be NULL for tpr_solve_batch: kept in
                                                            a stream-ordered workspace then)        */
    int32_t *status; /* [B]                                (may be NULL)                          */
} tpr_result;

/* Library / device management.  tpr_init(device) verifies that `device` is a gfx950 GPU and makes it
 * the default device of the CALLING THREAD's host-pointer calls (a thread that never called tpr_init
 * uses the device of the process's last tpr_init); it must succeed once before any other call and
 * fails (TPR_E_HIP / TPR_E_UNSUPPORTED) otherwise.  The calling thread's current HIP device is NOT
 * changed: every entry point runs on the device its data lives on -- the device of the pointers with
 * TPR_DEVICE_PTRS, the thread's tpr_init device otherwise -- and restores the caller's device on
 * return, so the library is safe next to frameworks (torch) that move the per-thread current device,
 * and threads working on different GPUs do not disturb each other.                                 */
int tpr_init(int device);
int tpr_device_count(void);
const char *tpr_last_error(void);
const char *tpr_version(void);
/* ABI guard (0.2): the structures of this header grow at their END from version to version (tpr_problem.active and
 * tpr_dense_problem.active joined in 0.2); a binding built against an older header would hand over a shorter structure
 * and the library would read past it.  A binding compares these sizes (bytes of tpr_problem, tpr_result,
 * tpr_dense_problem; NULL to skip one) with its own declarations before its first compute call.  Returns the ABI
 * revision (2).                                                                                                      */
int tpr_abi_sizes(int32_t *problem_bytes, int32_t *result_bytes, int32_t *dense_problem_bytes);

/* Replaces ReachabilityAlgorithm.compute_parameterization (+ TOPPRA._forward_step) for B
 * trajectories: algorithm/reachabilitybased/reachability_algorithm.py:240-376,
 * time_optimal_algorithm.py:55-92, including seidelWrapper.__init__'s
 * compute_constraint_params (cy_seidel_solverwrapper.pyx:425-531) which is fused in.            */
int tpr_solve_batch(const tpr_problem *p, const tpr_result *r, void *stream);

/* Replaces TOPPRAsd.compute_parameterization (algorithm/reachabilitybased/
 * desired_duration_algorithm.py:42-234) for B trajectories: the backward scan, the "fastest" and
 * "slowest" forward scans, the duration bisection on their convex combination (absolute tolerance
 * atol, reference default 1e-5) and the blended sd^2 / u.  desired [B] seconds; alpha [B] (may be
 * NULL) receives the blend factor.  Same result struct and status codes as tpr_solve_batch; up to 16 dof.
 * p->variant: 0 = auto -- from 9216 trajectories (9..15 dof: 14336 .. 36864) the certified lane kernel runs the
 * backward scan and both forward profiles in ONE launch, the rows-across-lanes kernels otherwise; 2 / 3 force one.
 * Bisection and blend: one wave per trajectory.                                                      */
int tpr_solve_desired_duration_batch(const tpr_problem *p, const double *desired, double atol,
                                     const tpr_result *r, double *alpha, void *stream);

/* Robust TOPP-RA (BASELINE config 4) -- PARITY UNPINNED against ECOS (absent here; the
 * reference holds no golden vectors for it); cross-checked at 1e-7 against an independent exact solver
 * (oracle/robust_independent.py, tests/test_gpu_robust.py); see csrc/tpr_robust.hip.inc.  Replaces
 * TOPPRA([JointVelocityConstraint, RobustLinearConstraint(JointAccelerationConstraint, ellipsoid)],
 * ..., solver_wrapper="ecos"): compute_parameterization (+ compute_feasible_sets into X when X !=
 * NULL) with the stage problems ecosWrapper.solve_stagewise_optim builds
 * (solverwrapper/ecos_solverwrapper.py:90-207, constraint/conic_constraint.py:19-26) solved exactly
 * instead of by ECOS's interior-point iteration.  ellipsoid = (ru, rx, rc) axes lengths.  The
 * acceleration discretisation follows p->flags (TPR_ACC_INTERPOLATION or Collocation).  p->variant: 0 = auto (rows
 * across lanes up to 16 dof -- Interpolation or Collocation, with or without X --, the generic one-trajectory-per-lane
 * kernel above), 1 = the generic kernel; same bits.                                                   */
int tpr_robust_solve_batch(const tpr_problem *p, const double *ellipsoid, const tpr_result *r, double *X,
                           void *stream);

/* Replaces ReachabilityAlgorithm.compute_controllable_sets(sdmin, sdmax)
 * (reachability_algorithm.py:166-238).  sdmin/sdmax [B]; K [B][N+1][2].  Kernel selection and flags as
 * tpr_solve_batch (backward scan only).                                                           */
int tpr_controllable_sets_batch(const tpr_problem *p, const double *sdmin, const double *sdmax,
                                double *K, void *stream);

/* Replaces ReachabilityAlgorithm.compute_feasible_sets (reachability_algorithm.py:131-164).
 * X [B][N+1][2].  p->variant: 0 = auto (one wave per trajectory for a handful of trajectories, above 16 dof or with
 * p->active; the certified lane kernel from 8192 trajectories up to 8 dof, 14336 .. 36864 at 9 .. 15 dof; rows across lanes
 * otherwise), 1 / 2 / 3 / 4 force a kernel family.  TPR_STRICT_SEIDEL / TPR_SOUND_CERTIFICATES as tpr_solve_batch.   */
int tpr_feasible_sets_batch(const tpr_problem *p, double *X, void *stream);

/* Replaces ReachabilityAlgorithm.compute_reachable_sets(sdmin, sdmax) (reachability_algorithm.py:378-431):
 * the feasible sets, then the forward propagation of the reachable velocities from [sdmin^2, sdmax^2] --
 * including the reference's use of the PREVIOUS interval's delta in the objective (:384) and the warm-start
 * state shared with the feasible-set pass.  sdmin/sdmax [B]; L [B][N+1][2] (zeros after a NaN stage, as in
 * the reference); X [B][N+1][2] feasible sets (may be NULL).                                          */
int tpr_reachable_sets_batch(const tpr_problem *p, const double *sdmin, const double *sdmax, double *L, double *X,
                             void *stream);

/* ---- the seidel path on DENSE rows: any canonical-linear constraint list ------------------------------
 * The entries above regenerate the rows of JointVelocityConstraint + JointAccelerationConstraint from the spline table.
 * seidelWrapper itself is more general (cy_seidel_solverwrapper.pyx:425-531): it flattens ANY list of canonical-linear
 * constraints (SecondOrderConstraint / JointTorqueConstraint with a user's inverse dynamics, varying velocity limits,
 * hand-written a, b, c, F, g) into a_arr, b_arr, c_arr [N+1][nC] (nC counts the two reserved x_next rows 0, 1),
 * low_arr, high_arr [N+1][2] and deltas [N]; solve_stagewise_optim (:549-697) and the scans of
 * reachability_algorithm.py:131-376 read nothing else.  These entries take exactly those arrays for B trajectories
 * (the host side builds them as the reference does: toppra_amd.solverwrapper.dense_rows) and run every stage LP through
 * the reference's full Seidel iteration with its warm-start state (rows across 8 / 16 / 32 lanes per trajectory; nC <= 122).
 * Results are the reference's bits.  flags: TPR_DEVICE_PTRS, TPR_BOUNDARY_SQUARED.                          */
typedef struct tpr_dense_problem {
    int32_t B, N, nC, flags;
    const double *a, *b, *c;         /* [B][N+1][nC]; entries 0, 1 of a stage are ignored (the x_next rows) */
    const double *low, *high;        /* [B][N+1][2]  variable boxes (u, x) */
    const double *deltas;            /* [B][N] */
    const double *sd_start, *sd_end; /* [B] or NULL (zeros) */
    int32_t *active;                 /* [B][4] or NULL: the wrapper object's warm-start state (active_c_up[2], active_c_down[2]),
                                        read at the start of a pass and written back at its end (as tpr_problem.active): passes
                                        chained on ONE object pivot in the reference's order.  NULL = a fresh object. */
} tpr_dense_problem;

/* compute_parameterization (reachability_algorithm.py:240-376): outputs and status codes as tpr_solve_batch (r->K is
 * required; sd2 / sd / u / status may be NULL).                                                            */
int tpr_solve_dense_batch(const tpr_dense_problem *p, const tpr_result *r, void *stream);
/* TOPPRAsd.compute_parameterization (desired_duration_algorithm.py:42-234) on dense rows: as
 * tpr_solve_desired_duration_batch (desired [B] seconds, atol, alpha [B] may be NULL).                       */
int tpr_solve_desired_duration_dense_batch(const tpr_dense_problem *p, const double *desired, double atol,
                                           const tpr_result *r, double *alpha, void *stream);
/* compute_controllable_sets(sdmin, sdmax) (:166-238): K [B][N+1][2]; p->sd_start / sd_end are not used.       */
int tpr_controllable_sets_dense_batch(const tpr_dense_problem *p, const double *sdmin, const double *sdmax, double *K,
                                      void *stream);
/* compute_feasible_sets (:131-164): X [B][N+1][2].                                                          */
int tpr_feasible_sets_dense_batch(const tpr_dense_problem *p, double *X, void *stream);
/* compute_reachable_sets(sdmin, sdmax) (:378-431): as tpr_reachable_sets_batch (L [B][N+1][2]; X may be NULL); one
 * trajectory per lane, the reference's per-stage calls with their stateful warm start.                     */
int tpr_reachable_sets_dense_batch(const tpr_dense_problem *p, const double *sdmin, const double *sdmax, double *L, double *X,
                                   void *stream);

/* Replaces Constraint.compute_constraint_params + the dense row build of seidelWrapper.__init__
 * (linear_joint_velocity.py:43-53, linear_joint_acceleration.py:63-104,
 * linear_constraint.py:164-190, cy_seidel_solverwrapper.pyx:474-520):
 *   a,b,c [B][N+1][nC] (rows 0,1 zero), low, high [B][N+1][2] (the wrapper's variable box),
 *   xbound [B][N+1][2] (the velocity constraint's own x bound, before the +-1e8 box is applied),
 *   qs, qss [B][N+1][d]
 * Any output may be NULL.  nC = 2 + (4d | 2d | 0) by flags.                                      */
int tpr_constraint_params_batch(const tpr_problem *p, double *a, double *b, double *c, double *low,
                                double *high, double *xbound, double *qs, double *qss, void *stream);

/* Replaces seidelWrapper.solve_stagewise_optim (cy_seidel_solverwrapper.pyx:549-697) for ONE
 * stage of each of B trajectories (the compatibility entry; 1 LP per call per trajectory).
 *   stage [B]; g [B][2]; xb [B][4] = x_min, x_max, x_next_min, x_next_max (NaN = absent);
 *   active [B][4] = active_c_up[2], active_c_down[2] warm-start state, updated in place;
 *   solve_lp1d: the constructor flag (:425); out [B][2] = [u, x] or NaNs.                        */
int tpr_solve_stagewise_batch(const tpr_problem *p, const int32_t *stage, const double *g,
                              const double *xb, int32_t *active, int solve_lp1d, double *out,
                              void *stream);

/* Replaces solve_lp1d / solve_lp2d (cy_seidel_solverwrapper.pyx:42-87 -> :93-144, :149-390) for n
 * independent LPs of nrows rows each (the known-answer-test entry).
 *   lp1d: v [n][2], a,b [n][nrows], low,high [n]
 *         -> result [n], optval [n], optvar [n], active [n]
 *   lp2d: v [n][3], a,b,c [n][nrows], low,high [n][2], active_in [n][2]
 *         -> result [n], optval [n], optvar [n][2], active_out [n][2]                            */
int tpr_lp1d_batch(int n, int nrows, const double *v, const double *a, const double *b,
                   const double *low, const double *high, int32_t *result, double *optval,
                   double *optvar, int32_t *active, void *stream);
int tpr_lp2d_batch(int n, int nrows, const double *v, const double *a, const double *b,
                   const double *c, const double *low, const double *high, const int32_t *active_in,
                   int32_t *result, double *optval, double *optvar, int32_t *active_out,
                   void *stream);

/* Replaces SplineInterpolator.__init__ -> scipy.interpolate.CubicSpline (interpolator.py:360-421)
 * for B paths: waypoints [B][m][d] at knots [m] (knots_per_path == 0) or [B][m] -> coef
 * [B][4][m-1][d], the layout tpr_problem.coef takes.  Boundary conditions per end:
 * TPR_BC_NOT_A_KNOT, TPR_BC_FIRST_DERIV (value [B][d], NULL = 0: scipy's "clamped"),
 * TPR_BC_SECOND_DERIV (NULL = 0: "natural").  m >= 2 (splines of more than 64 points keep their
 * working arrays in a stream-ordered global workspace of 6 m doubles per spline).  device_ptrs != 0: all
 * pointers are device pointers.                                                                    */
#define TPR_BC_NOT_A_KNOT 0
#define TPR_BC_FIRST_DERIV 1
#define TPR_BC_SECOND_DERIV 2
int tpr_spline_fit_batch(int B, int m, int d, const double *knots, int knots_per_path,
                         const double *waypoints, int bc_start, int bc_end, const double *bc_start_val,
                         const double *bc_end_val, double *coef, int device_ptrs, void *stream);

/* Replaces ParametrizeConstAccel (toppra/parametrizer.py:23-158) for B trajectories of one shape.
 * times:  _process_parametrization -- sd [B][N+1] -> ts [B][N+1] (running sum in the reference's
 *         order), us [B][N] (may be NULL).  Only N, B, flags (TPR_GRID_PER_TRAJ, TPR_DEVICE_PTRS) and
 *         grid of `p` are used.
 * eval:   __call__(t, order) -- times [B][T] -> out [B][T][d] = q(t) (order 0), dq/dt (1), d2q/dt2 (2),
 *         using p->coef/breaks/grid and the sd, ts, us arrays of the same trajectories.            */
int tpr_const_accel_times_batch(const tpr_problem *p, const double *sd, double *ts, double *us, void *stream);
int tpr_const_accel_eval_batch(const tpr_problem *p, const double *sd, const double *ts, const double *us,
                               int T, const double *times, int order, double *out, void *stream);

/* Replaces ParametrizeSpline (toppra/parametrizer.py:161-196), the reference's default output
 * parametrizer (algorithm/algorithm.py:121-125), for B trajectories: the gridpoint time stamps (a running
 * sum; a gridpoint reached in less than TINY = 1e-8 s is dropped, a standing stretch counts 5 s), the
 * waypoints q(s_i) and the cubic spline in time through them, clamped to q'(s) sd at both ends (scipy
 * CubicSpline arithmetic as in tpr_spline_fit_batch).  sd [B][N+1] -> knot_times [B][N+1] (compacted;
 * entries from counts[b] on are padding), counts [B] (gridpoints kept), coef_t [B][4][N][d] (segments from
 * counts[b]-1 on are a constant extension).  Uses p->coef/breaks/grid/flags and p->variant: 0 = auto, 1 = generic (two
 * kernels, any d), 2 = one fused kernel in LAPACK dgtsv's elimination order (d <= 64; the bits of scipy on an FMA-free
 * LAPACK), 3 = knot-parallel (d <= 16, all knots in LDS; cyclic reduction instead of dgtsv: the knot derivatives agree to
 * rounding, q(t) within 1e-10 of the reference's samples like variant 2; twice as fast).  Auto picks 3 where it fits,
 * else 2, else 1.  Time stamps, counts and waypoints are the same bits in every variant.
 * tpr_ppoly_eval_batch evaluates such tables -- SplineInterpolator.__call__(t, order), i.e. scipy PPoly
 * (interpolator.py:423-430) -- at times [B][T] -> out [B][T][d]; breaks [B][nseg+1]; counts may be NULL.   */
/* ParametrizeSpline + SplineInterpolator.__call__ in one launch: what a caller of compute_trajectory() does next
 * (traj(ts), traj(ts, 1), traj(ts, 2): examples/plot_kinematics.py:52-57) without the [B][4][N][d] coefficient table in
 * between -- the knot-parallel fit (variant 3 of tpr_param_spline_batch: d <= 16, knots in LDS) evaluates the samples from
 * its LDS-resident knot derivatives; the same bits as tpr_param_spline_batch(variant 3) + tpr_ppoly_eval_batch.
 * times [B][T] (times_per_traj) or [T]; fractions != 0: times are fractions of each trajectory's duration
 * (linspace(0, 1, T) * duration).  q / qd / qdd [B][T][d] (order 0 / 1 / 2; any may be NULL), duration [B] (may be NULL). */
int tpr_param_spline_sample_batch(const tpr_problem *p, const double *sd, int T, const double *times, int times_per_traj,
                                  int fractions, double *q, double *qd, double *qdd, double *duration, void *stream);
int tpr_param_spline_batch(const tpr_problem *p, const double *sd, double *knot_times, int32_t *counts,
                           double *coef_t, void *stream);
int tpr_ppoly_eval_batch(int B, int nseg, int d, const double *coef, const double *breaks, const int32_t *counts,
                         int T, const double *times, int order, double *out, int device_ptrs, void *stream);

/* Measurement helper used by bench.py: launches the tpr_solve_batch kernel(s) `reps` times on
 * `stream` between two hipEvents recorded on that same stream and returns the average
 * milliseconds per launch (device pointers required).                                            */
int tpr_solve_batch_timed(const tpr_problem *p, const tpr_result *r, void *stream, int reps,
                          float *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* TOPPRA_HIP_H */
