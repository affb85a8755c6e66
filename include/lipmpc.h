/* lipmpc.h — C ABI of the MI355X-native batched LIP-MPC / LDCBF step solver.
 *
 * The reference (salvatore373/Humanoid-Navigation-using-MPC-LDCBF) is pure Python and has no
 * FFI of its own; the boundary this library replaces is, per MPC step and per robot,
 *
 *   HumanoidMPC._precompute_theta_omega_naive      HumanoidNavigation/MPC/HumanoidMpc.py:137-160
 *   HumanoidMPC._get_list_c_and_eta                HumanoidNavigation/MPC/HumanoidMpc.py:296-319
 *     -> ObstaclesUtils.get_closest_point_and_normal_vector_from_obs
 *                                                  HumanoidNavigation/Utils/ObstaclesUtils.py:60-109
 *   HumanoidMPC._add_lcbf_constraint (+ CustomLCBF delta)
 *                                                  HumanoidMpc.py:263-294, HumanoidMPCCustomLCBF.py:30-31
 *   the constraint/cost definition                 HumanoidMpc.py:221-249, 321-333
 *   self.optim_prob.solve()  (CasADi Opti + IPOPT) HumanoidMpc.py:97-100, 417-418
 *   kth_solution.value(U_mpc[:,0]) / state advance HumanoidMpc.py:432-447
 *
 * batched over B independent (state, goal, obstacle-set) instances.  All pointers passed to
 * lipmpc_plan_step_batch / lipmpc_advance_batch are DEVICE pointers (HIP, the handle's
 * device); the caller owns every buffer; calls are asynchronous on `hip_stream` and results
 * are valid after that stream is synchronised.  No function throws; return 0 = ok, <0 = error
 * (lipmpc_strerror).  A handle is not thread-safe: one handle per (device, stream).
 */
#ifndef LIPMPC_H
#define LIPMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIPMPC_ABI_VERSION 5   /* 2: + lipmpc_plan_step_batch_c_eta, LIPMPC_STATUS_SENSOR_OVERFLOW; 3: + lipmpc_lidar_c_eta_batch;
                                * 4: lipmpc_plan_step_batch_c_eta takes the producer's overflow flags;
                                * 5: `active` is the primal tight set of the returned point (LIPMPC_TIGHT_TOL), the finish's
                                *    working set moves to the new optional output `working`, diag is [B,8],
                                *    + lipmpc_set_workspace / lipmpc_workspace_bytes */
/* An instrumented development build (tools/build_variant.sh: phase counters in `diag`, other buffer contracts) reports
 * LIPMPC_ABI_VERSION + LIPMPC_VARIANT_BASE from lipmpc_version(), so that a loader which checks the version refuses it. */
#define LIPMPC_VARIANT_BASE 1000

/* `active` bit i: canonical row i is in the problem and TIGHT at the returned point, slack_i(q) = h_i - g_i.q <=
 * LIPMPC_TIGHT_TOL.  The minimiser of the strictly convex step QP is unique, so this set is unique too -- unlike the set
 * of rows with a positive multiplier (`working`), which is not at a degenerate vertex (linearly dependent tight rows). */
#define LIPMPC_TIGHT_TOL 1e-7
#define LIPMPC_DIAG_WORDS 8

/* per-problem status written to status[b] */
#define LIPMPC_STATUS_SOLVED       0  /* exact optimum, KKT-certified active set */
#define LIPMPC_STATUS_MAX_ITER     1  /* interior-point iteration cap reached */
#define LIPMPC_STATUS_INFEASIBLE   2  /* no feasible step (incl. a violated constant k=0 LDCBF row);
                                         the reference raises inside solve() here (HumanoidMpc.py:419-429) */
#define LIPMPC_STATUS_DEGENERATE   3  /* x == c or zero-length edge: reference yields NaN (ObstaclesUtils.py:81,104) */
#define LIPMPC_STATUS_UNCERTIFIED  4  /* interior-point tolerance met, active-set finish not certified */
#define LIPMPC_STATUS_SENSOR_OVERFLOW 5 /* the sample's inferred obstacles did not fit the slots (overflow[b] != 0): the step is not
                                         solved (lipmpc_plan_step_batch_c_eta with overflow flags, lipmpc_sense_plan_step_batch)
                                         and the robot is stopped (lipmpc_fleet_update_batch) rather than planned against a
                                         truncated obstacle list -- the reference constrains against every inferred obstacle,
                                         HumanoidMPCUnknownEnvironment.py:55-64 */

/* flags */
#define LIPMPC_FLAG_INTERIOR 1  /* skip the active-set finish: return the strictly interior
                                   interior-point iterate (what IPOPT hands the reference's loop) */

#define LIPMPC_FLAG_WARM_START 2 /* lipmpc_rollout_batch: every MPC step of a robot's run starts from the previous step's
                                   interior-point result shifted by one stage (positions; multipliers clipped to [3, 100])
                                   instead of "stand still", z = 30 -- the reference seeds its next solve with the shifted
                                   prediction, HumanoidMpc.py:450-455.  Same optimum, fewer iterations (-15..-30 %).
                                   Ignored for more than 14 obstacle slots and for N = 1. */
#define LIPMPC_FLAG_NO_PRESOLVE 4 /* keep every LDCBF row in the solve.  By default (exact mode, cold start) the rows that the
                                   leg-reach rows make redundant -- obstacle j at stage k with eta_j.(p_0 - c_j) - delta >
                                   k * (largest CoM step the reach rows allow) + 1e-3: never active, never violated -- are
                                   dropped before the solve and replaced, in the interior-point phase only, by copies of one
                                   direction-free ballast row (oracle/lipmpc_oracle.py: presolve_ldcbf).  Same feasible set,
                                   same minimiser, same active set; only the interior iterates differ, which is why
                                   LIPMPC_FLAG_INTERIOR and LIPMPC_FLAG_WARM_START imply this flag. */

/* error codes */
#define LIPMPC_OK            0
#define LIPMPC_E_ARG        -1
#define LIPMPC_E_UNSUPPORTED -2
#define LIPMPC_E_HIP        -3
#define LIPMPC_E_NOMEM      -4

typedef struct lipmpc_params {
  int32_t N;            /* horizon, 1..16                         (N_horizon, HumanoidMpc.py:50) */
  int32_t n_obs_max;    /* obstacle slots per problem, 0..50 */
  int32_t v_max;        /* vertex slots per obstacle ring, 3..32 */
  int32_t max_iter;     /* interior-point iteration cap */
  int32_t flags;        /* LIPMPC_FLAG_* */
  int32_t finish_rounds; /* cap on the rounds of the certified active-set finish (tail-latency control:
                           a problem that needs more ends UNCERTIFIED with the interior-point answer); 0 = default (8 for N <= 8, else 16) */
  double dt;            /* DELTA_T            config.yml:2  */
  double g;             /* GRAVITY_CONST      config.yml:3  */
  double h_com;         /* COM_HEIGHT         config.yml:4  */
  double alpha;         /* ALPHA              config.yml:5  */
  double l_max[2];      /* L_MAX_X, L_MAX_Y   config.yml:6-7 */
  double l_min[2];      /* L_MIN_X, L_MIN_Y   config.yml:8-9 */
  double v_min[2];      /* V_MIN              config.yml:10 */
  double v_max_xy[2];   /* V_MAX              config.yml:11 */
  double omega_max;     /* 0.156*pi           HumanoidMpc.py:21 */
  double ell;           /* 0.05               HumanoidMpc.py:200 */
  double sampling_time; /* theta update step  HumanoidMpc.py:159 */
  double tol;           /* interior-point stop (exact path): max|r_p| <= tol and mu <= tol */
  double tol_interior;  /* the same for LIPMPC_FLAG_INTERIOR: sets how far inside its constraints the returned iterate
                           stays (the reference's IPOPT stops at 1e-5); too tight and a closed loop lands on LDCBF
                           boundaries where the next eta = (x-c)/|x-c| is 0/0 */
  double k0_tol;        /* tolerated violation of the constant k=0 LDCBF rows (IPOPT constr_viol_tol, HumanoidMpc.py:99) */
} lipmpc_params;

typedef struct lipmpc_handle lipmpc_handle;

/* fills *p with the reference's config.yml values, N=3, n_obs_max=0, v_max=5, tol=1e-11, tol_interior=1e-9 */
int lipmpc_default_params(lipmpc_params* p);

int lipmpc_create(const lipmpc_params* p, int device, lipmpc_handle** out);
void lipmpc_destroy(lipmpc_handle* h);

/* number of inequality rows in canonical order reach(4N) | manoeuvr(N) | vel(4N) | LDCBF((N+1)*n_obs_max)
 * (insertion order of HumanoidMpc.py:230-249, 284-292) and of 64-bit words of the active mask */
int64_t lipmpc_num_rows(const lipmpc_params* p);
int64_t lipmpc_active_words(const lipmpc_params* p);

/* One MPC step for B problems.
 *  state      [B,5]  (p_x, v_x, p_y, v_y, theta)            X_pred[:,k]      HumanoidMpc.py:396-397
 *  goal       [B,2]                                          self.goal        HumanoidMpc.py:83
 *  first_foot [B]    s_v of the current stance, +1 right / -1 left            HumanoidMpc.py:104-108,403
 *  delta      [B] or NULL (=0)  LDCBF safety margin          HumanoidMPCCustomLCBF.py:30-31
 *  obs_xy     [B,n_obs_max,v_max,2]  CCW rings hull.points[hull.vertices], padded
 *  obs_nv     [B,n_obs_max]          vertices used per ring, 0 = slot empty
 * outputs
 *  U      [B,N,2]    footsteps U_mpc            X [B,N+1,4] predicted states X_mpc
 *  theta  [B,N+1]    omega [B,N]                obj [B] objective incl. the constant k=0 term
 *  status [B]  iters [B]
 *  active  [B,lipmpc_active_words]  bit i = canonical row i is tight at the returned point (slack <= LIPMPC_TIGHT_TOL): the
 *          active set of the optimum in the textbook sense, unique because the optimum is; all zero unless status is SOLVED
 *          or UNCERTIFIED (there: the tight rows of the interior-point iterate handed out)
 *  working [B,lipmpc_active_words] or NULL: the working set the certified finish ended on = the rows that carry a positive
 *          multiplier in its KKT certificate (a subset of `active` up to LIPMPC_TIGHT_TOL; at a degenerate vertex one of
 *          several valid choices); UNCERTIFIED: the interior-point estimate z_i > 1e5 s_i
 *  c_eta  [B,n_obs_max,4] (c_x,c_y,eta_x,eta_y) or NULL
 *  bounds [B,4] or NULL: per-problem (V_MAX_x, V_MAX_y, ALPHA, OMEGA_MAX) replacing the handle's values —
 *         the knobs the reference's bounds_tuning sweep mutates in `conf` (bounds_tuning.py:17-26)
 *  diag   [B,LIPMPC_DIAG_WORDS] or NULL: 0 active-set rounds used, 1 final equality residual of the finish,
 *         2 identification margin min_i |log(z_i/(1e5 s_i))| of the interior-point phase, 3 certificate margin =
 *         min(smallest multiplier on the working set, smallest slack outside it): ~0 flags a weakly determined WORKING set,
 *         4 tightness margin min_i |slack_i - LIPMPC_TIGHT_TOL| over the rows of the problem: how far the nearest row is
 *         from changing sides in `active` (a perturbation of the answer below it leaves `active` unchanged), 5-7 reserved (0)
 */
int lipmpc_plan_step_batch(lipmpc_handle* h, int64_t B,
                           const double* state, const double* goal, const int8_t* first_foot,
                           const double* delta, const double* obs_xy, const int32_t* obs_nv,
                           double* U, double* X, double* theta, double* omega, double* obj,
                           int32_t* status, int32_t* iters, uint64_t* active, uint64_t* working, double* c_eta,
                           double* diag, const double* bounds, void* hip_stream);

/* Optional launch order for the step solves of a handle.  A wave lasts as long as the slowest of its problems and a launch
 * of more problems than the GPU holds at once (4096 at N <= 8) runs in rounds, so WHICH problems share a wave and which start
 * first matters: with a schedule every lipmpc_plan_step_batch / _c_eta launch of at most `capacity` problems leaves there each
 * problem's cost (interior-point iterations + finish rounds) and (one small extra kernel) the order -- costliest first,
 * like with like -- in which the next launch of the same batch size places them (+12 % throughput at 32768 problems when
 * consecutive launches see the same or slowly moving problems, as the samples of a closed loop do).  A pure scheduling
 * hint: outputs stay indexed by problem, every order gives the same results, a buffer holding no order for this B means
 * index order.  `schedule`: device buffer of lipmpc_schedule_words(capacity) int32, zeroed once by the caller, owned by the
 * caller and alive until it is unset (NULL) or the handle destroyed; launches on it must be stream-ordered. */
int lipmpc_set_schedule(lipmpc_handle* h, int32_t* schedule, int64_t capacity);
int64_t lipmpc_schedule_words(int64_t B);

/* Optional workspace for the SPLIT LAUNCH of a handle's step solves.  For 32-lane problems (N > 8) in the exact mode the
 * step kernel holds the solver bodies of 1, 2, 7 and the handle's LDCBF row slots per lane next to each other and a wave
 * picks the smallest that fits its problems after the presolve -- one register allocation for all of them, which spills
 * (304 B of scratch per lane at N = 16 / 50 obstacles).  With a workspace the step runs as: one classification pass (the
 * front end alone: which body each problem needs), a one-workgroup stable counting sort into one index list per body, and ONE
 * KERNEL PER BODY over its list, the kernels side by side on streams the handle owns (fork / join by events on the caller's
 * stream, so the call stays asynchronous and stream-ordered, and can be captured in a graph).  Results are bit-identical
 * to the single-kernel launch; problems of one class share waves.  Ignored (single kernel) for N <= 8, for the flags that
 * keep every row, and for batches larger than the workspace was sized for.  While it is set, the order of
 * lipmpc_set_schedule is not applied (the costs are still left).
 * `workspace`: device buffer of lipmpc_workspace_bytes(h, capacity) bytes, contents arbitrary, owned by the caller, alive
 * until unset (NULL) or the handle is destroyed; launches on it must be stream-ordered. */
int64_t lipmpc_workspace_bytes(const lipmpc_handle* h, int64_t capacity);
int lipmpc_set_workspace(lipmpc_handle* h, void* workspace, int64_t capacity);

/* The same step with the LDCBF half-spaces GIVEN instead of derived from obstacle rings: the reference's subclass
 * hooks HumanoidMPC._get_list_c_and_eta(x_k, y_k) -> (list_c, list_eta) (HumanoidMpc.py:296-319; overridden by
 * HumanoidMPCUnknownEnvironment.py:30-68) and _compute_single_lcbf(x, eta, c) (HumanoidMpc.py:252-261; overridden by
 * HumanoidMPCCustomLCBF.py:30-31) as data.  Row j of every stage k is  eta_j . (p_k - c_j) - delta >= 0  with
 *  c_eta_in [B,n_obs_max,4] (c_x, c_y, eta_x, eta_y); eta need not be a unit vector; eta = (0,0) marks an empty slot,
 *  a NaN in eta marks degenerate geometry met by whoever produced the row (status DEGENERATE, as the ring front end gives).
 *  overflow [B] int32 or NULL: the producer's "obstacles were dropped" flags (lipmpc_lidar_c_eta_batch: the scan's clusters
 *  did not fit the obstacle slots).  A flagged problem is NOT solved against its truncated list -- the reference constrains
 *  against every inferred obstacle (HumanoidMPCUnknownEnvironment.py:55-64) -- it gets status SENSOR_OVERFLOW and NaN
 *  outputs, so lipmpc_advance_batch and every other consumer of `status` leave the robot where it is.
 * Nothing of the geometry front end runs; the constant k = 0 row is still checked against k0_tol.
 * Outputs as lipmpc_plan_step_batch (without c_eta). */
int lipmpc_plan_step_batch_c_eta(lipmpc_handle* h, int64_t B,
                                 const double* state, const double* goal, const int8_t* first_foot,
                                 const double* delta, const double* c_eta_in, const int32_t* overflow,
                                 double* U, double* X, double* theta, double* omega, double* obj,
                                 int32_t* status, int32_t* iters, uint64_t* active, uint64_t* working, double* diag,
                                 const double* bounds, void* hip_stream);

/* Closed-loop state advance (HumanoidMpc.py:432-447): for problems with status SOLVED/UNCERTIFIED
 * state <- (A_l x + B_l U[b,0], theta[b,1]), first_foot <- -first_foot; others are left untouched.
 * In place on state/first_foot. */
int lipmpc_advance_batch(lipmpc_handle* h, int64_t B, double* state, int8_t* first_foot,
                         const double* U, const double* theta, const int32_t* status,
                         void* hip_stream);

/* One sample of a host-driven closed loop for a fleet (the bookkeeping of HumanoidMpc.py:392, 419-447 around a solve
 * that was just enqueued on the same stream), per robot b:
 *   walking[b] &= last_obj[b] >= stop_obj        (stop rule of this sample, from the previous objective, :392)
 *   if walking: last_status[b] = overflow[b] ? SENSOR_OVERFLOW : status[b];  walking[b] &= last_status in {SOLVED, UNCERTIFIED}   (:419-429)
 *   if still walking: last_obj = obj; state <- (A_l x + B_l U[b,0], theta[b,1]); first_foot <- -first_foot (:432-447);
 *                     n_steps[b] += 1; n_overflow[b] += overflow[b] (if given)
 *   U_pred[b, k, :] = (U[b,0,:], omega[b,0]);  X_pred[b, k+1, :] = state[b]   with k = *sample (read on the device)
 * and one thread advances *sample by one, so that a captured graph can be replayed sample after sample.
 * walking [B] int8 (1 = walking), sample [1] int32, X_pred [B,k_max+1,5], U_pred [B,k_max,3]; samples >= k_max are ignored. */
int lipmpc_fleet_update_batch(lipmpc_handle* h, int64_t B, int32_t k_max, double stop_obj,
                              double* state, int8_t* first_foot, int8_t* walking, double* last_obj,
                              int32_t* n_steps, int32_t* last_status, int32_t* n_overflow, int32_t* sample,
                              double* X_pred, double* U_pred,
                              const double* U, const double* theta, const double* omega, const double* obj,
                              const int32_t* status, const int32_t* overflow, void* hip_stream);

/* Closed loop on the device: HumanoidMPC.run_simulation (HumanoidMpc.py:345-459) for B robots, one group of
 * lanes per robot for the whole run, no host round trip.  Per sample k < k_max: stop when the previous
 * step's objective < stop_obj (0.05 in the reference, :392); on MPC samples (k % mpc_step == 0,
 * mpc_step = max(1, int(DELTA_T / sampling_time)), :74-75) solve the step and advance x+ = A x + B u_0
 * (:441-442), on the others only the heading moves (:443-447); a failed solve ends that robot's run
 * (:419-429).  Uses the handle's flags (LIPMPC_FLAG_INTERIOR = advance with the interior iterate).
 *  state0 [B,5], goal [B,2], first_foot [B], delta [B] or NULL, obs_xy/obs_nv as in lipmpc_plan_step_batch
 * outputs
 *  X_pred  [B,k_max+1,5]  (p_x,v_x,p_y,v_y,theta) per sample; rows 0..n_steps[b] are valid
 *  U_pred  [B,k_max,3]    (f_x,f_y,omega) per sample;        rows 0..n_steps[b]-1 are valid
 *  n_steps [B] samples completed, last_status [B] status of the last solve, total_iters [B] sum of IPM iterations
 */
int lipmpc_rollout_batch(lipmpc_handle* h, int64_t B, int32_t k_max, int32_t mpc_step, double stop_obj,
                         const double* state0, const double* goal, const int8_t* first_foot, const double* delta,
                         const double* obs_xy, const int32_t* obs_nv, double* X_pred, double* U_pred,
                         int32_t* n_steps, int32_t* last_status, int32_t* total_iters, const double* bounds,
                         void* hip_stream);

/* Unknown-environment front end (BASELINE config 5): what HumanoidMPCUnknownEnvironment._get_list_c_and_eta does
 * before the closest-point step (HumanoidMPCUnknownEnvironment.py:30-55): range_finder() =
 * compute_lidar_readings -> Gaussian noise -> DBSCAN(eps, min_samples) -> convex hull per cluster
 * (RangeFinder/range_finder_wth_polygons_dbscan.py:26-63, 100-126, 157-180).  One wavefront per robot; the rings
 * come out in the layout lipmpc_plan_step_batch takes as obs_xy / obs_nv.
 *  state      [B,5]  only (p_x, p_y) are read
 *  env_xy     [n_env,v_env,2] if env_shared else [B,n_env,v_env,2]; env_nv likewise: the TRUE map as vertex rings
 *             in the order the reference iterates them (`ch.points`, HumanoidMPCUnknownEnvironment.py:46); n_env <= 65535
 *  ray_table  [resolution,2] (cos, sin) of angle_i = i * 2 pi / resolution, computed on the host (bit-identical
 *             directions to the reference's math.cos / math.sin); resolution <= 384
 *  noise      [B,resolution,2] added to valid readings, or NULL.  The reference draws N(0, 0.01) from numpy's
 *             global, unseeded generator (:162-172); here the caller supplies the (seeded) sample.
 * outputs
 *  obs_xy [B,n_obs_max,v_max,2], obs_nv [B,n_obs_max]: CCW rings of the inferred obstacles, cluster order
 *  n_inferred [B]; overflow [B] = 1 if clusters/vertices did not fit (n_obs_max, v_max), if more than 384 true
 *  obstacles were within range of the robot, or if an env_nv entry exceeded v_env (the surplus is dropped, never read)
 *  hits [B,resolution,2] (NaN = no reading) or NULL; labels [B,resolution] (-2 no reading, -1 noise, k cluster) or NULL
 */
int lipmpc_lidar_sense_batch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                             int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                             int32_t n_obs_max, int32_t v_max, const double* state, const double* env_xy,
                             const int32_t* env_nv, const double* ray_table, const double* noise,
                             double* obs_xy, int32_t* obs_nv, int32_t* n_inferred, int32_t* overflow,
                             double* hits, int32_t* labels, void* hip_stream);

/* Unknown-environment CONSTRAINT ASSEMBLY in one launch (BASELINE config 5: "LiDAR point cloud -> convex-hull obstacle
 * rebuild fused into the constraint-assembly kernel"): everything HumanoidMPCUnknownEnvironment._get_list_c_and_eta does
 * (HumanoidMPCUnknownEnvironment.py:30-68) -- range_finder() as in lipmpc_lidar_sense_batch, then for every inferred
 * hull the closest point c and the unit normal eta at the robot's CoM with the inside flip (:54-62 ->
 * ObstaclesUtils.py:60-109) -- with the hulls never leaving LDS.  The (c, eta) rows are the LDCBF half-spaces
 * lipmpc_plan_step_batch_c_eta solves against; they are bit-identical to what lipmpc_plan_step_batch derives from the
 * rings lipmpc_lidar_sense_batch writes.
 *  inputs as lipmpc_lidar_sense_batch
 *  c_eta [B,n_obs_max,4] (c_x, c_y, eta_x, eta_y) per inferred obstacle, cluster order; empty slots all zero;
 *        eta = NaN where the geometry is degenerate (CoM on the hull boundary, zero-length hull edge)
 *  n_inferred [B], overflow [B] as lipmpc_lidar_sense_batch
 *  obs_xy / obs_nv: the rings as well, or both NULL;  hits, labels: or NULL
 *  schedule: NULL (the robots are scanned in index order), or a device buffer of lipmpc_lidar_schedule_words(B) int32, contents
 *        arbitrary: scratch for the LAUNCH ORDER of this call.  A scan's length grows with its reading count, a whole batch of
 *        4096 robots is resident at once (16 waves per compute unit), and the launch lasts as long as its most loaded SIMD.  With
 *        the buffer the call first ranks its robots -- one small kernel estimates every robot's reading count (all rays for a
 *        robot inside an obstacle, else from the bounding circles of the obstacles in range), a one-workgroup counting sort
 *        turns the estimates into launch positions that give every SIMD a heavy robot with light ones -- and then starts
 *        the scans in that order (both included in the call: ~11 us per 4096 robots).  Nothing carries over
 *        from one call to the next; every order gives the same results; calls sharing a buffer must be stream-ordered. */
int lipmpc_lidar_c_eta_batch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                             int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                             int32_t n_obs_max, int32_t v_max, const double* state, const double* env_xy,
                             const int32_t* env_nv, const double* ray_table, const double* noise,
                             double* c_eta, int32_t* n_inferred, int32_t* overflow, double* obs_xy,
                             int32_t* obs_nv, double* hits, int32_t* labels, int32_t* schedule, void* hip_stream);
int64_t lipmpc_lidar_schedule_words(int64_t B);

/* One MPC step of the unknown-environment variant in ONE call (what HumanoidMPCUnknownEnvironment does per step,
 * HumanoidMPCUnknownEnvironment.py:30-68 + HumanoidMpc.py:387-418): lipmpc_lidar_c_eta_batch (scan, clusters, hulls,
 * closest point / normal: one launch, n_obs_max / v_max from the handle) followed on the same stream by
 * lipmpc_plan_step_batch_c_eta against those half-spaces and the scan's overflow flags (a robot whose scan overflowed gets
 * status SENSOR_OVERFLOW, not a plan).  c_eta [B,n_obs_max,4] is the hand-over buffer (and an output);
 * schedule as in lipmpc_lidar_c_eta_batch or NULL; every other argument as in the two functions. */
int lipmpc_sense_plan_step_batch(lipmpc_handle* h, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                                 int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                                 const double* state, const double* goal, const int8_t* first_foot, const double* delta,
                                 const double* env_xy, const int32_t* env_nv, const double* ray_table, const double* noise,
                                 double* c_eta, int32_t* n_inferred, int32_t* overflow, int32_t* schedule,
                                 double* U, double* X, double* theta, double* omega, double* obj, int32_t* status,
                                 int32_t* iters, uint64_t* active, uint64_t* working, double* diag, const double* bounds,
                                 void* hip_stream);

const char* lipmpc_strerror(int code);
int lipmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LIPMPC_H */
