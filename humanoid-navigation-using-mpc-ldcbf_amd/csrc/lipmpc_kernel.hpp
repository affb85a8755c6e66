// lipmpc_kernel.hpp — batched LIP-MPC / LDCBF step solver for gfx950 (MI355X), C ABI in include/lipmpc.h.
//
// One QP per group of G lanes (G = 16 for N <= 8, 32 for N <= 16; 4 or 2 problems per
// 64-wide wavefront).  Lane l of a group owns decision variable l = 2a + c, i.e. coordinate c
// of the CoM position p_{a+1} (position form of the QP: H = 2I), and the
// inequality rows of "its" stage:
//   c = 0: leg-reach x (upper, lower), walking-velocity longitudinal (upper, lower), manoeuvrability
//   c = 1: leg-reach y (upper, lower), walking-velocity lateral (upper, lower)
//   LDCBF rows of stage a+1: obstacle j on lane c = j & 1.
// Rows are generated, never stored: G q, G^T w and K = 2I + G^T D G are applied through the
// problem's structure (rotation blocks, the alternating-sum velocity map, per-stage 2x2 LDCBF
// blocks), so a problem's live state is n + ~3 m doubles in registers.  For large obstacle sets
// (more than 7 LDCBF rows per lane) those rows are STREAMED instead: (s, z) per row in LDS, everything
// else recomputed in each pass (step_solve, STREAM).  Horizons up to 4 run the factorisation on 8 variable
// slots (NVAR = 8).
// A step = front_end (theta / omega, closest point and normal per obstacle, presolve of the LDCBF rows the leg-reach rows
// make redundant, compaction of the obstacles that still have a row) + step_solve<G, NOBS_L, NVAR> (interior point +
// certified primal active-set finish on NOBS_L row slots per lane); step_body picks the body: in the exact mode with the
// presolve the smallest of {1, 2, 7, the handle's} slots that holds the wave's neediest problem.  Kernels: plan_step_kernel
// (one step for B problems, optionally in the cost order of the previous launch: lipmpc_set_schedule; DISPATCH = with /
// without the small bodies) and rollout_kernel (the whole closed loop per robot, one launch).
//
// Reference semantics followed (HumanoidNavigation/...):
//   theta/omega            MPC/HumanoidMpc.py:137-160
//   closest point / eta    Utils/ObstaclesUtils.py:50-109
//   rows                   MPC/HumanoidMpc.py:183-249, 252-294; HumanoidMPCCustomLCBF.py:30-31
//   cost                   MPC/HumanoidMpc.py:321-333
//   dynamics / advance     MPC/HumanoidMpc.py:34-48, 335-343, 432-447
// Solver: Mehrotra predictor-corrector on the normal equations + certified active-set finish
// (the algorithm of oracle/lipmpc_oracle.py, which is the parity checker, not a dependency).



#pragma once
#include "lipmpc_solve.hpp"

namespace lipmpc_dev {

// The whole MPC step of one problem on one group of G lanes: front end, then the SMALLEST solver body that holds the
// obstacles which still have a row after the presolve (1, 2, 7 or the handle's NOBS_L row slots per lane; the wave takes the
// body its neediest group needs).  On the BASELINE fields a step keeps 0-3 of 10 obstacles (N = 8) or 0-6 of 50 (N = 16), so
// the 2-slot body solves what the 5-slot / the 25-slot streamed body was sized for.  DISPATCH = false (the closed-loop kernel,
// whose callers run the interior mode, where every present obstacle keeps its rows): the handle's body only -- one body per
// kernel keeps that kernel's register allocation.  Output pointers may be null.
template <int G, int NOBS_L, int NVAR = G, bool DISPATCH = true, bool LEAN = false>
__device__ __forceinline__ StepOut step_body(
    const KArgs& P, const StepIn& in, const double* __restrict__ obs_xy, const int32_t* __restrict__ obs_nv,
    double* __restrict__ U, double* __restrict__ X, double* __restrict__ theta_out,
    double* __restrict__ omega_out, double* __restrict__ obj_out, int32_t* __restrict__ status_out,
    int32_t* __restrict__ iters_out, unsigned long long* __restrict__ active_out,
    unsigned long long* __restrict__ working_out, double* __restrict__ c_eta,
    double* __restrict__ diag, const double* __restrict__ c_eta_in = nullptr, WarmIO* __restrict__ warm = nullptr,
    int32_t* __restrict__ cost_out = nullptr) {
  constexpr int GPW = 64 / G;
  constexpr int MAXOBS = 2 * NOBS_L;
  // a problem's obstacle rings staged in LDS by one coalesced sweep of the group when they fit (n_obs x v_max vertices
  // <= RING_CAP: the 10 x 5 of BASELINE config 2 do): the edge walk of the closest-point step then reads LDS instead
  // of paying one global-memory latency per edge
  constexpr int RING_CAP = (G == 16) ? 64 : 256;
  __shared__ double lds_ring[GPW][MAXOBS > 0 ? RING_CAP : 1][2];
  __shared__ double lds_obs[GPW][MAXOBS > 0 ? MAXOBS : 1][4];   // eta_x, eta_y, b = eta.c + delta, kfirst
  __shared__ int lds_perm[GPW][MAXOBS > 0 ? MAXOBS : 1];
  __shared__ int lds_flag[GPW];
  __shared__ int lds_need[GPW];
  const int grp = threadIdx.x / G;
  const bool cold = (warm == nullptr) || (warm->lds == nullptr);
  const FrontOut<G> F = front_end<G, MAXOBS, !LEAN>(P, in, obs_xy, obs_nv, theta_out, omega_out, c_eta, c_eta_in, cold, lds_ring[grp],
                                                    lds_obs[grp], lds_perm[grp], &lds_flag[grp]);
#define LIPMPC_SOLVE(NL) step_solve<G, NL, NVAR, LEAN>(P, in, F, lds_obs[grp], lds_perm[grp], U, X, obj_out, status_out, iters_out, active_out, working_out, diag, warm, cost_out)
  if constexpr (DISPATCH && NOBS_L > 1) {
    if ((threadIdx.x & (G - 1)) == 0) lds_need[grp] = (F.n_rel + 1) >> 1;      // row slots per lane this group's obstacles need
    wave_sync();
    int need = 0;
#pragma unroll
    for (int g = 0; g < GPW; ++g) need = max(need, lds_need[g]);
    need = __builtin_amdgcn_readfirstlane(need);
    if (need <= 1) return LIPMPC_SOLVE(1);
    if constexpr (NOBS_L > 2) { if (need <= 2) return LIPMPC_SOLVE(2); }
    if constexpr (NOBS_L > 7) { if (need <= 7) return LIPMPC_SOLVE(7); }
  }
  return LIPMPC_SOLVE(NOBS_L);
#undef LIPMPC_SOLVE
}

// ------------------------------------------------------------------------------------------
// kernel 1: one MPC step for B problems (lipmpc_plan_step_batch)
// ------------------------------------------------------------------------------------------
// DISPATCH: the kernel holds the small solver bodies next to the handle's own and a wave picks the smallest that fits (exact
// mode with the presolve); false: the handle's body alone (interior mode / LIPMPC_FLAG_NO_PRESOLVE, where every present
// obstacle keeps its rows -- and where the streamed body keeps the register allocation it has when it is alone: inlined next
// to the small bodies it spills into its row sweeps, 59 instead of 36 us per iteration at N = 16 / 50 obstacles).
template <int G, int NOBS_L, int NVAR, bool DISPATCH>
__global__ __launch_bounds__(WAVE) void plan_step_kernel(
    KArgs P, long B, const double* __restrict__ state, const double* __restrict__ goal,
    const int8_t* __restrict__ first_foot, const double* __restrict__ delta_in,
    const double* __restrict__ obs_xy, const int32_t* __restrict__ obs_nv,
    double* __restrict__ U, double* __restrict__ X, double* __restrict__ theta_out,
    double* __restrict__ omega_out, double* __restrict__ obj_out, int32_t* __restrict__ status_out,
    int32_t* __restrict__ iters_out, unsigned long long* __restrict__ active_out,
    unsigned long long* __restrict__ working_out, double* __restrict__ c_eta,
    double* __restrict__ diag, const double* __restrict__ bounds, const double* __restrict__ c_eta_in,
    int32_t* __restrict__ sched, const int32_t* __restrict__ overflow_in) {
  constexpr int GPW = WAVE / G;
  static_assert(G == 16 || G == 32, "a problem is one or two DPP rows of one wavefront");
  if (blockDim.x != WAVE) __builtin_trap();          // wave_sync() and every group exchange assume a one-wave workgroup
  const long prob_raw = (long)blockIdx.x * GPW + threadIdx.x / G;
  // Which problem this group solves: its position in the launch, or -- on a schedule (lipmpc_set_schedule) -- the problem
  // the order left by the previous launch puts there: by descending cost (iterations + finish rounds).  That starts the
  // long solves first, and, as important, puts problems of like cost into the same wave: a wave lasts as long as the
  // slowest of its groups, and the mean of that maximum over four random problems is 12 % above the mean problem.
  long pb = prob_raw < B ? prob_raw : (B - 1);
  if (sched && sched[SCHED_VALID] == (int)B) {
    const long r = sched[SCHED_ORDER + pb];
    if (r >= 0 && r < B) pb = r;
  }
  const StepIn in = load_step_in(P, pb, prob_raw < B, state, goal, first_foot, delta_in, bounds, overflow_in);
  step_body<G, NOBS_L, NVAR, DISPATCH>(P, in, obs_xy, obs_nv, U, X, theta_out, omega_out, obj_out, status_out, iters_out, active_out,
                                       working_out, c_eta, diag, c_eta_in, nullptr, sched ? sched + SCHED_ORDER + B : nullptr);
}

// ------------------------------------------------------------------------------------------
// kernels 1b: the split launch of one MPC step (see SPLIT_CLASSES above): classify -> bin -> one kernel per solver body
// ------------------------------------------------------------------------------------------
// The class of every problem: the front end alone (0.7 % of a step at N = 16 / 50 obstacles), one group of G lanes per
// problem exactly as the solving kernels run it -- the same code on the same inputs, hence the same count.
template <int G>
__global__ __launch_bounds__(WAVE) void classify_kernel(
    KArgs P, long B, const double* __restrict__ state, const double* __restrict__ goal, const double* __restrict__ delta_in,
    const double* __restrict__ obs_xy, const int32_t* __restrict__ obs_nv, const double* __restrict__ bounds,
    const double* __restrict__ c_eta_in, int32_t* __restrict__ ws) {
  constexpr int GPW = WAVE / G;
  constexpr int RING_CAP = (G == 16) ? 64 : 256;
  if (blockDim.x != WAVE) __builtin_trap();
  __shared__ double lds_ring[GPW][RING_CAP][2];
  __shared__ double lds_obs[GPW][SPLIT_MAXOBS][4];
  __shared__ int lds_perm[GPW][SPLIT_MAXOBS];
  __shared__ int lds_flag[GPW];
  const int grp = threadIdx.x / G;
  const long prob_raw = (long)blockIdx.x * GPW + grp;
  const long pb = prob_raw < B ? prob_raw : (B - 1);
  const StepIn in = load_step_in(P, pb, prob_raw < B, state, goal, nullptr, delta_in, bounds, nullptr);
  const FrontOut<G> F = front_end<G, SPLIT_MAXOBS>(P, in, obs_xy, obs_nv, nullptr, nullptr, nullptr, c_eta_in, true, lds_ring[grp],
                                                   lds_obs[grp], lds_perm[grp], &lds_flag[grp]);
  // sort key: class, then the cost hint's bucket (0 = dearest)
  if (in.valid && (threadIdx.x & (G - 1)) == 0)
    ws[SPLIT_HEAD + pb] = split_class_of((F.n_rel + 1) >> 1) * SPLIT_BUCKETS +
                          split_cost_bucket(F.h0_min, F.rows_kept, sqrt(in.v0x * in.v0x + in.v0y * in.v0y));
}

// One solver body over its class's list: the step kernel with NL row slots per lane and nothing else in its register
// allocation.  Launched with the grid of the whole batch (the counts live on the device); the blocks past the list's end leave
// at once.
template <int G, int NL, int NVAR>
__global__ __launch_bounds__(WAVE) void solve_list_kernel(
    KArgs P, long B, int cls, const int32_t* __restrict__ ws, const double* __restrict__ state, const double* __restrict__ goal,
    const int8_t* __restrict__ first_foot, const double* __restrict__ delta_in,
    const double* __restrict__ obs_xy, const int32_t* __restrict__ obs_nv,
    double* __restrict__ U, double* __restrict__ X, double* __restrict__ theta_out,
    double* __restrict__ omega_out, double* __restrict__ obj_out, int32_t* __restrict__ status_out,
    int32_t* __restrict__ iters_out, unsigned long long* __restrict__ active_out,
    unsigned long long* __restrict__ working_out, double* __restrict__ c_eta,
    double* __restrict__ diag, const double* __restrict__ bounds, const double* __restrict__ c_eta_in,
    int32_t* __restrict__ cost_out, const int32_t* __restrict__ overflow_in) {
  constexpr int GPW = WAVE / G;
  constexpr int RING_CAP = (G == 16) ? 64 : 256;
  if (blockDim.x != WAVE) __builtin_trap();
  const int count = ws[cls];
  if ((long)blockIdx.x * GPW >= count) return;                    // (wave-uniform; no loop over the list: a loop around the step
                                                                  // costs every body 200-500 B of scratch per lane)
  __shared__ double lds_ring[GPW][RING_CAP][2];
  __shared__ double lds_obs[GPW][SPLIT_MAXOBS][4];
  __shared__ int lds_perm[GPW][SPLIT_MAXOBS];
  __shared__ int lds_flag[GPW];
  const int grp = threadIdx.x / G;
  const long idx = (long)blockIdx.x * GPW + grp;
  const long pb = ws[SPLIT_HEAD + B * (1 + cls) + (idx < count ? idx : count - 1)];
  const StepIn in = load_step_in(P, pb, idx < count, state, goal, first_foot, delta_in, bounds, overflow_in);
  FrontOut<G> F = front_end<G, SPLIT_MAXOBS>(P, in, obs_xy, obs_nv, theta_out, omega_out, c_eta, c_eta_in, true, lds_ring[grp],
                                             lds_obs[grp], lds_perm[grp], &lds_flag[grp]);
  if (F.n_rel > 2 * NL) F.front_flag |= 8;
  step_solve<G, NL, NVAR, false>(P, in, F, lds_obs[grp], lds_perm[grp], U, X, obj_out, status_out, iters_out, active_out, working_out,
                                 diag, nullptr, cost_out);
}

// ------------------------------------------------------------------------------------------
// kernel 2: the closed loop of HumanoidMPC.run_simulation on the device (lipmpc_rollout_batch),
// HumanoidMpc.py:380-459: per sample k: stop if the previous objective < 0.05 (:392); on MPC samples
// (k % mpc_step == 0) solve the step, keep u_0 (:432), advance x+ = A x + B u_0 (:441-442); on the
// other samples only the heading moves (:443-447); theta <- theta_1 (:447); the stance foot of MPC step
// number floor(k / mpc_step) alternates (:104-108, 401-403).  A failed solve ends the robot's run (:419-429).
// Each group owns one robot for the whole run: no host round trip, no batch-wide barrier per step.
// ------------------------------------------------------------------------------------------
template <int G, int NOBS_L, int NVAR>
__global__ __launch_bounds__(WAVE) void rollout_kernel(
    KArgs P, long B, int k_max, int mpc_step, double stop_obj, const double* __restrict__ state0,
    const double* __restrict__ goal, const int8_t* __restrict__ first_foot, const double* __restrict__ delta_in,
    const double* __restrict__ obs_xy, const int32_t* __restrict__ obs_nv, double* __restrict__ X_pred,
    double* __restrict__ U_pred, int32_t* __restrict__ n_steps, int32_t* __restrict__ last_status,
    int32_t* __restrict__ total_iters, const double* __restrict__ bounds) {
  constexpr int GPW = WAVE / G;
  if (blockDim.x != WAVE) __builtin_trap();          // see plan_step_kernel
  const int lane = threadIdx.x & (G - 1);
  const long prob_raw = (long)blockIdx.x * GPW + threadIdx.x / G;
  StepIn in;
  in.valid = prob_raw < B;
  const long pb = in.valid ? prob_raw : (B - 1);
  in.pb = pb;
  load_bounds(P, bounds, pb, in);
  in.p0x = state0[pb * 5 + 0]; in.v0x = state0[pb * 5 + 1]; in.p0y = state0[pb * 5 + 2]; in.v0y = state0[pb * 5 + 3];
  in.th0 = state0[pb * 5 + 4];
  in.gx = goal[pb * 2 + 0]; in.gy = goal[pb * 2 + 1];
  in.foot0 = (double)first_foot[pb];
  in.delta = delta_in ? delta_in[pb] : 0.0;
  double* Xp = X_pred + pb * (long)(k_max + 1) * 5;
  double* Up = U_pred + pb * (long)k_max * 3;
  if (in.valid && lane == 0) { Xp[0] = in.p0x; Xp[1] = in.v0x; Xp[2] = in.p0y; Xp[3] = in.v0y; Xp[4] = in.th0; }
  bool fin = false;
  int k_done = 0, st_last = LIPMPC_STATUS_SOLVED, it_sum = 0;
  double last_obj = INFINITY, ukx = 0.0, uky = 0.0;
  // warm start between MPC steps: register rows only (streamed instantiations start cold), horizons of 2 and more
  constexpr bool CAN_WARM = NOBS_L <= 7;
  const bool use_warm = CAN_WARM && (P.flags & LIPMPC_FLAG_WARM_START) && P.N >= 2;
  __shared__ double lds_warm[GPW][1 + WARM_ROWS][G];
  WarmIO ws;
  ws.lds = use_warm ? &lds_warm[threadIdx.x / G][0][0] : nullptr;
  ws.have = false;
  for (int k = 0; k < k_max; ++k) {
    if (!fin && last_obj < stop_obj) fin = true;
    if (__all(fin)) break;
    if (!fin) {
      const bool is_mpc = (k % mpc_step) == 0;
      double theta1, omega0;
      if (is_mpc) {     // group-uniform (k and mpc_step are wave-uniform)
        const StepOut r = step_body<G, NOBS_L, NVAR, false, true>(P, in, obs_xy, obs_nv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ws);
        if (use_warm) ws.have = true;             // (a failed solve ends the run anyway)
        st_last = r.status;
        it_sum += r.iters;
        theta1 = r.theta1; omega0 = r.omega0;
        if (r.status != LIPMPC_STATUS_SOLVED && r.status != LIPMPC_STATUS_UNCERTIFIED) fin = true;
        else { last_obj = r.obj; ukx = r.ux; uky = r.uy; }
      } else {
        const double psi = atan2(in.gy - in.p0y, in.gx - in.p0x);
        omega0 = fmin(fmax(psi - in.th0, -in.omega_max), in.omega_max);
        theta1 = in.th0 + omega0 * P.tau;
      }
      if (!fin) {
        if (is_mpc) {
          const double px = in.p0x, vx = in.v0x, py = in.p0y, vy = in.v0y;
          in.p0x = P.ch * px + P.sh_over_beta * vx + (1.0 - P.ch) * ukx;
          in.v0x = P.beta_sh * px + P.ch * vx - P.beta_sh * ukx;
          in.p0y = P.ch * py + P.sh_over_beta * vy + (1.0 - P.ch) * uky;
          in.v0y = P.beta_sh * py + P.ch * vy - P.beta_sh * uky;
        }
        in.th0 = theta1;
        if ((k + 1) % mpc_step == 0) in.foot0 = -in.foot0;
        if (in.valid && lane == 0) {
          Up[3 * k] = ukx; Up[3 * k + 1] = uky; Up[3 * k + 2] = omega0;
          double* xo = Xp + (long)(k + 1) * 5;
          xo[0] = in.p0x; xo[1] = in.v0x; xo[2] = in.p0y; xo[3] = in.v0y; xo[4] = in.th0;
        }
        k_done = k + 1;
      }
    }
  }
  if (in.valid && lane == 0) { n_steps[pb] = k_done; last_status[pb] = st_last; total_iters[pb] = it_sum; }
}

// host-side launcher of one instantiation (defined in lipmpc_inst.hip, one object per (G, NOBS_L))
template <int G, int NOBS_L, int NVAR>
void launch_plan_step(const KArgs& k, long B, const double* state, const double* goal, const int8_t* first_foot,
                      const double* delta, const double* obs_xy, const int32_t* obs_nv, double* U, double* X,
                      double* theta, double* omega, double* obj, int32_t* status, int32_t* iters,
                      unsigned long long* active, unsigned long long* working, double* c_eta, double* diag, const double* bounds,
                      const double* c_eta_in, int32_t* sched, const int32_t* overflow_in, hipStream_t stream);
template <int G, int NOBS_L, int NVAR>
void launch_rollout(const KArgs& k, long B, int k_max, int mpc_step, double stop_obj, const double* state0,
                    const double* goal, const int8_t* first_foot, const double* delta, const double* obs_xy,
                    const int32_t* obs_nv, double* X_pred, double* U_pred, int32_t* n_steps, int32_t* last_status,
                    int32_t* total_iters, const double* bounds, hipStream_t stream);
// one solver body of the split launch over its class's list (32 lanes per problem; defined in lipmpc_inst.hip)
template <int G, int NL, int NVAR>
void launch_solve_list(const KArgs& k, long B, int cls, const int32_t* ws, const double* state, const double* goal,
                       const int8_t* first_foot, const double* delta, const double* obs_xy, const int32_t* obs_nv, double* U,
                       double* X, double* theta, double* omega, double* obj, int32_t* status, int32_t* iters,
                       unsigned long long* active, unsigned long long* working, double* c_eta, double* diag, const double* bounds,
                       const double* c_eta_in, int32_t* cost_out, const int32_t* overflow_in, hipStream_t stream);

}  // namespace lipmpc_dev
