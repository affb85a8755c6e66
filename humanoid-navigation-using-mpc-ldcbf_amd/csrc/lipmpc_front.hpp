// lipmpc_front.hpp -- a problem's inputs and the front end of a step: theta / omega, closest point and normal per obstacle, presolve, compaction (HumanoidMpc.py:137-160, 296-319)
// Part of the MI355X-native batched LIP-MPC / LDCBF step solver (csrc/lipmpc_kernel.hpp includes the parts in order).
#pragma once
#include "lipmpc_comm.hpp"
#include "lipmpc_geometry.hpp"

namespace lipmpc_dev {

// ------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------
// local row slots of a lane
constexpr int R_RU = 0, R_RL = 1, R_VU = 2, R_VL = 3, R_M = 4, R_CBF = 5;

// one problem's inputs as the group sees them / what the closed loop needs back
struct StepIn {
  double p0x, v0x, p0y, v0y, th0, gx, gy, foot0, delta;
  double vmax_x, vmax_y, alpha_over_pi, omega_max;   // per-problem bounds (handle values unless overridden)
  long pb;          // problem index (obstacle arrays, step outputs)
  bool valid;       // false: padding group of the last workgroup (computes, never writes)
  bool sensor_overflow = false;   // the producer of the given half-spaces dropped obstacles (lipmpc_lidar_c_eta_batch: overflow): not solved
#ifdef LIPMPC_PHASE_TIMING
  unsigned long long t_start_wall = 0ull, t_start_ticks = 0ull;      // kernel entry (dev instrumentation)
#endif
};
struct StepOut {
  int status, iters;
  double ux, uy, theta1, omega0, obj;   // first footstep, next heading, first turning rate, objective
};

// Closed-loop warm start (LIPMPC_FLAG_WARM_START, rollout kernel): the interior-point result of a step -- position and
// multipliers per lane -- parked in LDS (row r of the group's block: lane-contiguous) until the next step reads it back
// SHIFTED by one stage, i.e. from lane + 2 (oracle: shift_warm_start; the reference seeds its next solve with the
// shifted prediction, HumanoidMpc.py:450-455).  In LDS rather than registers: the state is dead through the whole solve.
struct WarmIO {
  double* lds;               // [1 + WARM_ROWS][G] doubles of this group, or nullptr: no warm start
  bool have;                 // a previous step's result is parked there
};

// per-problem overrides of (V_MAX_x, V_MAX_y, ALPHA, OMEGA_MAX) — the knobs bounds_tuning.py:17-26 sweeps
__device__ __forceinline__ void load_bounds(const KArgs& P, const double* __restrict__ bounds, long pb, StepIn& in) {
  in.vmax_x = P.v_max[0]; in.vmax_y = P.v_max[1]; in.alpha_over_pi = P.alpha_over_pi; in.omega_max = P.omega_max;
  if (bounds) {
    in.vmax_x = bounds[pb * 4 + 0]; in.vmax_y = bounds[pb * 4 + 1];
    in.alpha_over_pi = bounds[pb * 4 + 2] * (1.0 / M_PI); in.omega_max = bounds[pb * 4 + 3];
  }
}

// one problem's inputs, as every lane of its group reads them (the same 64 B: one broadcast transaction)
__device__ __forceinline__ StepIn load_step_in(const KArgs& P, long pb, bool valid, const double* __restrict__ state,
                                               const double* __restrict__ goal, const int8_t* __restrict__ first_foot,
                                               const double* __restrict__ delta_in, const double* __restrict__ bounds,
                                               const int32_t* __restrict__ overflow_in) {
  StepIn in;
#ifdef LIPMPC_PHASE_TIMING
  in.t_start_wall = wall_clock64(); in.t_start_ticks = (unsigned long long)__builtin_readcyclecounter();
#endif
  in.valid = valid;
  in.pb = pb;
  load_bounds(P, bounds, pb, in);
  in.p0x = state[pb * 5 + 0]; in.v0x = state[pb * 5 + 1]; in.p0y = state[pb * 5 + 2]; in.v0y = state[pb * 5 + 3];
  in.th0 = state[pb * 5 + 4];
  in.gx = goal[pb * 2 + 0]; in.gy = goal[pb * 2 + 1];
  in.foot0 = first_foot ? (double)first_foot[pb] : 1.0;
  in.delta = delta_in ? delta_in[pb] : 0.0;
  in.sensor_overflow = overflow_in && overflow_in[pb] != 0;
  return in;
}

// What the front end of a step hands to its solve: headings of the lane's stage, the obstacles' half-spaces COMPACTED in
// LDS (obs[slot] = eta_x, eta_y, b = eta.c + delta, kfirst = first stage whose row of this obstacle is in the problem;
// perm[slot] = the obstacle's index in the caller's list, for the canonical row numbers), and the ballast row of the presolve.
template <int G> struct FrontOut {
  double th_r, th_v, om_a, theta1, omega0, s_own, c_own;   // theta_a, theta_{a+1}, omega_a of the lane's stage; theta_1, omega_0; sin / cos of the lane's angle
  double n_ball, s_ball;                                    // presolve: number of dropped rows, their mean slack at p_0
  int front_flag;                                           // 1: a constant k = 0 row is violated, 2: degenerate geometry,
                                                            // 8: more obstacles keep a row than the solver body holds (split launch: cannot happen)
  int n_rel;                                                // obstacle slots in use (group-uniform)
  double h0_min, rows_kept;                                 // clearance of the nearest obstacle (its row's value at p_0), LDCBF rows in
                                                            // the solve: what the split launch's cost hint looks at (group-uniform)
#ifdef LIPMPC_PHASE_TIMING
  unsigned long long t_front = 0ull, t_geom = 0ull;         // wall clock at the end of the front end / of its heading arithmetic
#endif
};

// Front end of a step (shared by every solver body of a kernel): theta / omega, closest point and normal per obstacle,
// presolve, compaction of the obstacles that still have a row into the leading slots.
template <int G, int MAXOBS, bool PREFETCH = true>
__device__ __forceinline__ FrontOut<G> front_end(
    const KArgs& P, const StepIn& in, const double* __restrict__ obs_xy, const int32_t* __restrict__ obs_nv,
    double* __restrict__ theta_out, double* __restrict__ omega_out, double* __restrict__ c_eta,
    const double* __restrict__ c_eta_in, bool cold, double (*lds_ring)[2], double (*lds_obs)[4], int* lds_perm, int* lds_flag_g) {
  constexpr int RING_CAP = (G == 16) ? 64 : 256;
  FrontOut<G> F;
  const int tid = threadIdx.x;
  const int lane = tid & (G - 1);
  const int grp = tid / G;
  const bool valid = in.valid;
  const long pb = in.pb;
  const int N = P.N;
  const int a = lane >> 1;             // stage index: variable = p_{a+1}
  const int c = lane & 1;              // coordinate
  const double p0x = in.p0x, p0y = in.p0y, th0 = in.th0;
  const double gx = in.gx, gy = in.gy, delta = in.delta;

  // ---- the obstacle data of the problem: every global load issued NOW, consumed after the heading arithmetic ------------
  // (a group's rings, vertex counts / given half-spaces depend on the problem index alone; fetched where they are used they
  // were two further memory round trips in a row behind the state's -- 2 us of a wave's 8 us fixed cost at one wave per SIMD)
  constexpr int RING_REGS = (2 * RING_CAP + G - 1) / G;      // doubles of the staged rings per lane
  constexpr int SWEEPS = (MAXOBS + G - 1) / G;               // obstacle sweeps of the group
  const bool staged = MAXOBS > 0 && !c_eta_in && P.n_obs * P.nvert_max <= RING_CAP;     // wave-uniform
  // (PREFETCH = false: the closed-loop kernel, whose register file is full -- it fetches where it stores, as before)
  double ring_pre[MAXOBS > 0 ? RING_REGS : 1];
  int nv_pre[SWEEPS > 0 ? SWEEPS : 1];
  double ce_pre[SWEEPS > 0 ? SWEEPS : 1][4];
  auto fetch_obstacles = [&]() {
    if (staged) {
      const double* src = obs_xy + pb * (long)P.n_obs * P.nvert_max * 2;
      const int total = P.n_obs * P.nvert_max * 2;
#pragma unroll
      for (int r = 0; r < RING_REGS; ++r) { const int v = lane + r * G; ring_pre[r] = (v < total) ? src[v] : 0.0; }
    }
#pragma unroll
    for (int sw = 0; sw < SWEEPS; ++sw) {
      const int j = sw * G + lane;
      nv_pre[sw] = 0;
      ce_pre[sw][0] = ce_pre[sw][1] = ce_pre[sw][2] = ce_pre[sw][3] = 0.0;
      if (j < P.n_obs) {
        if (c_eta_in) {
          const double* ce = c_eta_in + (pb * P.n_obs + j) * 4;
          ce_pre[sw][0] = ce[0]; ce_pre[sw][1] = ce[1]; ce_pre[sw][2] = ce[2]; ce_pre[sw][3] = ce[3];
        } else {
          nv_pre[sw] = obs_nv[pb * P.n_obs + j];
        }
      }
    }
  };
  if constexpr (MAXOBS > 0 && PREFETCH) fetch_obstacles();

  // ---- theta / omega (HumanoidMpc.py:137-160) -------------------------------------------------
  const double psi = atan2(gy - p0y, gx - p0x);
  double th_r = 0.0, th_v = 0.0, om_a = 0.0, theta1 = th0, omega0 = 0.0;
  {
    double th = th0;
    if (valid && lane == 0 && theta_out) theta_out[pb * (N + 1)] = th0;
    for (int k = 0; k < N; ++k) {
      double w = fmin(fmax(psi - th, -in.omega_max), in.omega_max);
      double thn = th + w * P.tau;
      if (k == a) { th_r = th; th_v = thn; om_a = w; }
      if (k == 0) { theta1 = thn; omega0 = w; }
      if (valid && lane == 0 && theta_out) { omega_out[pb * N + k] = w; theta_out[pb * (N + 1) + k + 1] = thn; }
      th = thn;
    }
  }
  F.th_r = th_r; F.th_v = th_v; F.om_a = om_a; F.theta1 = theta1; F.omega0 = omega0;
  // R(theta_a) and W(theta_{a+1}): one sincos per lane (the c = 0 lane of a stage takes theta_a, its partner
  // theta_{a+1}), exchanged inside the stage by the solve
  sincos(c ? th_v : th_r, &F.s_own, &F.c_own);
#ifdef LIPMPC_PHASE_TIMING
  F.t_geom = wall_clock64();
#endif

  // ---- obstacles: c_j, eta_j at the current CoM (HumanoidMpc.py:296-319) ----------------------
  // Presolve (oracle: presolve_ldcbf): every feasible p_k lies within k * reach_step of p_0, so the LDCBF row of obstacle j
  // at stage k is REDUNDANT -- never active, never violated -- where its value at p_0 exceeds |eta_j| k reach_step by a
  // margin; such rows leave the problem (kfirst_j = the first stage that keeps its row) and n_d copies of one ballast row
  // 0.q <= s_bar (their mean slack) keep their averaging effect on mu / sigma in the interior-point phase.  ONE rule, the same
  // in both oracles and in the launcher's choice of kernel: the presolve runs unless a flag says the interior iterates matter
  // (LIPMPC_FLAG_INTERIOR, LIPMPC_FLAG_WARM_START) or turns it off (LIPMPC_FLAG_NO_PRESOLVE) -- whether or not this particular
  // step actually has a warm start to read.
  // Compaction (cold start only: a warm start parks per-slot state between steps): the obstacles that still have a row move
  // to the leading slots, so that the wave can run the smallest solver body that holds them (step_body).
  const bool presolve = !(P.flags & (LIPMPC_FLAG_INTERIOR | LIPMPC_FLAG_NO_PRESOLVE | LIPMPC_FLAG_WARM_START));
  const bool compact = cold;
  double nd_l = 0.0, ss_l = 0.0;          // this lane's share of n_d and of the dropped rows' slack sum
  double h0min_l = INFINITY, np_l = 0.0;  // ... of the smallest h0 and of the number of present obstacles
  if (lane == 0) *lds_flag_g = 0;
  if constexpr (MAXOBS > 0) {
    if (staged) {
      double* dst = &lds_ring[0][0];
      const int total = P.n_obs * P.nvert_max * 2;
      if constexpr (PREFETCH) {
#pragma unroll
        for (int r = 0; r < RING_REGS; ++r) { const int v = lane + r * G; if (v < total) dst[v] = ring_pre[r]; }
      } else {
        const double* src = obs_xy + pb * (long)P.n_obs * P.nvert_max * 2;
        for (int v = lane; v < total; v += G) dst[v] = src[v];
      }
    }
  }
  if (MAXOBS > 0) {
    for (int j = lane; j < MAXOBS; j += G) {       // every slot starts empty (kfirst = +inf), harmless values
      lds_obs[j][0] = 0.0; lds_obs[j][1] = 0.0; lds_obs[j][2] = 0.0; lds_obs[j][3] = INFINITY;
      lds_perm[j] = j;
    }
  }
  wave_sync();
  int n_rel = 0;
  if (MAXOBS > 0) {
#pragma unroll
    for (int sw = 0; sw < SWEEPS; ++sw) {          // (uniform trip count: the compaction is a ballot over the wave)
      const int j0 = sw * G;
      const int j = j0 + lane;
      bool keep = false;                           // this obstacle takes a slot
      double cx = 0, cy = 0, ex = 0, ey = 0, bb = 0, h0 = INFINITY, kfirst = INFINITY;
      if (j < P.n_obs) {
        const long oidx = pb * P.n_obs + j;
        bool there, degen = false;
        if (c_eta_in) {      // caller-supplied half-spaces (the reference's _get_list_c_and_eta hook): eta = (0,0) = empty slot, NaN = degenerate
          if constexpr (PREFETCH) { cx = ce_pre[sw][0]; cy = ce_pre[sw][1]; ex = ce_pre[sw][2]; ey = ce_pre[sw][3]; }
          else { const double* ce = c_eta_in + oidx * 4; cx = ce[0]; cy = ce[1]; ex = ce[2]; ey = ce[3]; }
          there = (ex != 0.0) || (ey != 0.0);
          degen = (ex != ex) || (ey != ey);       // NaN normal: the producer met degenerate geometry (lipmpc_lidar_c_eta_batch)
        } else {
          const int nv = PREFETCH ? nv_pre[sw] : obs_nv[oidx];
          there = nv > 0;
          if (there) {
            const ClosestPoint cp = staged ? closest_point_impl<PREFETCH ? 5 : 1>(&lds_ring[j * P.nvert_max][0], nv, p0x, p0y)
                                           : closest_point_normal(obs_xy + oidx * P.nvert_max * 2, nv, p0x, p0y);
            cx = cp.cx; cy = cp.cy; ex = cp.ex; ey = cp.ey;
            degen = cp.degenerate != 0;
          }
        }
        if (there) {
#pragma clang fp contract(off)                   // (the same roundings as the oracles: the screening test below compares them)
          const double ec = ex * cx + ey * cy;
          bb = ec + delta;
          h0 = (ex * p0x + ey * p0y) - ec - delta;
          if (degen) atomicOr(lds_flag_g, 2);
          else if (h0 < -P.k0_tol) atomicOr(lds_flag_g, 1);
          h0min_l = fmin(h0min_l, h0);
          np_l += 1.0;
          int kf = 1;
          if (presolve) {
            const double es = sqrt(ex * ex + ey * ey) * P.reach_step;
            while (kf <= N && h0 > es * (double)kf + SCREEN_MARGIN) ++kf;
            nd_l += (double)(kf - 1);
            ss_l += (double)(kf - 1) * h0;
          }
          kfirst = (double)kf;
          keep = kf <= N;                          // (every stage dropped: the obstacle is in the ballast only)
        }                                          // nv == 0: empty slot
        if (c_eta && valid) {
          double* o = c_eta + oidx * 4;
          o[0] = cx; o[1] = cy; o[2] = ex; o[3] = ey;
        }
      }
      // slot: position among the group's obstacles that keep a row, or the caller's own slot
      const unsigned long long bal = __ballot(keep);
      const unsigned gm = (unsigned)(bal >> (grp * G)) & (G == 32 ? 0xffffffffu : 0xffffu);
      const int pos = compact ? n_rel + __popc(gm & ((1u << lane) - 1u)) : j;
      if (keep) {
        lds_obs[pos][0] = ex; lds_obs[pos][1] = ey; lds_obs[pos][2] = bb; lds_obs[pos][3] = kfirst;
        lds_perm[pos] = j;
      }
      n_rel += __popc(gm);
    }
    if (!compact) n_rel = P.n_obs;                 // slots by the caller's numbering: all of them count
  }
  wave_sync();
  F.front_flag = *lds_flag_g;
  F.n_rel = n_rel;
  // the ballast row: n_d copies of 0.q <= s_bar
  F.n_ball = MAXOBS > 0 ? gsum<G>(nd_l) : 0.0;
  F.s_ball = F.n_ball > 0.0 ? gsum<G>(ss_l) / F.n_ball : 0.0;
  F.h0_min = INFINITY; F.rows_kept = 0.0;
  if constexpr (PREFETCH && MAXOBS > 0) {          // (the closed-loop kernel has no use for them)
    F.h0_min = gmin<G>(h0min_l);
    F.rows_kept = gsum<G>(np_l) * (double)N - F.n_ball;
  }
#ifdef LIPMPC_PHASE_TIMING
  F.t_front = wall_clock64();
#endif
  return F;
}


}  // namespace lipmpc_dev
