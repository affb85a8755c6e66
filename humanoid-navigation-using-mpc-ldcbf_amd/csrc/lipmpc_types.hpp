// lipmpc_types.hpp -- solver constants (the values of oracle/lipmpc_oracle.py), schedule / split-launch buffer layouts, kernel arguments
// Part of the MI355X-native batched LIP-MPC / LDCBF step solver (csrc/lipmpc_kernel.hpp includes the parts in order).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/lipmpc.h"

namespace lipmpc_dev {


// solver constants (same values as the oracle)
constexpr double IPM_S_FLOOR = 0.1;
constexpr double IPM_Z0 = 30.0;
constexpr double IPM_STEP_FRAC = 0.995;
constexpr double IPM_Z_DIVERGE = 1e13;
constexpr double IPM_STALL_TOL = 1e-6;   // factorisation breakdown below this (r_p, mu) counts as converged
constexpr int IPM_SLOW_FROM = 8;
constexpr double IPM_SLOW_RATIO = 0.9, IPM_SLOW_SIGMA = 0.5;   // no progress in mu (from iteration 8 on) -> centre up to half way
constexpr double FIN_RHO = 1e10;
constexpr double FIN_EPS = 1e-9;
constexpr int FIN_ROUNDS = 8;            // default of lipmpc_params.finish_rounds for N <= 8 (tail latency: see DESIGN.md)
constexpr int FIN_ROUNDS_LONG = 16;      // ... and for longer horizons (worse conditioned, more exchanges needed)
constexpr double FIN_RHO_POLISH = 1e12, FIN_POLISH_TOL = 1e-10;    // polish round of the finish (oracle: finish_active_set, 5.)
constexpr double FIN_GD_MIN = 1e-14;     // ratio test: a direction component below this does not run into its row
constexpr double FIN_DUAL_REL = 1e-14;   // stationarity tolerance of the certificate: FIN_EPS + this x largest multiplier
constexpr double FIN_IDENT = 1e5;   // initial working set z > FIN_IDENT * s: a deliberate under-estimate (oracle docstring)
constexpr int FIN_INNER = 6;
constexpr double FIN_INNER_TOL = 1e-11;
constexpr double FIN_STALL = 0.5;      // a correction that leaves more than this share of the residual has stalled
constexpr double SCREEN_MARGIN = 1e-3;   // presolve: an LDCBF row is dropped when the leg-reach rows keep it this far from active
constexpr double WARM_Z_MIN = 3.0, WARM_Z_MAX = 100.0;   // closed-loop warm start: band of the shifted previous multipliers
constexpr int WARM_ROWS = 12;                            // register row slots a lane can hold (5 kinematic + 7 LDCBF)

// schedule buffer (int32, lipmpc_set_schedule): [B the order is valid for, -, order[B] (problem at launch position i), cost[B]]
constexpr int SCHED_VALID = 0, SCHED_ORDER = 2, SCHED_COST_BINS = 128;

// Split launch (lipmpc_set_workspace; 32-lane problems, exact mode with the presolve): a classification pass writes each
// problem's class = the smallest solver body that holds the obstacles which keep a row after the presolve, a one-workgroup
// stable counting sort turns the classes into one index list per class, and ONE KERNEL PER BODY solves its list -- each body
// with its own register allocation (inlined into one kernel the 1 / 2 / 7 / 25-slot bodies of the 32-lane dispatching kernel
// share one allocation and spill 304 B per lane).  Workspace (int32): [SPLIT_CLASSES counts, padded to 8 | class of problem
// b: B | list of class c: B each].
constexpr int SPLIT_CLASSES = 5;
constexpr int SPLIT_HEAD = 8;
// Inside a class the list is ordered by a COST HINT, dearest first, in SPLIT_BUCKETS steps: a launch of more waves than the GPU
// holds at once ends when its last wave does, so the long solves should start first, and problems of like cost should share a
// wave.  Nothing predicts a solve's iteration count well; three quantities the front end has anyway predict it a little
// (correlation 0.35 with the measured cost on the N = 16 / 50-obstacle batches): the clearance of the nearest obstacle, the
// number of LDCBF rows the presolve keeps, the robot's speed.  A scheduling hint only: every order gives the same results.
constexpr int SPLIT_BUCKETS = 16;
__device__ __forceinline__ int split_cost_bucket(double h0_min, double rows_kept, double speed) {
  const double us = 190.0 - 37.0 * fmin(fmax(h0_min, 0.0), 0.5) + 0.7 * rows_kept + 22.0 * speed;     // fitted once, in microseconds
  const int bkt = (int)((236.0 - us) * (1.0 / 4.0));                                                  // 0 = dearest
  return min(max(bkt, 0), SPLIT_BUCKETS - 1);
}
// row slots per lane of the five bodies: 1, 2, 4 in registers, 13 and 25 streamed through LDS -- every one compiles without
// scratch on its own (a 5- or 7-slot register body does not: 32 / 208 B per lane)
__host__ __device__ constexpr int split_slots(int cls) { return cls == 0 ? 1 : cls == 1 ? 2 : cls == 2 ? 4 : cls == 3 ? 13 : 25; }
__host__ __device__ constexpr int split_class_of(int need) { return need <= 1 ? 0 : need <= 2 ? 1 : need <= 4 ? 2 : need <= 13 ? 3 : 4; }
constexpr int SPLIT_MAXOBS = 50;       // obstacle slots of the split kernels' front end (every handle's n_obs_max fits)

struct KArgs {
  int N, n_obs, nvert_max, max_iter, flags, fin_rounds;
  int m_tot, words;
  double kappa, ch, sh_over_beta, inv_one_minus_ch, beta_sh;
  double l_max[2], l_min[2], v_min[2], v_max[2];
  double alpha_over_pi, omega_max, ell, tau, tol, k0_tol;
  double reach_step;      // largest CoM displacement per stage the leg-reach rows allow (presolve of the LDCBF rows)
};


}  // namespace lipmpc_dev
