// lipmpc_solve.hpp -- the solve of one problem on one group of G lanes: set-up of the lane constants here, then four sections of the same function body in their own files (row operators + factorisation, interior point, finish, outputs)
// Part of the MI355X-native batched LIP-MPC / LDCBF step solver (csrc/lipmpc_kernel.hpp includes the parts in order).
#pragma once
#include "lipmpc_front.hpp"

namespace lipmpc_dev {

// The solve of one problem on one group of G lanes, from the front end's half-spaces: NOBS_L LDCBF row slots per lane
// (obstacle slots 0 .. 2 NOBS_L - 1 of `obs`).  Output pointers may be null.
template <int G, int NOBS_L, int NVAR = G, bool LEAN = false>
__device__ __forceinline__ StepOut step_solve(
    const KArgs& P, const StepIn& in, const FrontOut<G>& F, const double (*obs)[4], const int* perm,
    double* __restrict__ U, double* __restrict__ X, double* __restrict__ obj_out, int32_t* __restrict__ status_out,
    int32_t* __restrict__ iters_out, unsigned long long* __restrict__ active_out,
    unsigned long long* __restrict__ working_out,
    double* __restrict__ diag, WarmIO* __restrict__ warm, int32_t* __restrict__ cost_out) {
  // NVAR = variable slots of the factorisation: G (every lane holds a variable: horizons up to G / 2), or 8 on a 16-lane
  // group for horizons up to 4 -- the reference's default N_horizon = 3, BASELINE config 5 -- where lanes 8..15 hold no
  // variable, their rows of K are 2I and decouple, and the factorisation / substitutions run on the leading 8 x 8 block
  static_assert(NVAR == G || (G == 16 && NVAR == 8), "variable slots: all lanes, or the first 8 of a 16-lane group");
  constexpr int NMAX = NVAR / 2;       // stages the factorisation holds (N <= NMAX, checked by the host)
  constexpr int LMAX = G / 2;          // stages by lane position (a = lane >> 1 runs up to here)
  constexpr int NV = NVAR;             // variable slots
  constexpr int GPW = 64 / G;          // groups per wavefront
  // LDCBF rows of a lane live in registers for small obstacle sets (NOBS_R of them) and are STREAMED for
  // large ones: only (s, z) per row is kept, in LDS, and every pass over the rows recomputes the rest from
  // the obstacle's (eta, b) in LDS and the stage's position — no per-row register state, no spills.
  constexpr bool STREAM = NOBS_L > 7;
  constexpr int NOBS_R = STREAM ? 0 : NOBS_L;
  constexpr int NOBS_S = STREAM ? NOBS_L : 0;
  constexpr int NR = R_CBF + NOBS_R;   // local row slots held in registers
  constexpr int MAXOBS = 2 * NOBS_L;
  constexpr int MAXWORDS = 16;         // (9*16 + 17*50 + 63)/64 = 16
  constexpr bool FUSED = (G == 16);    // one-instruction substitution / elimination steps (fmac_bcast)
  constexpr bool FUSED32 = (G == 32);  // the same on two DPP rows per problem (FactorStep32, solve32_*: row-masked chains)
  // the fused substitution chains keep 2 x 31 coefficients per lane next to the factor: only the body with two row slots per
  // lane has the registers for them (with 5 or more slots, or inside the closed-loop kernel, they spill to scratch: those
  // keep the unfused substitution)
  constexpr bool FUSED32_SOLVE = FUSED32 && NOBS_L <= 2 && !LEAN;

  __shared__ double lds_P[GPW][LMAX][2][2];                     // P_b blocks of the velocity part of K
  __shared__ unsigned long long lds_act[GPW][MAXWORDS];
  __shared__ double lds_mu[GPW][2];      // no-progress safeguard: mu of the previous iteration, sigma floor of this one
  __shared__ double lds_sz[GPW][NOBS_S > 0 ? NOBS_S : 1][G][2];   // streamed rows: (s, z) then (s, y); lane-contiguous

  PH_DECL
  const int tid = threadIdx.x;
  const int lane = tid & (G - 1);
  const int grp = tid / G;
  const bool valid = in.valid;
  const long pb = in.pb;
  const int N = P.N;
  const int a = lane >> 1;             // stage index: variable = p_{a+1}
  const int c = lane & 1;              // coordinate
  const bool var_on = a < N;
  const double sgn_a = (a & 1) ? -1.0 : 1.0;
  const double kap = P.kappa;

  const double p0x = in.p0x, v0x = in.v0x, p0y = in.p0y, v0y = in.v0y, th0 = in.th0;
  const double gx = in.gx, gy = in.gy, foot0 = in.foot0, delta = in.delta;
  const double p0c = c ? p0y : p0x, v0c = c ? v0y : v0x, gc = c ? gy : gx;

  const double th_r = F.th_r, th_v = F.th_v, om_a = F.om_a, theta1 = F.theta1, omega0 = F.omega0;
  const double s_own = F.s_own, c_own = F.c_own;
  (void)th_r; (void)th_v;
  // R(theta_a) and W(theta_{a+1}): the lane's sin / cos (front end) exchanged inside the stage
  const double s_oth = gxor<G, 1>(s_own), c_oth = gxor<G, 1>(c_own);
  const double sr = c ? s_oth : s_own, cr = c ? c_oth : c_own, sv = c ? s_own : s_oth, cv = c ? c_own : c_oth;
  const double foot_r = (a & 1) ? -foot0 : foot0;      // s_v[a]
  const double foot_v = -foot_r;                        // s_v[a+1]
  // Row vectors in OWN / PARTNER form: lane (a, c) holds coordinate c of its stage ("own") and gets the other one from
  // lane ^ 1 ("partner").  Reach row c of stage a: r = ro * d_own + rq * d_partner (R(theta) = [[cr, sr], [-sr, cr]]);
  // velocity row c: w = wo * v_own + wq * v_partner (W = [[cv, sv], [-sv, cv s]]).  The transposes use (ro, -rq) and
  // (wo, -wq).  Coefficients are ZERO on lanes without a variable (a >= N), so that every row map, transpose and K
  // block of such a lane vanishes by itself: no select on c, a or var_on is left in the solver loops.
  const double on = var_on ? 1.0 : 0.0;
  const double ro = on * cr, rq = on * (c ? -sr : sr);
  const double wo = on * (c ? cv * foot_v : cv), wq = on * (c ? -sv : sv);
  const double cm = (c == 0) ? on : 0.0;          // the manoeuvrability row lives on the c = 0 lane of its stage
  const double kap_l = on * kap;

  const int front_flag = F.front_flag;
  // the ballast row: n_d copies of 0.q <= s_bar, hosted in the manoeuvrability slot of lane 1 (a = 0, c = 1: that slot holds no
  // row there and its direction coefficient cm is zero), weighted n_d in the two sums it enters
  const double n_ball = F.n_ball, s_ball = F.s_ball;
  const bool ball = (lane == 1) && (n_ball > 0.0);
  const double ball_w = ball ? n_ball - 1.0 : 0.0;       // the slot counts once by itself

  // per-lane LDCBF rows: obstacle j = 2t + c, h = oo * p_own + oq * p_partner - ob (eta in own / partner order);
  // an absent slot is the constant row 0 . p - (-1) = 1
  double oo[NOBS_R > 0 ? NOBS_R : 1], oq[NOBS_R > 0 ? NOBS_R : 1], ob[NOBS_R > 0 ? NOBS_R : 1];
  RowFlags pres;
#pragma unroll
  for (int t = 0; t < NOBS_R; ++t) {
    const int j = 2 * t + c;
    const bool there = var_on && ((double)(a + 1) >= obs[j][3]);
    const double ex = obs[j][0], ey = obs[j][1];
    oo[t] = there ? (c ? ey : ex) : 0.0; oq[t] = there ? (c ? ex : ey) : 0.0;
    ob[t] = there ? obs[j][2] : -1.0;
    pres.set(R_CBF + t, there);
  }
  // streamed rows: presence bits, accessors
  unsigned pbits = 0u;
#pragma unroll
  for (int t = 0; t < NOBS_S; ++t)
    if (var_on && (double)(a + 1) >= obs[2 * t + c][3]) pbits |= 1u << t;
  auto s_obs = [&](int t, double& ex, double& ey, double& b) {
    const double* o = obs[2 * t + c];
    ex = o[0]; ey = o[1]; b = o[2];
  };
  auto s_pm = [&](int t) -> double { return ((pbits >> t) & 1u) ? 1.0 : 0.0; };
  pres.set(R_RU, var_on); pres.set(R_RL, var_on); pres.set(R_VU, var_on); pres.set(R_VL, var_on);
  pres.set(R_M, (var_on && (c == 0)) || ball);

  // bounds of the non-LDCBF rows
  const double hi_r = P.l_max[c], lo_r = P.l_min[c];
  const double hi_v = c ? in.vmax_y : in.vmax_x, lo_v = P.v_min[c];
  const double hi_m = in.vmax_x - in.alpha_over_pi * fabs(om_a);
  // affine parts: reach r = rr.(p_{a+1} - p_a) + (c ? s_a*ell : 0); p_0 is a constant for a = 0
  const double p0q = c ? p0x : p0y, v0q = c ? v0x : v0y;          // partner coordinate of p_0, v_0
  const double r_c = (c ? foot_r * P.ell : 0.0) - ((a == 0) ? (ro * p0c + rq * p0q) : 0.0);
  // v_{a+1} = kappa x_a + 2 kappa (-1)^a sum_{j<a} (-1)^j x_j + (-1)^{a+1} (v_0 + kappa p_0)
  const double w_c = wo * (-sgn_a * (v0c + kap * p0c)) + wq * (-sgn_a * (v0q + kap * p0q));

  int n_rows_l = __popc(pbits);
#pragma unroll
  for (int i = 0; i < NR; ++i) n_rows_l += pres[i] ? 1 : 0;
  const double m_rows = gsum<G>((double)n_rows_l) + fmax(n_ball - 1.0, 0.0);
  const double inv_m = 1.0 / fmax(m_rows, 1.0);

#include "lipmpc_solve_ops.inc"
#include "lipmpc_solve_ipm.inc"
#include "lipmpc_solve_finish.inc"
#include "lipmpc_solve_outputs.inc"
  StepOut r;
  r.status = status; r.iters = iters; r.theta1 = theta1; r.omega0 = omega0; r.obj = objv;
  r.ux = gbcast<G, 0>(u); r.uy = gbcast<G, 1>(u);
  return r;
}


}  // namespace lipmpc_dev
