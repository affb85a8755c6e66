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

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/lipmpc.h"
#pragma once

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

// ------------------------------------------------------------------------------------------
// group-level communication (G lanes, G in {16, 32}).
// G = 16: a group is exactly one DPP row, so every exchange is a VALU DPP move (no LDS crossbar):
//   broadcast of lane j      v_mov_b64_dpp row_newbcast:j
//   xor 1 / 2                quad_perm, xor 4: row_shl:4 / row_shr:4 under bank masks, xor 8: row_ror:8
//   shift by 2/4/8 stages    row_shr / row_shl with zero fill
// G = 32 (two rows): in-row steps by DPP, cross-row steps by v_permlane16_swap_b32 (no LDS crossbar either).
// ------------------------------------------------------------------------------------------
// unroll factor of the loops over streamed LDCBF rows: enough independent LDS reads in flight to cover their latency
#ifndef STREAM_UNROLL
#define STREAM_UNROLL 5
#endif
template <int I, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}
template <int I, int E, class F> __device__ __forceinline__ void static_rfor(F&& f) {   // I-1 down to E
  if constexpr (I > E) {
    f(std::integral_constant<int, I - 1>{});
    static_rfor<I - 1, E>(f);
  }
}

// The value of x, opaque to the optimiser: a lane-position predicate built from it (lane > j, lane & 16, ...) is
// recomputed where it is used (one v_cmp) instead of being hoisted out of every loop as one more 64-bit lane mask
// that lives in an SGPR pair for the whole kernel -- there are dozens of them, and they were most of the SGPR spills.
__device__ __forceinline__ int fresh(int x) {
#ifndef LIPMPC_NO_FRESH
  asm volatile("" : "+v"(x));
#endif
  return x;
}
template <int CTRL, int BANK = 0xf, class T> __device__ __forceinline__ T dpp0(T x) {     // invalid source -> 0
  return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, BANK, true);        // no 'old' operand: no zero-init move
}
template <int M, class T> __device__ __forceinline__ T row_xor(T x) {
  if constexpr (M == 1) return dpp0<0xB1>(x);                 // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return dpp0<0x4E>(x);            // quad_perm [2,3,0,1]
  else if constexpr (M == 4) {
    T r = __builtin_amdgcn_mov_dpp(x, 0x104, 0xf, 0x5, false);            // banks 0,2 <- lane+4
    return __builtin_amdgcn_update_dpp(r, x, 0x114, 0xf, 0xA, false);     // banks 1,3 <- lane-4
  } else return dpp0<0x128>(x);                               // row_ror:8
}
// G = 32: a group is two DPP rows.  v_permlane16_swap_b32 (gfx950) with both operands = v returns
// {even row's v replicated over the row pair, odd row's v replicated}: the cross-row half of every exchange,
// as a VALU instruction (no LDS crossbar).
template <class T> __device__ __forceinline__ void rowpair(T v, T& even_rep, T& odd_rep) {
  if constexpr (sizeof(T) == 8) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    even_rep = __builtin_bit_cast(T, ((unsigned long long)b[0] << 32) | a[0]);
    odd_rep = __builtin_bit_cast(T, ((unsigned long long)b[1] << 32) | a[1]);
  } else {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];        // by value: __builtin_bit_cast of a vector-element lvalue reads element 0
    even_rep = __builtin_bit_cast(T, a0);
    odd_rep = __builtin_bit_cast(T, a1);
  }
}
// the value the lane 16 positions away holds (row swap inside a 32-lane group)
template <class T> __device__ __forceinline__ T rowswap(T v) {
  T e, o;
  rowpair(v, e, o);
  const T r = (fresh(threadIdx.x) & 16) ? e : o;
  return r;
}
template <int G, int M, class T> __device__ __forceinline__ T gxor(T x) {
  if constexpr (M < 16) return row_xor<M>(x);
  else return rowswap(x);
}
// value held by lane J of the group, J a compile-time constant
template <int G, int J> __device__ __forceinline__ double gbcast(double x) {
  const double t = __builtin_amdgcn_mov_dpp(x, 0x150 + (J & 15), 0xf, 0xf, false);   // row_newbcast inside each row
  if constexpr (G == 16) return t;
  else {
    double e, o;
    rowpair(t, e, o);
    return (J < 16) ? e : o;
  }
}
// value of the lane D below / above (0 outside the group)
template <int G, int D> __device__ __forceinline__ double gup(double x, int lane_) {
  const int lane = (G == 16) ? lane_ : fresh(lane_);
  if constexpr (G == 16) return dpp0<0x110 + D>(x);
  else if constexpr (D == 16) { const double w = rowswap(x); return (lane & 16) ? w : 0.0; }
  else {
    const double t = dpp0<0x120 + D>(x);            // row_ror:D -> t[i] = x[(i - D) mod 16] of the same row
    const double w = rowswap(t);                     // the other row's rotated copy
    return ((lane & 15) >= D) ? t : ((lane & 16) ? w : 0.0);
  }
}
template <int G, int D> __device__ __forceinline__ double gdown(double x, int lane_) {
  const int lane = (G == 16) ? lane_ : fresh(lane_);
  if constexpr (G == 16) return dpp0<0x100 + D>(x);
  else if constexpr (D == 16) { const double w = rowswap(x); return (lane & 16) ? 0.0 : w; }
  else {
    const double t = dpp0<0x120 + (16 - D)>(x);     // row_ror:(16-D) -> t[i] = x[(i + D) mod 16]
    const double w = rowswap(t);
    return ((lane & 15) + D < 16) ? t : ((lane & 16) ? 0.0 : w);
  }
}

template <int G> __device__ __forceinline__ double gsum(double x) {
  x += gxor<G, 1>(x); x += gxor<G, 2>(x); x += gxor<G, 4>(x); x += gxor<G, 8>(x);
  if constexpr (G == 32) x += gxor<G, 16>(x);
  return x;
}
template <int G> __device__ __forceinline__ double gmin(double x) {
  x = fmin(x, gxor<G, 1>(x)); x = fmin(x, gxor<G, 2>(x)); x = fmin(x, gxor<G, 4>(x)); x = fmin(x, gxor<G, 8>(x));
  if constexpr (G == 32) x = fmin(x, gxor<G, 16>(x));
  return x;
}
template <int G> __device__ __forceinline__ int gmin_int(int x) {
  x = min(x, gxor<G, 1>(x)); x = min(x, gxor<G, 2>(x)); x = min(x, gxor<G, 4>(x)); x = min(x, gxor<G, 8>(x));
  if constexpr (G == 32) x = min(x, gxor<G, 16>(x));
  return x;
}
template <int G> __device__ __forceinline__ double gmax(double x) {
  x = fmax(x, gxor<G, 1>(x)); x = fmax(x, gxor<G, 2>(x)); x = fmax(x, gxor<G, 4>(x)); x = fmax(x, gxor<G, 8>(x));
  if constexpr (G == 32) x = fmax(x, gxor<G, 16>(x));
  return x;
}
// (value, index) arg-min with ties to the lower index (numpy argmin order on canonical rows)
template <int G, int M> __device__ __forceinline__ void gargmin_step(double& v, int& i) {
  const double ov = gxor<G, M>(v);
  const int oi = gxor<G, M>(i);
  const bool take = (ov < v) || (ov == v && oi < i);
  v = take ? ov : v;
  i = take ? oi : i;
}
template <int G> __device__ __forceinline__ void gargmin(double& v, int& i) {
  gargmin_step<G, 1>(v, i); gargmin_step<G, 2>(v, i); gargmin_step<G, 4>(v, i); gargmin_step<G, 8>(v, i);
  if constexpr (G == 32) gargmin_step<G, 16>(v, i);
}
// sums over earlier / later stages of the same coordinate (lane stride 2), exclusive
template <int G> __device__ __forceinline__ double prefix_excl2(double v, int lane) {
  double s = v;
  s += gup<G, 2>(s, lane); s += gup<G, 4>(s, lane); s += gup<G, 8>(s, lane);
  if constexpr (G == 32) s += gup<G, 16>(s, lane);
  return s - v;
}
template <int G> __device__ __forceinline__ double suffix_excl2(double v, int lane) {
  double s = v;
  s += gdown<G, 2>(s, lane); s += gdown<G, 4>(s, lane); s += gdown<G, 8>(s, lane);
  if constexpr (G == 32) s += gdown<G, 16>(s, lane);
  return s - v;
}
// 1/sqrt(x), 1/x to working precision from the hardware seeds (v_rsq_f64 / v_rcp_f64) + Newton
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
// v_rcp_f64 is accurate to 4.5e-8 (measured, tools/rcp_test.hip); one Newton step gives 2e-15
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, y, 1.0), y, y);
}
// reciprocal on the iteration's serial chain (step lengths, mu ratios)
__device__ __forceinline__ double chain_rcp(double x) {
#ifdef LIPMPC_IEEE_DIV
  return 1.0 / x;
#else
  return fast_rcp(x);
#endif
}

// acc + (value of `src` on lane J of the lane's DPP row) * mult in ONE instruction: v_fmac_f64 is the only FP64
// arithmetic that takes a DPP operand on gfx950 (row_newbcast only), and the compiler never folds a
// v_mov_b64_dpp into it, so it is written out.  Hazards, by hand (the compiler cannot see inside the asm):
//  * a VGPR written by a VALU instruction needs 2 wait states before a DPP read (s_nop 1 in front of every fused
//    operation whose DPP source may just have been written).  Measured on MI355X: with no wait state
//    v_add_f64 -> v_fmac_f64_dpp reads the stale value on every lane (tools/dpp_hazard_test.hip), and a chain of
//    dependent v_fmac_f64_dpp without wait states -- exact in that one-lane test -- returns garbage in the solver's
//    substitution chains, where the broadcast lane moves along the row.  No link goes without its s_nop;
//  * dpp_fence() before a sequence covers the 5 wait states after an EXEC write.
template <int J> __device__ __forceinline__ double fmac_bcast_self(double acc, double mult) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mult), "n"(J));
  return acc;
}
__device__ __forceinline__ void dpp_fence() { asm volatile("s_nop 4"); }
#include "lipmpc_fused_steps.inc"

// x where c holds, otherwise x with its high word cleared (|value| < 2^-1042, i.e. nothing once it meets a normal
// number in an FMA): ONE v_cndmask instead of the two a 64-bit select costs.  Use it on temporaries (broadcast
// results, products), where the low word needs no copy.
__device__ __forceinline__ double zero_unless(bool c, double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned hi = c ? (unsigned)(u >> 32) : 0u;
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned)u);
}
// Per-lane flags of a lane's row slots as bits of ONE register.  As `bool x[NR]` every flag is a 64-bit lane mask in
// an SGPR pair that lives across the whole kernel: with the masks of the divergent regions that made several hundred
// SGPR spills (v_writelane / v_readlane) per kernel.
// A workgroup of these kernels is exactly ONE wavefront (WAVE = 64 threads: __launch_bounds__(WAVE), the launchers
// pass dim3(WAVE), and the kernels trap on any other block size), so "barrier" means only: this wave's LDS writes are
// visible to its own later LDS reads.  A wave's LDS operations execute in order; what is left to do is keep the
// compiler from moving accesses across the point -- a workgroup-scope fence, no s_barrier.  Unlike __syncthreads()
// this is well defined inside the divergent regions it is used in (groups of a wave leave the solver loops
// independently).
// Dev instrumentation (tools/phase_cycles.py, -DLIPMPC_PHASE_TIMING variant only; such a build reports another lipmpc_version()
// and is refused by the product loader): time per section of the step, accounted PER WAVE -- a workgroup is one wave, the
// accumulators live in LDS and every marker is booked once per wave pass by the first lane that is active there, whatever
// subset of the wave's groups is still running (per-lane accumulators, as rounds 2-3 had them, charge a finished group's
// waiting time to its next marker).  Constant 100 MHz clock (wall_clock64).  Record of a wave, written to
// diag[(first problem of the wave) * 32 + k]: k < 12 ns per section, 12 + k the part of it spent with ONE group of the wave
// alive (the tail inside the wave), 24 wave lifetime ns, 25 wave lifetime in shader-clock ticks, 26 / 27 iterations / rounds
// of the wave's slowest group.  Sections: 0 iteration head (statistics, streamed pass A), 1 reciprocals + K, 2 factorisation,
// 3 predictor rhs + solve, 4 predictor rows / ratio / mu_aff, 5 corrector rhs + solve, 6 corrector rows / ratio / update,
// 7 finish: K + factorisation, 8 finish: equality solve, 9 finish: ratio test / exchange / certificate, 10 front end,
// 11 outputs.
#ifdef LIPMPC_PHASE_TIMING
constexpr int PH_WORDS = 32;
__device__ __forceinline__ void ph_mark(unsigned long long* acc, int k, int G) {
  const unsigned long long now = wall_clock64();
  const unsigned long long live = __ballot(1);
  if ((int)threadIdx.x == __ffsll((long long)live) - 1) {
    const unsigned long long dt = now - acc[31];
    acc[k] += dt;
    if (__popcll(live) <= G) acc[12 + k] += dt;
    acc[31] = now;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
#define PH_DECL __shared__ unsigned long long ph_acc_[PH_WORDS];                                                   \
  if (threadIdx.x == 0) { for (int k_ = 0; k_ < PH_WORDS; ++k_) ph_acc_[k_] = 0ull; ph_acc_[31] = in.t_start_wall; }     \
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
#define PH(k) ph_mark(ph_acc_, k, G);
#else
#define PH_DECL
#define PH(k)
#endif
constexpr int WAVE = 64;
__device__ __forceinline__ void wave_sync() {
#ifdef LIPMPC_SYNCTHREADS
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#endif
}
struct RowFlags {
  unsigned m = 0u;
  __device__ __forceinline__ bool operator[](int i) const { return (m >> i) & 1u; }
  __device__ __forceinline__ void set(int i, bool v) { m = v ? (m | (1u << i)) : (m & ~(1u << i)); }
};
// ------------------------------------------------------------------------------------------
// geometry: closest point on a convex ring, unit normal, inside flip (ObstaclesUtils.py:50-109)
// contraction off so that comparisons see the same roundings as the CPU oracle
// ------------------------------------------------------------------------------------------
struct ClosestPoint { double cx, cy, ex, ey; int degenerate; };      // returned in registers: no stack traffic for the call
// The per-edge arithmetic (two IEEE square roots and a division: ~100 dependent instructions) of EU edges runs side by side
// -- independent chains the single wave of a SIMD can overlap -- and the comparisons that pick the closest edge and count the
// crossings follow in edge order: the same operations on the same operands in the same order as the plain edge loop of the
// oracles, hence the same bits; only the latency of the chains is shared (3.3 -> 1.x us per 10 pentagons at one wave per SIMD).
// EU = 1 (the closed-loop kernel, whose register file is full): the plain loop.
template <int EU, class RingPtr>
__device__ __forceinline__ ClosestPoint closest_point_impl(RingPtr ring, int nv, double px, double py) {
#pragma clang fp contract(off)
  double best = INFINITY;
  ClosestPoint r;
  r.cx = NAN; r.cy = NAN; r.ex = 0.0; r.ey = 0.0;
  r.degenerate = 0;
  bool inside = false;
  double x0v = ring[2 * (nv - 1)], y0v = ring[2 * (nv - 1) + 1];
  bool f0 = y0v >= py;
  for (int i0 = 0; i0 < nv; i0 += EU) {
    double ax[EU], ay[EU], qx[EU], qy[EU], dd[EU], den[EU];
#pragma unroll
    for (int e = 0; e < EU; ++e) {
      const int i = (i0 + e < nv) ? i0 + e : i0;            // (an edge past the ring's end repeats edge i0: computed, never looked at)
      ax[e] = ring[2 * i]; ay[e] = ring[2 * i + 1];
      const int i1 = (i + 1 == nv) ? 0 : i + 1;
      const double bx = ring[2 * i1], by = ring[2 * i1 + 1];
      const double dx = bx - ax[e], dy = by - ay[e];
      const double nrm = sqrt(dx * dx + dy * dy);
      den[e] = nrm * nrm;                          // sqrt-then-square, ObstaclesUtils.py:81
      double t = ((px - ax[e]) * dx + (py - ay[e]) * dy) / den[e];
      t = fmax(0.0, fmin(1.0, t));
      qx[e] = ax[e] + t * dx; qy[e] = ay[e] + t * dy;
      const double ux = qx[e] - px, uy = qy[e] - py;
      dd[e] = sqrt(ux * ux + uy * uy);
    }
#pragma unroll
    for (int e = 0; e < EU; ++e) {
      if (i0 + e < nv) {
        if (den[e] == 0.0) r.degenerate = 1;
        else if (dd[e] < best) { best = dd[e]; r.cx = qx[e]; r.cy = qy[e]; }
        // crossing test of edge (ring[i-1] -> ring[i]) with the +X ray (matplotlib Path.contains_point)
        const bool f1 = ay[e] >= py;
        if (f0 != f1) {
          const bool hit = ((ay[e] - py) * (x0v - ax[e]) >= (ax[e] - px) * (y0v - ay[e])) == f1;
          if (hit) inside = !inside;
        }
        x0v = ax[e]; y0v = ay[e]; f0 = f1;
      }
    }
  }
  double nx = px - r.cx, ny = py - r.cy;
  double nn = sqrt(nx * nx + ny * ny);
  if (!(nn > 0.0)) { r.degenerate = 1; return r; }
  nx = nx / nn; ny = ny / nn;
  if (inside) { nx = -nx; ny = -ny; }
  r.ex = nx; r.ey = ny;
  return r;
}
// rings in global memory: out of line (one copy per kernel, result in registers)
__device__ __noinline__ ClosestPoint closest_point_normal(const double* __restrict__ ring, int nv, double px, double py) {
  return closest_point_impl<1>(ring, nv, px, py);
}

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

  // ---- linear row maps -------------------------------------------------------------------------
  // rows(x): lin[R_RU] = rr.(x_a - x_{a-1}); lin[R_VU] = wv.v_a(x); lin[R_CBF+t] = eta_t . x_a
  auto rows_lin = [&](double x, double& r_lin, double& w_lin, double (&h_lin)[NOBS_R > 0 ? NOBS_R : 1], double& xx,
                      double& xy) {
    const double xp = gxor<G, 1>(x);
    if constexpr (STREAM) { xx = c ? xp : x; xy = c ? x : xp; }      // streamed rows take (x, y) of the stage
    const double dxo = x - gup<G, 2>(x, lane);                        // own coordinate of p_{a+1} - p_a (p_0 is in r_c)
    r_lin = fma(rq, gxor<G, 1>(dxo), ro * dxo);
    const double ps = prefix_excl2<G>(sgn_a * x, lane);
    const double vl = kap * x + 2.0 * kap * sgn_a * ps;
    w_lin = fma(wq, gxor<G, 1>(vl), wo * vl);
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) h_lin[t] = fma(oq[t], xp, oo[t] * x);
  };
  // (G^T w)_lane from direction weights: tr (reach dir), tv (velocity dir), wc[t] (LDCBF rows, g = -eta)
  // (axs, ays): sum_t eta_t w_t over this lane's streamed rows
  // (axs, ays): sum_t eta_t w_t over this lane's streamed rows, (x, y) order
  auto GT_apply = [&](double tr, double tv, const double (&wc)[NOBS_R > 0 ? NOBS_R : 1], double axs, double ays) -> double {
    const double reach_own = fma(-rq, gxor<G, 1>(tr), ro * tr);
    double res = reach_own - gdown<G, 2>(reach_own, lane);          // lanes past the last stage hold zeros
    const double uu = fma(-wq, gxor<G, 1>(tv), wo * tv);
    const double suf = suffix_excl2<G>(sgn_a * uu, lane);
    res = fma(kap_l, uu, res);
    res = fma(2.0 * kap_l * sgn_a, suf, res);
    double aown = 0.0, apart = 0.0;                                   // eta-weighted sums for own / partner coordinate
    if constexpr (STREAM) { aown = c ? ays : axs; apart = c ? axs : ays; }
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) { aown = fma(oo[t], wc[t], aown); apart = fma(oq[t], wc[t], apart); }
    return res - (aown + gxor<G, 1>(apart));
  };

  // ---- K = 2I + G^T D G (lane = row), square-root-free factorisation, solves -----------------------
  double Krow[NV];
  double eqm[NMAX];                      // eqm[b] = 1 if this lane's stage is b
#pragma unroll
  for (int b = 0; b < NMAX; ++b) eqm[b] = (b == a) ? 1.0 : 0.0;
  // dr = d_RU + d_RL, dv = d_VU + d_VL (+ d_M), dc[t] = LDCBF row weights
  // (cxs, cxys, cys): sum_t d_t eta_t eta_t^T over this lane's streamed rows
  auto form_K = [&](double dr, double dv, const double (&dc)[NOBS_R > 0 ? NOBS_R : 1], double cxs, double cxys, double cys) {
    const double drp = gxor<G, 1>(dr);
    const double d0 = c ? drp : dr, d1 = c ? dr : drp;
    // F = Rr^T diag(d0,d1) Rr, Rr = [[cr,sr],[-sr,cr]]; this lane keeps row c
    const double F00 = cr * cr * d0 + sr * sr * d1, F01 = cr * sr * (d0 - d1), F11 = sr * sr * d0 + cr * cr * d1;
    const double Fc0 = c ? F01 : F00, Fc1 = c ? F11 : F01;
    const double dvp = gxor<G, 1>(dv);
    const double e0 = c ? dvp : dv, e1 = c ? dv : dvp;
    // E = Wv^T diag(e0,e1) Wv, Wv = [[cv,sv],[-sv,cv*s]]
    const double E00 = cv * cv * e0 + sv * sv * e1, E01 = cv * sv * e0 - sv * cv * foot_v * e1, E11 = sv * sv * e0 + cv * cv * e1;
    const double Ec0 = on * (c ? E01 : E00), Ec1 = on * (c ? E11 : E01);
    const double S0 = suffix_excl2<G>(Ec0, lane), S1 = suffix_excl2<G>(Ec1, lane);
    const double k2 = kap * kap;
    const double Pc0 = 2.0 * k2 * Ec0 + 4.0 * k2 * S0, Pc1 = 2.0 * k2 * Ec1 + 4.0 * k2 * S1;
    lds_P[grp][a][c][0] = Pc0; lds_P[grp][a][c][1] = Pc1;
    // LDCBF block sum_t d_t eta eta^T of the stage: accumulated as (own own, own partner, partner partner); the two
    // lanes of a stage hold the same three sums with own / partner swapped
    double coo = c ? cys : cxs, cop = cxys, cpp = c ? cxs : cys;
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) {
      const double do_ = dc[t] * oo[t];
      coo = fma(do_, oo[t], coo); cop = fma(do_, oq[t], cop); cpp = fma(dc[t] * oq[t], oq[t], cpp);
    }
    coo += gxor<G, 1>(cpp); cop += gxor<G, 1>(cop);
    const double Cc0 = c ? cop : coo, Cc1 = c ? coo : cop;
    // F_{a+1}, row c (0 past the last stage: lanes without rows have d = 0, hence F = 0)
    const double Fn0 = gdown<G, 2>(Fc0, lane), Fn1 = gdown<G, 2>(Fc1, lane);
    const double Dg0 = (c ? 0.0 : 2.0) - k2 * Ec0 + Fc0 + Fn0 + Cc0;
    const double Dg1 = (c ? 2.0 : 0.0) - k2 * Ec1 + Fc1 + Fn1 + Cc1;
    wave_sync();
    // block b of the row: (-1)^(a+b) P_max(a,b) + [b==a] Dg - [b==a-1] F_a - [b==a+1] F_{a+1}; the three
    // indicator terms are FMAs against 0/1 masks (eqm), not selects.  Lanes/stages beyond N fall
    // out as rows of 2I because all their weights are zero.
#pragma unroll
    for (int b = 0; b < NMAX; ++b) {
      const bool own = (b <= a);
      const double q0v = own ? Pc0 : lds_P[grp][b][c][0];
      const double q1v = own ? Pc1 : lds_P[grp][b][c][1];
      const double sg = sgn_a * ((b & 1) ? -1.0 : 1.0);
      double k0 = fma(eqm[b], Dg0, sg * q0v), k1 = fma(eqm[b], Dg1, sg * q1v);
      if (b + 1 < NMAX) { k0 = fma(-eqm[b + 1], Fc0, k0); k1 = fma(-eqm[b + 1], Fc1, k1); }
      if (b >= 1) { k0 = fma(-eqm[b - 1], Fn0, k0); k1 = fma(-eqm[b - 1], Fn1, k1); }
      Krow[2 * b] = k0;
      Krow[2 * b + 1] = k1;
    }
    wave_sync();
  };
  // Square-root-free right-looking factorisation K = Lt D^-1 Lt^T in full symmetric storage, one
  // row per lane, nothing rescaled: after step j lane j keeps row j of the Schur complement
  // (= column j of Lt, by symmetry) and lanes l > j keep Lt[l][j] in Krow[j]; ipiv = 1/pivot.
  // Per (j, c) that is one row_newbcast move and one FMA, with f = 0 on lanes <= j instead of
  // predication.  Returns false on a non-positive pivot.
  double ipiv = 0.5;
  // broadcast inside the lane's own DPP row (16 lanes)
  auto bc16 = [](auto ic, double x) -> double {
    return __builtin_amdgcn_mov_dpp(x, 0x150 + decltype(ic)::value, 0xf, 0xf, false);
  };
  // G = 16: the triangular factors in the form the fused substitution steps want them (solve):
  //   Xl[j] = -Lt[l][j] / p_j on lanes l > j, 0 elsewhere;  Yu[j] = -Lt[j][l] / p_l ... = -ipiv_l S_l[j] on lanes l < j, 0 elsewhere
  double Xl[(FUSED || FUSED32_SOLVE) ? NV : 1], Yu[(FUSED || FUSED32_SOLVE) ? NV : 1];
  auto factor = [&]() -> bool {
    bool ok = true;
    const int ln = fresh(lane);
    dpp_fence();
    static_for<0, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (FUSED) {
        // One DPP row per problem: pivot chain (broadcast, reciprocal, one Newton step, scaled column), then the NV - 1 - j
        // updates of the step as one asm statement.  (Issuing the updates BETWEEN the links of the next step's pivot chain --
        // "lookahead", tried in round 4 -- gains nothing: a dependent FP64 instruction costs 7 cycles against 4.6 for an
        // independent one, the chain is issue slots, not latency; profiles/r04_dev_tools/r04_iter_cost_lookahead_factorisation.txt.)
        const double pj = gbcast<G, j>(Krow[j]);
        ok = ok && (pj > 0.0);
        const double ip = fast_rcp(pj);
        const double nf = zero_unless(ln > j, Krow[j] * -ip);
        ipiv = (ln == j) ? ip : ipiv;
        Xl[j] = nf;
        FactorStep<NV, j>::run(Krow, nf);
      } else {
        // Two DPP rows per problem.  Row j of the Schur complement equals its column j, and the column is lane-distributed
        // (lane cc holds S[cc][j] in Krow[j]): ONE cross-row exchange per step makes both 16-lane halves of the column
        // visible in every row, after which each update S[l][cc] -= (S[l][j] / p_j) S[cc][j] is one v_fmac_f64_dpp with the
        // broadcast of S[cc][j] as its DPP operand.  (LDL^T form: a Cholesky-form update g_l g_cc with g = S[.][j] / sqrt(p_j)
        // keeps the two triangles bit-identical but breaks down -- pivot <= 0 -- on 3 % of the N = 16 / 50-obstacle problems
        // where this form does not.)
        double cA, cB;                                    // S[0..15][j], S[16..31][j] by local lane position
        rowpair(Krow[j], cA, cB);
        const double pj = bc16(std::integral_constant<int, (j & 15)>{}, j < 16 ? cA : cB);
        ok = ok && (pj > 0.0);
        const double ip = fast_rcp(pj);
        ipiv = (ln == j) ? ip : ipiv;
        const double ng = zero_unless(ln > j, Krow[j] * -ip);
        if constexpr (FUSED32_SOLVE) Xl[j] = ng;
        FactorStep32<j>::run(Krow, cA, cB, ng);
      }
    });
    if constexpr (FUSED || FUSED32_SOLVE) {
      const double nip = -ipiv;
      static_for<1, NV>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        Yu[j] = zero_unless(fresh(lane) < j, Krow[j] * nip);
      });
    }
    return ok;
  };
  auto solve = [&](double b) -> double {
    const int ln = fresh(lane);
    if constexpr (FUSED) {
      // forward: b_l += Xl_l[j] b_j (lanes l > j); lane j's b is final after step j-1.  backward on x = w - ipiv acc:
      // x_l += Yu_l[j] x_j (lanes l < j), lane j final once the columns above it are done.  One instruction per step.
      dpp_fence();
      b = solve_forward_chain(b, Xl);
      return solve_backward_chain(b * ipiv, Yu);
    } else if constexpr (FUSED32_SOLVE) {
      // The same recurrences on two DPP rows, the substitution chain inside one row at a time (row_mask): columns 0..15
      // among the lanes of the low row, the high row catches up on those 16 columns after ONE cross-row exchange of the
      // finished values, columns 16..31 inside the high row; and the mirror image backwards.  Xl / Yu hold the
      // coefficients by group lane position (zero where a link does not apply), so a link is one v_fmac_f64_dpp.
      dpp_fence();
      b = solve32_fwd_lo(b, Xl);
      double lo_rep, hi_rep;
      rowpair(b, lo_rep, hi_rep);                         // lo_rep: the low row's b_0..b_15 by local lane position
      b = solve32_fwd_x(b, lo_rep, Xl);
      b = solve32_fwd_hi(b, Xl);
      double x = b * ipiv;
      x = solve32_bwd_hi(x, Yu);
      rowpair(x, lo_rep, hi_rep);                         // hi_rep: x_16..x_31
      x = solve32_bwd_x(x, hi_rep, Yu);
      return solve32_bwd_lo(x, Yu);
    } else {
      // Same recurrences, organised so that the substitution chain stays inside one DPP row at a time:
      // columns 0..15 are eliminated among the lanes of row 0, the lanes of row 1 catch up on those 16 columns
      // after ONE cross-row exchange of the finished w_0..w_15, then columns 16..31 run inside row 1
      // (and the mirror image backwards).
      const bool hi = (ln & 16) != 0;
      static_for<0, 16>([&](auto ic) {
        constexpr int j = decltype(ic)::value;
        const double wj = zero_unless(!hi && ln > j, bc16(ic, b * ipiv));
        b = fma(-Krow[j], wj, b);
      });
      {
        double wA, wB;
        rowpair(b * ipiv, wA, wB);                        // wA: w_0..w_15 by local lane position
        static_for<0, 16>([&](auto ic) {
          constexpr int j = decltype(ic)::value;
          b = fma(-Krow[j], zero_unless(hi, bc16(ic, wA)), b);
        });
      }
      static_for<0, 16>([&](auto ic) {
        constexpr int j = 16 + decltype(ic)::value;
        const double wj = zero_unless(hi && ln > j, bc16(ic, b * ipiv));
        b = fma(-Krow[j], wj, b);
      });
      const double w = b * ipiv;
      double acc = 0.0;
      static_rfor<16, 0>([&](auto ic) {
        constexpr int j = 16 + decltype(ic)::value;
        const double xj = zero_unless(hi && ln < j, bc16(ic, fma(-ipiv, acc, w)));
        acc = fma(Krow[j], xj, acc);
      });
      {
        double xA, xB;
        rowpair(fma(-ipiv, acc, w), xA, xB);              // xB: x_16..x_31 by local lane position
        static_for<0, 16>([&](auto ic) {
          constexpr int j = 16 + decltype(ic)::value;
          acc = fma(Krow[j], zero_unless(!hi, bc16(ic, xB)), acc);
        });
      }
      static_rfor<16, 0>([&](auto ic) {
        constexpr int j = decltype(ic)::value;
        const double xj = zero_unless(!hi && ln < j, bc16(ic, fma(-ipiv, acc, w)));
        acc = fma(Krow[j], xj, acc);
      });
      return fma(-ipiv, acc, w);
    }
  };

  // ---- interior point ---------------------------------------------------------------------------
  const bool warm_on = warm != nullptr && warm->lds != nullptr;         // wave-uniform
  const bool warm_in = warm_on && warm->have;
  const bool last_stage = a >= N - 1;
  const int wsrc = last_stage ? lane : lane + 2;                          // stage a takes over stage a + 1; the last keeps its own
  double q = var_on ? p0c : 0.0;
  if (warm_in) {
    const double qo = warm->lds[lane], qn = warm->lds[wsrc], qb = warm->lds[lane >= 2 ? lane - 2 : lane];
    q = var_on ? (last_stage ? qo + (qo - qb) : qn) : 0.0;              // the last stage extrapolates one more step
  }
  double s[NR], z[NR], slk[NR];     // slack variable, multiplier, slack function value h - g.q
  double hl[NOBS_R > 0 ? NOBS_R : 1];
  double cx_ = 0.0, cy_ = 0.0;       // stage position / direction of the last rows_lin call, (x, y) order (streamed rows only)
  auto slack_values = [&](double x) {
    double r_lin, w_lin;
    rows_lin(x, r_lin, w_lin, hl, cx_, cy_);
    const double r = r_lin + r_c, w = w_lin + w_c;
    slk[R_RU] = hi_r - r; slk[R_RL] = r - lo_r;
    slk[R_VU] = hi_v - w; slk[R_VL] = w - lo_v;
    slk[R_M] = hi_m - w;
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) slk[R_CBF + t] = hl[t] - ob[t];
  };
  // g_i . dx for every local row from the linear maps
  auto rows_dir = [&](double dx, double (&dl)[NR]) {
    double r_lin, w_lin;
    rows_lin(dx, r_lin, w_lin, hl, cx_, cy_);
    dl[R_RU] = r_lin; dl[R_RL] = -r_lin; dl[R_VU] = w_lin; dl[R_VL] = -w_lin; dl[R_M] = cm * w_lin;
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) dl[R_CBF + t] = -hl[t];
  };
  auto GT_rows = [&](const double (&w)[NR], double axs, double ays) -> double {
    double wc[NOBS_R > 0 ? NOBS_R : 1];
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) wc[t] = w[R_CBF + t];
    return GT_apply(w[R_RU] - w[R_RL], fma(cm, w[R_M], w[R_VU] - w[R_VL]), wc, axs, ays);      // (cm: not the ballast row's slot)
  };
  auto K_rows = [&](const double (&d)[NR], double cxs, double cxys, double cys) {
    double dc[NOBS_R > 0 ? NOBS_R : 1];
#pragma unroll
    for (int t = 0; t < NOBS_R; ++t) dc[t] = d[R_CBF + t];
    form_K(d[R_RU] + d[R_RL], fma(cm, d[R_M], d[R_VU] + d[R_VL]), dc, cxs, cxys, cys);
  };

  slack_values(q);
  slk[R_M] = ball ? s_ball : slk[R_M];
  // An absent row (empty obstacle slot, manoeuvrability on the c = 1 lane, lane without a variable) is the constant
  // row 0 . q <= 1 with s = slk = 1, z = 0: its direction coefficients are zero, so r_p, ds, dz and its weights stay
  // exactly zero through the iteration with ONE masked quantity, 1/s (below), instead of a mask on every product.
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    slk[i] = pres[i] ? slk[i] : 1.0;
    s[i] = pres[i] ? fmax(slk[i], IPM_S_FLOOR) : 1.0;
    z[i] = pres[i] ? IPM_Z0 : 1e-300;         // not 0: 1/z stays finite without a guard (a multiplier never reaches 0:
                                              // every step keeps at least 0.005 of it)
    if (warm_in && i < WARM_ROWS) z[i] = pres[i] ? fmin(fmax(warm->lds[(1 + i) * G + wsrc], WARM_Z_MIN), WARM_Z_MAX) : 1e-300;
  }
  if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
    for (int t = 0; t < NOBS_S; ++t) {
      double ex, ey, b;
      s_obs(t, ex, ey, b);
      const bool on = (pbits >> t) & 1u;
      lds_sz[grp][t][lane][0] = on ? fmax(ex * cx_ + ey * cy_ - b, IPM_S_FLOOR) : 1.0;
      lds_sz[grp][t][lane][1] = on ? IPM_Z0 : 0.0;
    }
  }
  // One streamed row as every pass sees it: (s, z) from LDS, the rest recomputed from (eta, b) and the
  // stage position (px, py) of the current iterate.
  struct SRow { double s, z, rp, is, d, ex, ey, pm; };
  auto s_row = [&](int t, double px, double py) -> SRow {
    SRow r;
    double b;
    s_obs(t, r.ex, r.ey, b);
    r.pm = s_pm(t);
    r.s = lds_sz[grp][t][lane][0]; r.z = lds_sz[grp][t][lane][1];
    r.rp = (r.s - (r.ex * px + r.ey * py - b)) * r.pm;
    r.is = fast_rcp(r.s);
    r.d = r.z * r.is;
    return r;
  };

  int status = LIPMPC_STATUS_MAX_ITER;
  int iters = 0;
  bool done = false;
  if (front_flag & 2) { status = LIPMPC_STATUS_DEGENERATE; done = true; }
  else if (front_flag & 1) { status = LIPMPC_STATUS_INFEASIBLE; done = true; }
  if (in.sensor_overflow) { status = LIPMPC_STATUS_SENSOR_OVERFLOW; done = true; }     // a truncated obstacle list is not planned against
  if (front_flag & 8) { status = LIPMPC_STATUS_MAX_ITER; done = true; }                // (never: the body was chosen by this very count)
  if (m_rows == 0.0 && !done) { status = LIPMPC_STATUS_SOLVED; done = true; q = var_on ? gc : 0.0; }

  // row-presence masks as 0/1 doubles: an absent row keeps s = 1, z = 0 and is neutralised by four
  // multiplies per iteration instead of predicated selects everywhere
  double pm[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) pm[i] = pres[i] ? 2.0 : 1.0;      // the Newton constant of 1/s (below)

  double rp[NR];                     // primal residual s - (h - g.q) of every row, carried through the iterations
#pragma unroll
  for (int i = 0; i < NR; ++i) rp[i] = s[i] - slk[i];
  // Groups of a wave leave the loop independently (real divergence: a finished group's lanes are
  // simply masked off; all exchanges inside are row-local DPP / group-local LDS).
  if (lane == 0) { lds_mu[grp][0] = INFINITY; lds_mu[grp][1] = 0.0; }
  PH(10)
  for (int it = 0; it <= P.max_iter; ++it) {
    if (__all(done)) break;
    if (!done) {
      double w[NR], d[NR];
      double mu_l = 0.0, rpmax_l = 0.0, zmax_l = 0.0;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        mu_l = fma(s[i], z[i], mu_l);
        rpmax_l = fmax(rpmax_l, fabs(rp[i]));
        zmax_l = fmax(zmax_l, z[i]);
      }
      mu_l = fma(ball_w * s[R_M], z[R_M], mu_l);           // the other n_d - 1 copies of the ballast row
      // streamed rows, pass A: residual statistics, K blocks and predictor weights in one sweep
      double qx = 0.0, qy = 0.0, cxs = 0.0, cxys = 0.0, cys = 0.0, axs = 0.0, ays = 0.0;
      if constexpr (STREAM) {
        const double qp = gxor<G, 1>(q);
        qx = c ? qp : q; qy = c ? q : qp;
#pragma unroll STREAM_UNROLL
        for (int t = 0; t < NOBS_S; ++t) {
          const SRow r = s_row(t, qx, qy);
          mu_l = fma(r.s, r.z, mu_l);
          rpmax_l = fmax(rpmax_l, fabs(r.rp));
          zmax_l = fmax(zmax_l, r.z);
          cxs = fma(r.d * r.ex, r.ex, cxs); cxys = fma(r.d * r.ex, r.ey, cxys); cys = fma(r.d * r.ey, r.ey, cys);
          const double wt = fma(r.d, r.rp - r.s, r.z);            // z + d (rp - s)
          axs = fma(r.ex, wt, axs); ays = fma(r.ey, wt, ays);
        }
      }
      const double musum = gsum<G>(mu_l);
      const double mu = musum * inv_m;
      {   // no-progress safeguard, a ramp in mu / mu_prev (oracle/lipmpc_oracle.py).  Both values live in LDS: one more
          // double kept in registers across the factorisation costs 5 % of the iteration in AGPR traffic.
        const double mu_prev = lds_mu[grp][0];
        // (hardware reciprocal seed, 4.5e-8: the ramp is continuous, so that is as good as a division here)
        const double ramp = fmin(1.0, fmax(0.0, (mu * __builtin_amdgcn_rcp(mu_prev) - IPM_SLOW_RATIO) * (1.0 / (1.0 - IPM_SLOW_RATIO))));
        if (lane == 0) { lds_mu[grp][0] = mu; lds_mu[grp][1] = (it >= IPM_SLOW_FROM) ? IPM_SLOW_SIGMA * ramp : 0.0; }
      }
      // largest primal residual and the divergence test (z or |q| out of range, NaN included) in ONE group reduction: a
      // lane that sees divergence contributes +inf
      const double zq_l = fmax(zmax_l * (1.0 / IPM_Z_DIVERGE), fabs(q) * 1e-300);        // >= 1: diverged
      const double rpmax = gmax<G>(!(zq_l < 1.0) ? INFINITY : rpmax_l);
      const bool bad = !(rpmax < INFINITY);
      if (rpmax <= P.tol && mu <= P.tol) { status = LIPMPC_STATUS_SOLVED; done = true; iters = it; }
      else if (it == P.max_iter) { done = true; iters = it; }
      else if (bad) { status = LIPMPC_STATUS_INFEASIBLE; done = true; iters = it; }
      PH(0)
      if (!done) {
        // Reciprocals once per row and iteration; every later division becomes a multiply, and the
        // ratio tests run on -ds/s, -dz/z (largest ratio r => step 1/r) so they need no division.
        // 1/z only feeds a ratio test: the 4.5e-8-accurate hardware seed is enough there.
        double is_[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          // 1/s by one Newton step on the hardware seed, y (c1 - s y) with c1 = 2; an absent row has s = 1 exactly
          // (seed exact) and c1 = 1, which makes its 1/s exactly 0: d, w, dz of that row vanish at no extra cost
          const double y0 = __builtin_amdgcn_rcp(s[i]);
          is_[i] = y0 * fma(-s[i], y0, pm[i]);
          d[i] = z[i] * is_[i];
        }
        K_rows(d, cxs, cxys, cys);
        PH(1)
        const bool fok = factor();
        PH(2)
        if (!fok) {
          // K loses numerical definiteness once max(z/s) ~ 1e15: near the solution that is
          // "converged to working precision" (the finish takes over), elsewhere infeasibility
          status = (rpmax <= IPM_STALL_TOL && mu <= IPM_STALL_TOL) ? LIPMPC_STATUS_SOLVED : LIPMPC_STATUS_INFEASIBLE;
          done = true; iters = it;
        }
        // right-hand sides  -r_d - G^T w = -2(q - g) - G^T (z + w): one transpose apply per solve.
        // predictor: rc = s z  ->  w = z (rp - s) / s
        const double m2qg = var_on ? -2.0 * (q - gc) : 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) w[i] = fma(d[i], rp[i] - s[i], z[i]);
        const double dqa = solve(m2qg - GT_rows(w, axs, ays));
        PH(3)
        double dl[NR], c2[NR];                             // c2 = ds_aff dz_aff: all the corrector needs of the predictor
        rows_dir(dqa, dl);
        const double ax_ = cx_, ay_ = cy_;                  // predictor direction of this stage (streamed rows)
        double r_l = 1.0;                                  // largest of 1, -ds/s, -dz/z
        double s2_l = 0.0;                                 // sum ds dz: mu_aff = ((1 - a) sum(s z) + a^2 sum(ds dz)) / m,
                                                           // because s dz + z ds = -s z holds row by row for the predictor
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const double dsa = -rp[i] - dl[i];
          const double t = dsa * is_[i];                    // ds/s; and -dz/z = (s + ds)/s = 1 + t, no 1/z needed
          const double dza = -fma(d[i], dsa, z[i]);         // -(s z + z ds)/s
          r_l = fmax(r_l, fmax(-t, 1.0 + t));
          c2[i] = dsa * dza;
          s2_l += c2[i];
        }
        s2_l = fma(ball_w, c2[R_M], s2_l);
        if constexpr (STREAM) {                            // pass B
#pragma unroll STREAM_UNROLL
          for (int t = 0; t < NOBS_S; ++t) {
            const SRow r = s_row(t, qx, qy);
            const double dsa_t = (-r.rp + (r.ex * ax_ + r.ey * ay_)) * r.pm;      // g = -eta
            const double dza_t = -r.d * (r.s + dsa_t);
            r_l = fmax(r_l, fmax(-dsa_t * r.is, -dza_t * __builtin_amdgcn_rcp(fmax(r.z, 1e-300))));
            s2_l = fma(dsa_t, dza_t, s2_l);
          }
        }
        // reciprocals to 2e-15 (v_rcp_f64 + one Newton step) instead of IEEE divisions: five of them sat on the
        // iteration's serial chain at ~10 dependent instructions each
        const double a_aff = chain_rcp(gmax<G>(r_l));
        const double mu_aff = fma(a_aff * a_aff, gsum<G>(s2_l), (1.0 - a_aff) * musum) * inv_m;
        const double ratio = mu_aff * chain_rcp(mu);
        double sigma = ratio * ratio * ratio;
        sigma = fmax(sigma, lds_mu[grp][1]);      // no-progress safeguard: floor computed at the top of the iteration
        const double sigma_mu = sigma * mu;
        PH(4)
        // corrector: rc = s z + ds_a dz_a - sigma mu
        double rc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          rc[i] = fma(s[i], z[i], c2[i]) - sigma_mu;
          w[i] = fma(fma(z[i], rp[i], -rc[i]), is_[i], z[i]);
        }
        axs = 0.0; ays = 0.0;
        if constexpr (STREAM) {                            // pass D
#pragma unroll STREAM_UNROLL
          for (int t = 0; t < NOBS_S; ++t) {
            const SRow r = s_row(t, qx, qy);
            const double dsa_t = (-r.rp + (r.ex * ax_ + r.ey * ay_)) * r.pm;
            const double dza_t = -r.d * (r.s + dsa_t);
            const double rc_t = (fma(r.s, r.z, dsa_t * dza_t) - sigma_mu) * r.pm;
            const double wt = fma(fma(r.z, r.rp, -rc_t), r.is, r.z);
            axs = fma(r.ex, wt, axs); ays = fma(r.ey, wt, ays);
          }
        }
        const double dq = solve(m2qg - GT_rows(w, axs, ays));
        PH(5)
        rows_dir(dq, dl);
        const double bx_ = cx_, by_ = cy_;                  // corrector direction of this stage
        r_l = IPM_STEP_FRAC;                               // alpha = min(1, 0.995 / max ratio)
        double ds[NR], dz[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          ds[i] = -rp[i] - dl[i];
          dz[i] = -fma(z[i], ds[i], rc[i]) * is_[i];
          // 1/z only feeds this ratio test: the 4.5e-8-accurate hardware seed is enough
          r_l = fmax(r_l, fmax(-ds[i] * is_[i], -dz[i] * __builtin_amdgcn_rcp(z[i])));
        }
        // streamed rows: (ds, dz) of the combined direction, recomputed identically in passes E and F
        auto s_step = [&](const SRow& r, double& ds_t, double& dz_t) {
          const double dsa_t = (-r.rp + (r.ex * ax_ + r.ey * ay_)) * r.pm;
          const double dza_t = -r.d * (r.s + dsa_t);
          const double rc_t = (fma(r.s, r.z, dsa_t * dza_t) - sigma_mu) * r.pm;
          ds_t = (-r.rp + (r.ex * bx_ + r.ey * by_)) * r.pm;
          dz_t = -fma(r.z, ds_t, rc_t) * r.is;
        };
        if constexpr (STREAM) {                            // pass E
#pragma unroll STREAM_UNROLL
          for (int t = 0; t < NOBS_S; ++t) {
            const SRow r = s_row(t, qx, qy);
            double ds_t, dz_t;
            s_step(r, ds_t, dz_t);
            r_l = fmax(r_l, fmax(-ds_t * r.is, -dz_t * __builtin_amdgcn_rcp(fmax(r.z, 1e-300))));
          }
        }
        const double alpha = IPM_STEP_FRAC * chain_rcp(gmax<G>(r_l));
        if (!done) {
          if constexpr (STREAM) {                          // pass F (before q moves: rows are evaluated at the old iterate)
#pragma unroll STREAM_UNROLL
            for (int t = 0; t < NOBS_S; ++t) {
              const SRow r = s_row(t, qx, qy);
              double ds_t, dz_t;
              s_step(r, ds_t, dz_t);
              lds_sz[grp][t][lane][0] = fma(alpha, ds_t, r.s);
              lds_sz[grp][t][lane][1] = fma(alpha, dz_t, r.z);
            }
          }
          q = fma(alpha, dq, q);
          // the primal residual of a row shrinks by exactly 1 - alpha along a Newton step (ds + g.dq = -rp), so it is
          // carried by that recurrence instead of being re-formed from a slack value kept per row (one persistent double
          // per row less; the finish recomputes the slack functions from q)
          const double oma = 1.0 - alpha;
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            s[i] = fma(alpha, ds[i], s[i]); z[i] = fma(alpha, dz[i], z[i]); rp[i] *= oma;
          }
        }
        PH(6)
      }
    }
  }

  if (ball) { pres.set(R_M, false); s[R_M] = 1.0; z[R_M] = 0.0; }      // the ballast row ends with the interior-point phase
  if (warm_on) {                              // park this step's interior-point result for the next step
    wave_sync();                              // (every lane has read its neighbours' previous values by now)
    warm->lds[lane] = q;
#pragma unroll
    for (int i = 0; i < NR && i < WARM_ROWS; ++i) warm->lds[(1 + i) * G + lane] = z[i];
    wave_sync();
  }
  // canonical row index (include/lipmpc.h) of a local row slot / of streamed row t
  auto ci_of = [&](int i) -> int {
    if (i == R_RU) return 4 * a + c;
    if (i == R_RL) return 4 * a + 2 + c;
    if (i == R_VU) return 5 * N + 4 * a + c;
    if (i == R_VL) return 5 * N + 4 * a + 2 + c;
    if (i == R_M) return 4 * N + a;
    return 9 * N + (a + 1) * P.n_obs + perm[2 * (i - R_CBF) + c];
  };
  auto ci_s = [&](int t) -> int { return 9 * N + (a + 1) * P.n_obs + perm[2 * t + c]; };

  // diagnostics: identification margin min |log(z/(1e5 s))| and final mu of the interior-point phase;
  // initial working set z > 1e5 s
  // (the logarithms cost ~2 us per wave: only when the caller asked for diag)
  double marg_l = INFINITY;
  const bool want_diag = diag != nullptr;
  RowFlags act;
  unsigned abits = 0u;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (want_diag && pres[i]) marg_l = fmin(marg_l, fabs(log(z[i] / (FIN_IDENT * s[i]))));
    act.set(i, pres[i] && (z[i] > FIN_IDENT * s[i]));
  }
  if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
    for (int t = 0; t < NOBS_S; ++t) {
      if ((pbits >> t) & 1u) {
        const double st = lds_sz[grp][t][lane][0], zt = lds_sz[grp][t][lane][1];
        if (want_diag) marg_l = fmin(marg_l, fabs(log(zt / (FIN_IDENT * st))));
        if (zt > FIN_IDENT * st) abits |= 1u << t;
      }
    }
  }
  const unsigned fbits = abits;          // fallback working set of an uncertified finish
  const double margin = gmin<G>(marg_l);
  double diag_rounds = 0.0, diag_eres = 0.0, diag_cert = 0.0;

  // ---- certified active-set finish --------------------------------------------------------------
  // A PRIMAL active-set method from the interior-point iterate (oracle: finish_active_set): per round one factorisation of
  // K_A = 2I + rho G_A^T G_A, the minimiser x_A on the working set by the method of multipliers, a ratio test along
  // d = x_A - x over the rows outside A (a blocked step stops at the blocking row, which joins A), otherwise the rows
  // with a negative multiplier leave A, otherwise x_A is the optimum.  x stays feasible, the objective never increases, and a
  // blocking row is never dependent on A (g.d = 0 for every row in A's span), which is what the degenerate vertices of
  // the long-horizon / many-obstacle problems need.
  const bool ipm_ok = (status == LIPMPC_STATUS_SOLVED) && (m_rows > 0.0);
  if (!(P.flags & LIPMPC_FLAG_INTERIOR)) {
    bool fin_done = !ipm_ok;        // groups that never converged skip the finish
    bool certified = false;
    double rho = FIN_RHO;           // penalty of the equality solves (FIN_RHO_POLISH after a polish request, below)
    double xf = q;                  // the feasible point the rounds move
    double qf = q;                  // minimiser on the working set
    double y[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) y[i] = act[i] ? z[i] : 0.0;
    if constexpr (STREAM) {           // the z slot of a streamed row now holds its multiplier y
#pragma unroll STREAM_UNROLL
      for (int t = 0; t < NOBS_S; ++t)
        if (!((abits >> t) & 1u)) lds_sz[grp][t][lane][1] = 0.0;
    }
    // slack function of streamed row t at the stage position / its ratio-test entry: evaluated twice per round (value,
    // then index of the minimum), so both evaluations must round alike
    auto s_slack = [&](int t, double fx, double fy) -> double {
#pragma clang fp contract(off)
      double ex, ey, bb;
      s_obs(t, ex, ey, bb);
      return ex * fx + ey * fy - bb;
    };
    auto s_ratio = [&](int t, double fx, double fy, double ddx, double ddy) -> double {
#pragma clang fp contract(off)
      double ex, ey, bb;
      s_obs(t, ex, ey, bb);
      const double sl = ex * fx + ey * fy - bb, gd = -(ex * ddx + ey * ddy);       // g = -eta
      const bool cand = gd > FIN_GD_MIN;
      return cand ? fmax(sl + gd, 0.0) * fast_rcp(gd) : INFINITY;
    };
    for (int rnd = 0; rnd < P.fin_rounds; ++rnd) {
      if (__all(fin_done)) break;
      double d[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) d[i] = act[i] ? rho : 0.0;
      double cxs = 0.0, cxys = 0.0, cys = 0.0;
      if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
        for (int t = 0; t < NOBS_S; ++t) {
          if ((abits >> t) & 1u) {
            double ex, ey, bb;
            s_obs(t, ex, ey, bb);
            cxs = fma(rho * ex, ex, cxs); cxys = fma(rho * ex, ey, cxys); cys = fma(rho * ey, ey, cys);
          }
        }
      }
      PH(9)
      K_rows(d, cxs, cxys, cys);
      const bool fok = factor();
      PH(7)
      qf = xf;                                         // the equality solve starts at the current point
      double eres = INFINITY, rd_g = 0.0, r_g = 0.0;
      for (int in = 0; in <= FIN_INNER; ++in) {
        slack_values(qf);
        const double fx = cx_, fy = cy_;               // stage position of qf
        double wr[NR], rmax_l = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const double r = act[i] ? -slk[i] : 0.0;      // G_A q - h_A
          wr[i] = -d[i] * slk[i];                       // rho r (d = rho on active rows, 0 elsewhere)
          rmax_l = fmax(rmax_l, fabs(r));
        }
        double ayx = 0.0, ayy = 0.0, awx = 0.0, awy = 0.0;
        if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
          for (int t = 0; t < NOBS_S; ++t) {
            if ((abits >> t) & 1u) {
              double ex, ey, bb;
              s_obs(t, ex, ey, bb);
              const double r = -(ex * fx + ey * fy - bb);
              const double yt = lds_sz[grp][t][lane][1];
              rmax_l = fmax(rmax_l, fabs(r));
              ayx = fma(ex, yt, ayx); ayy = fma(ey, yt, ayy);
              awx = fma(ex, rho * r, awx); awy = fma(ey, rho * r, awy);
            }
          }
        }
        const double gty = GT_rows(y, ayx, ayy);
        const double rd = var_on ? (2.0 * (qf - gc) + gty) : 0.0;
        const double eprev = eres;
        rd_g = gmax<G>(fabs(rd));
        r_g = gmax<G>(rmax_l);
        eres = fmax(rd_g, r_g);
        // converged, out of corrections, or stalled on its rounding floor below what the certificate needs (an
        // ill-conditioned working set sits at 1e-10 forever: four corrections of ~1.7 us each, per round, for nothing)
        const bool stop = (eres <= FIN_INNER_TOL) || (in == FIN_INNER) || (eres <= FIN_EPS && eres > FIN_STALL * eprev);
        if (__all(stop || fin_done)) break;
        const double dq = solve(-rd - GT_rows(wr, awx, awy));
        double dl[NR];
        rows_dir(dq, dl);
        if (!stop && !fin_done) {
          qf += dq;
#pragma unroll
          for (int i = 0; i < NR; ++i) y[i] = fma(d[i], dl[i] - slk[i], y[i]);     // rho on active rows only
          if constexpr (STREAM) {
            const double ddx = cx_, ddy = cy_;
#pragma unroll STREAM_UNROLL
            for (int t = 0; t < NOBS_S; ++t) {
              if ((abits >> t) & 1u) {
                double ex, ey, bb;
                s_obs(t, ex, ey, bb);
                const double slk_t = ex * fx + ey * fy - bb, dl_t = -(ex * ddx + ey * ddy);
                lds_sz[grp][t][lane][1] += rho * (dl_t - slk_t);
              }
            }
          }
        }
      }
      PH(8)
      // (slk holds the slack functions of the final qf: every pass of the loop above evaluates them before it decides to stop)
      // ratio test along d = qf - xf: slack at x = slack at qf + g.d; entry of a row outside A the direction runs into:
      // max(slack at x, 0) / g.d; the smallest entry below 1 blocks the step
      const double fx = cx_, fy = cy_;                  // stage position of qf (streamed rows)
      const double dd = qf - xf;
      double gd[NR], rr[NR];
      rows_dir(dd, gd);
      const double ddx = cx_, ddy = cy_;                // stage direction
      double rbest = INFINITY;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const bool cand = pres[i] & !act[i] & (gd[i] > FIN_GD_MIN);
        // (reciprocal to 2e-15 instead of the IEEE division of the oracles: 26 instead of 74 cycles per row on the round's
        // serial path; the last bit of a ratio only matters on an exact tie between two blocking rows)
        rr[i] = cand ? fmax(slk[i] + gd[i], 0.0) * fast_rcp(gd[i]) : INFINITY;
        rbest = fmin(rbest, rr[i]);
      }
      // multipliers on A and slack functions outside A at qf: the VALUES first (group extrema).  Most rounds end here,
      // certified, and never look at a row index; only a group that has to exchange a row finds which one: the lowest
      // canonical index among the rows attaining the extremum (numpy's argmin order), one integer group minimum each.
      double ymin = INFINITY, ymax = 0.0, smin = INFINITY;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const bool ta = act[i], ti = pres[i] & !ta;
        ymin = fmin(ymin, ta ? y[i] : INFINITY);
        ymax = fmax(ymax, ta ? y[i] : 0.0);
        smin = fmin(smin, ti ? slk[i] : INFINITY);
      }
      if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
        for (int t = 0; t < NOBS_S; ++t) {
          if ((pbits >> t) & 1u) {
            if ((abits >> t) & 1u) { const double yt = lds_sz[grp][t][lane][1]; ymin = fmin(ymin, yt); ymax = fmax(ymax, yt); }
            else { smin = fmin(smin, s_slack(t, fx, fy)); rbest = fmin(rbest, s_ratio(t, fx, fy, ddx, ddy)); }
          }
        }
      }
      rbest = gmin<G>(rbest);
      ymin = gmin<G>(ymin);
      int bi = 0x7fffffff;
      const bool blocked = !fin_done & (rbest < 1.0);
      const bool dropping = !fin_done & !blocked & (ymin < -FIN_EPS);
      if (__any(blocked)) {                         // which row blocks: lowest canonical index among the rows at the minimum
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int ci = ci_of(i);
          bi = ((rr[i] == rbest) & (ci < bi)) ? ci : bi;          // (rr is finite on candidates only)
        }
        if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
          for (int t = 0; t < NOBS_S; ++t) {
            if (((pbits >> t) & 1u) && !((abits >> t) & 1u)) {
              const int ci = ci_s(t);
              if (s_ratio(t, fx, fy, ddx, ddy) == rbest && ci < bi) bi = ci;
            }
          }
        }
        bi = gmin_int<G>(bi);
      }
      if (__any(blocked | dropping)) {
        // register rows by selects (the groups of a wave take different arms): the blocking row joins; every row with a
        // negative multiplier leaves
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const bool hd = dropping & act[i] & (y[i] < -FIN_EPS), ha = blocked & pres[i] & !act[i] & (ci_of(i) == bi);
          y[i] = hd ? 0.0 : y[i];
          act.m = (act.m & ~((unsigned)hd << i)) | ((unsigned)ha << i);
        }
        if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
          for (int t = 0; t < NOBS_S; ++t) {
            const bool on_t = (pbits >> t) & 1u, act_t = (abits >> t) & 1u;
            if (dropping && act_t && lds_sz[grp][t][lane][1] < -FIN_EPS) { abits &= ~(1u << t); lds_sz[grp][t][lane][1] = 0.0; }
            if (blocked && on_t && !act_t && ci_s(t) == bi) abits |= 1u << t;
          }
        }
      }
      if (!fin_done) {
        if (blocked) {
          xf = fma(rbest, dd, xf);
        } else {
          xf = qf;
          // polish: an equality solve left above FIN_POLISH_TOL (two active rows a few 1e-6 from parallel: the multiplier
          // iteration contracts that direction by 2 / (2 + rho sigma^2) per correction only) gets one more round on the
          // same set at the stiffer penalty before it may certify
          const bool polish = !dropping & (eres > FIN_POLISH_TOL) & (rho == FIN_RHO) & (rnd + 1 < P.fin_rounds);
          rho = polish ? FIN_RHO_POLISH : rho;
          if (!dropping && !polish) {
            ymax = gmax<G>(ymax);
            smin = gmin<G>(smin);
            const double qabs = gmax<G>(fabs(qf));
            fin_done = true;
            certified = fok && (r_g <= FIN_EPS) && (rd_g <= FIN_EPS + FIN_DUAL_REL * ymax) && (smin >= -FIN_EPS) && (qabs < 1e300);
            diag_cert = fmin(ymin, smin);      // how decisively the certificate holds (weakly active rows -> ~0)
          }
        }
        diag_rounds = rnd + 1;
        diag_eres = eres;
      }
    }
    if (ipm_ok) {
      if (certified) {
        q = xf;
      } else {
        status = LIPMPC_STATUS_UNCERTIFIED;
#pragma unroll
        for (int i = 0; i < NR; ++i) act.set(i, pres[i] && (z[i] > FIN_IDENT * s[i]));
        abits = fbits;
      }
    }
  }

  PH(9)
  // ---- outputs -----------------------------------------------------------------------------------
  const bool have_sol = (status == LIPMPC_STATUS_SOLVED) || (status == LIPMPC_STATUS_UNCERTIFIED);
  // velocities v_{a+1} of the solution
  const double ps = prefix_excl2<G>(sgn_a * q, lane);
  const double vsol = kap * q + 2.0 * kap * sgn_a * ps - sgn_a * (v0c + kap * p0c);
  double pprev = gup<G, 2>(q, lane), vprev = gup<G, 2>(vsol, lane);
  if (a == 0) { pprev = p0c; vprev = v0c; }
  const double u = (q - P.ch * pprev - P.sh_over_beta * vprev) * P.inv_one_minus_ch;
  const double dg = var_on ? (q - gc) : 0.0;
  const double objv = gsum<G>(dg * dg) + (p0x - gx) * (p0x - gx) + (p0y - gy) * (p0y - gy);
  // The canonical active set (include/lipmpc.h: `active`): the rows of the problem that are TIGHT at the returned point,
  // slack <= LIPMPC_TIGHT_TOL -- unique because the minimiser is (the finish's working set, `working`, is one of several
  // valid certificates at a degenerate vertex) -- and the tightness margin, the distance of the nearest row from changing
  // sides.  Evaluated the same way in both oracles (oracle/lipmpc_oracle.py: tight_set).
  RowFlags tight;
  unsigned tbits = 0u;
  double tm_l = INFINITY;
  if (have_sol && X) {
    slack_values(q);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      tight.set(i, pres[i] && (slk[i] <= LIPMPC_TIGHT_TOL));
      tm_l = fmin(tm_l, pres[i] ? fabs(slk[i] - LIPMPC_TIGHT_TOL) : INFINITY);
    }
    if constexpr (STREAM) {
      const double fx = cx_, fy = cy_;
#pragma unroll STREAM_UNROLL
      for (int t = 0; t < NOBS_S; ++t) {
        if ((pbits >> t) & 1u) {
          double ex, ey, bb;
          s_obs(t, ex, ey, bb);
          const double sl = ex * fx + ey * fy - bb;
          if (sl <= LIPMPC_TIGHT_TOL) tbits |= 1u << t;
          tm_l = fmin(tm_l, fabs(sl - LIPMPC_TIGHT_TOL));
        }
      }
    }
  }
  const double tight_margin = (diag != nullptr) ? gmin<G>(tm_l) : 0.0;
  // a row set as bits of the group's mask words in LDS, then to the caller's buffer
  auto put_mask = [&](const RowFlags& rf, unsigned sb, unsigned long long* __restrict__ dst) {
    for (int wi = lane; wi < P.words; wi += G) lds_act[grp][wi] = 0ull;
    wave_sync();
    if (have_sol) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int ci = ci_of(i);
        if (rf[i]) atomicOr(&lds_act[grp][ci >> 6], 1ull << (ci & 63));
      }
      if constexpr (STREAM) {
#pragma unroll STREAM_UNROLL
        for (int t = 0; t < NOBS_S; ++t) {
          const int ci = ci_s(t);
          if ((sb >> t) & 1u) atomicOr(&lds_act[grp][ci >> 6], 1ull << (ci & 63));
        }
      }
    }
    wave_sync();
    if (valid) for (int wi = lane; wi < P.words; wi += G) dst[pb * P.words + wi] = lds_act[grp][wi];
    wave_sync();
  };
  if (X) put_mask(tight, tbits, active_out);
  if (X && working_out) put_mask(act, abits, working_out);
  if (valid && X) {
    const double nanv = NAN;
    if (var_on) {
      double* Xo = X + (pb * (N + 1) + a + 1) * 4;
      Xo[2 * c] = have_sol ? q : nanv;
      Xo[2 * c + 1] = have_sol ? vsol : nanv;
      U[(pb * N + a) * 2 + c] = have_sol ? u : nanv;
    }
    if (lane < 2) {
      double* Xo = X + pb * (N + 1) * 4;
      Xo[2 * lane] = have_sol ? p0c : nanv;
      Xo[2 * lane + 1] = have_sol ? v0c : nanv;
    }
    if (lane == 0) {
      obj_out[pb] = have_sol ? objv : nanv;
      status_out[pb] = status;
      iters_out[pb] = iters;
#ifndef LIPMPC_PHASE_TIMING
      if (diag) {
        double* dg_ = diag + pb * LIPMPC_DIAG_WORDS;
        dg_[0] = diag_rounds; dg_[1] = diag_eres; dg_[2] = margin; dg_[3] = diag_cert; dg_[4] = tight_margin;
        dg_[5] = 0.0; dg_[6] = 0.0; dg_[7] = 0.0;
      }
#endif
      if (cost_out) cost_out[pb] = iters + 2 * (int)diag_rounds;      // this problem's weight for the next launch's order (a finish round ~ 1.5-2 iterations)
    }
  }
  PH(11)
#ifdef LIPMPC_PHASE_TIMING
  {
    const int it_w = max(iters, max(__shfl_xor(iters, 16), max(__shfl_xor(iters, 32), __shfl_xor(iters, 48))));
    const int rd = (int)diag_rounds;
    const int rd_w = max(rd, max(__shfl_xor(rd, 16), max(__shfl_xor(rd, 32), __shfl_xor(rd, 48))));
    const long slot = __builtin_amdgcn_readfirstlane((int)pb);
    if (diag && threadIdx.x == 0) {
      double* o = diag + slot * PH_WORDS;
      for (int k = 0; k < 24; ++k) o[k] = 10.0 * (double)ph_acc_[k];                   // 100 MHz ticks -> ns
      o[24] = 10.0 * (double)(wall_clock64() - in.t_start_wall);
      o[25] = (double)((unsigned long long)__builtin_readcyclecounter() - in.t_start_ticks);
      o[26] = it_w; o[27] = rd_w; o[28] = 1.0;
      o[29] = 10.0 * (double)(F.t_front - in.t_start_wall);      // kernel entry -> end of the front end
      o[30] = 10.0 * (double)(F.t_geom - in.t_start_wall);       // kernel entry -> headings done (loads + atan2 + sincos)
    }
  }
#endif
  StepOut r;
  r.status = status; r.iters = iters; r.theta1 = theta1; r.omega0 = omega0; r.obj = objv;
  r.ux = gbcast<G, 0>(u); r.uy = gbcast<G, 1>(u);
  return r;
}

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
#ifdef LIPMPC_WAVES2      // dev experiment: two resident waves per SIMD (256 registers per wave)
#define LIPMPC_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define LIPMPC_OCC
#endif
template <int G, int NOBS_L, int NVAR, bool DISPATCH>
__global__ __launch_bounds__(WAVE) LIPMPC_OCC void plan_step_kernel(
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
