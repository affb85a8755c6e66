// lipmpc_comm.hpp -- exchanges inside a group of G lanes (DPP rows, v_permlane16_swap), reductions, reciprocals, the fused DPP multiply-adds, wave-local fences, the dev-only phase clock
// Part of the MI355X-native batched LIP-MPC / LDCBF step solver (csrc/lipmpc_kernel.hpp includes the parts in order).
#pragma once
#include "lipmpc_types.hpp"

namespace lipmpc_dev {

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
  asm volatile("" : "+v"(x));
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
__device__ __forceinline__ double chain_rcp(double x) { return fast_rcp(x); }

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
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
struct RowFlags {
  unsigned m = 0u;
  __device__ __forceinline__ bool operator[](int i) const { return (m >> i) & 1u; }
  __device__ __forceinline__ void set(int i, bool v) { m = v ? (m | (1u << i)) : (m & ~(1u << i)); }
};

}  // namespace lipmpc_dev
