// lipmpc_lidar.hip — unknown-environment front end (BASELINE config 5): per robot and MPC step, a 2-D LiDAR scan of the
// true map -> noisy readings -> DBSCAN clusters -> convex hull per cluster, i.e. the obstacle rings the step kernel
// consumes.  Restates
//   range_finder / compute_lidar_readings / retrieve_clusters / build_local_obstacles
//       HumanoidNavigation/RangeFinder/range_finder_wth_polygons_dbscan.py:26-63, 65-83, 100-126, 157-180
//   line_polygon_intersection (compute_intersection)      HumanoidNavigation/Utils/obstacles.py:95-139
//   the call site                                          HumanoidNavigation/MPC/HumanoidMPCVariants/HumanoidMPCUnknownEnvironment.py:30-68
// One wavefront (64 lanes) per robot; everything between the ray casting and the half-spaces stays in LDS / registers
// (10.1 KB of LDS and 128 registers per wave: 16 waves per CU, a whole batch of 4096 robots resident at once):
//   1. rays: lane l owns rays l, l+64, ...; the edges of the obstacles within range are staged once per robot in LDS
//      (edge vector, the robot's offset from the edge's first vertex and their cross product: the operands of
//      compute_intersection, no global / scalar load left in the ray loops) and every ray walks them in list order, keeping
//      the nearest hit strictly inside the range (contraction off: the hit points are bit-identical to the reference's)
//   2. DBSCAN(eps, min_samples) by its order-free characterisation (oracle/lidar_oracle.py), on the readings
//      compacted in ray order: neighbour bit rows (pruned by the bounding boxes of runs of 16 readings: far pairs of runs
//      skipped, pairs within eps corner to corner set without a test, the rest tested column by column with the verdict
//      shifted in through the carry), core flags, connected components of the core points (forest of "smallest core
//      neighbour" pointers + pointer jumping, then merging trees through ballot masks of tree membership — bit operations,
//      no sweeps over neighbours' labels), clusters numbered by their smallest core index, border points to the smallest
//      neighbouring cluster
//   3. hull per cluster: Jarvis march from the lexicographically smallest point, farthest point on collinear ties
//      (= the CCW ring of extreme points Qhull / monotone chain return, same rotation as np.unique + monotone chain);
//      four clusters march at once, one per 16-lane DPP row, every lane's candidates held in registers; a step's winner is
//      guessed by a single-precision turning key and proved with the exact predicate (exact reduction only when that fails)
//   4. constraint assembly (HumanoidMPCUnknownEnvironment.py:54-62 -> ObstaclesUtils.py:60-109): closest point c and
//      unit normal eta of every hull at the robot's CoM, one hull edge per lane, from the hull still staged in LDS --
//      the (c, eta) rows are what the step solver consumes (lipmpc_plan_step_batch_c_eta); the rings themselves go to
//      HBM only when the caller asks for them
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "lipmpc_kernel.hpp"      // DPP row exchanges (lipmpc_dev::row_xor)

namespace {

constexpr int RMAX = 384;            // rays per scan (reference: 360)
constexpr int WORDS = RMAX / 64;     // neighbour bit row
constexpr int NO_ROOT = 0x7fffffff;
constexpr int SOLO_MIN = 64;        // a cluster of at least this many points gets the whole wave in the hull stage
constexpr int NCC = 64;             // candidates whose bounding circle is kept for the per-pass sector test
constexpr int VFAST = 8;            // rings of at most this many vertices (v_env) are fetched in one round of loads
constexpr int VSTAGE = 64;          // hull vertices staged per cluster (v_max <= VSTAGE)
// order buffer (int32, scratch of ONE call): [B the order is valid for, -, order[B] (robot at launch position i), weight[B]]
constexpr int SCHED_VALID = 0, SCHED_PERIOD = 1, SCHED_ORDER = 2;
constexpr int NRUN = RMAX / 16;      // runs of 16 consecutive readings
constexpr int ECAP = 152;           // edges staged per chunk of the ray phase (5 doubles each, where the points go afterwards)
static_assert(ECAP * 5 <= 2 * RMAX && ECAP % 2 == 0, "the staged edges live in the point array");
static_assert(ECAP <= 65535 && RMAX <= 65535, "16-bit edge offsets and point indices");

struct Cand { double x, y; int idx; };
#ifdef LIPMPC_LIDAR_PHASES
#define LIDAR_PHASE_END(n) if (dbg_stop == (n)) return
#else
#define LIDAR_PHASE_END(n)
#endif

// Is candidate b a better "next hull vertex" than a when standing on p?  (b strictly to the right of p->a, or collinear and
// farther; a.idx < 0 = no candidate yet.)  Candidates are given as OFFSETS from p (a.x = x_a - p_x ...), formed once per step
// instead of once per comparison.  Straight-line on purpose (the verdict assembled from masks): the march is a chain of
// dependent steps, and the nest of divergent branches the short-circuit form compiles to cost more than the arithmetic it
// skipped.  Exactly collinear pairs are rare enough for the tie-break to sit behind a wave-uniform branch.
__device__ __forceinline__ bool better_from(const Cand& a, const Cand& b) {
#pragma clang fp contract(off)
  const double cr = a.x * b.y - a.y * b.x;
  const bool both = (a.idx >= 0) & (b.idx >= 0);
  bool tie = false;
  if (__any(both & (cr == 0.0))) {
    const double da = a.x * a.x + a.y * a.y;
    const double db = b.x * b.x + b.y * b.y;
    tie = (db > da) | ((db == da) & (b.idx < a.idx));
  }
  const bool geo = (cr < 0.0) | ((cr == 0.0) & tie);
  return (b.idx >= 0) & ((a.idx < 0) | geo);
}
// the value lane l (wave-uniform) holds
__device__ __forceinline__ double lane_value(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// the value held 16 / 32 lanes away, by v_permlane16_swap / v_permlane32_swap (VALU, no trip through the LDS crossbar)
template <class T> __device__ __forceinline__ T wave_xor16(T v) { return lipmpc_dev::rowswap(v); }
__device__ __forceinline__ unsigned wave_xor32_u(unsigned u) {
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // {lanes 0-31 everywhere, lanes 32-63 everywhere}
  const unsigned lo = r[0], hi = r[1];
  return (lipmpc_dev::fresh(threadIdx.x) & 32) ? lo : hi;
}
template <class T> __device__ __forceinline__ T wave_xor32(T v) {
  if constexpr (sizeof(T) == 8) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = wave_xor32_u((unsigned)u), hi = wave_xor32_u((unsigned)(u >> 32));
    return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
  } else {
    return __builtin_bit_cast(T, wave_xor32_u(__builtin_bit_cast(unsigned, v)));
  }
}

// closest point of one hull edge (a -> b) to p and the crossing test of the edge (prev -> a) with the +X ray from p:
// the arithmetic of lipmpc_dev::closest_point_normal (ObstaclesUtils.py:50-109), one edge per call
struct EdgeCp { double d, qx, qy; bool degen, hit; };
__device__ __forceinline__ EdgeCp edge_closest(double pvx, double pvy, double ax, double ay, double bx, double by, double px,
                                               double py) {
#pragma clang fp contract(off)
  EdgeCp r;
  const double dx = bx - ax, dy = by - ay;
  const double nrm = sqrt(dx * dx + dy * dy);
  const double den = nrm * nrm;                      // sqrt-then-square, ObstaclesUtils.py:81
  r.degen = den == 0.0;
  double t = ((px - ax) * dx + (py - ay) * dy) / den;
  t = fmax(0.0, fmin(1.0, t));
  const double qx = ax + t * dx, qy = ay + t * dy;
  const double ux = qx - px, uy = qy - py;
  r.d = r.degen ? INFINITY : sqrt(ux * ux + uy * uy);
  r.qx = r.degen ? NAN : qx; r.qy = r.degen ? NAN : qy;
  const bool f0 = pvy >= py, f1 = ay >= py;
  r.hit = (f0 != f1) && (((ay - py) * (pvx - ax) >= (ax - px) * (pvy - ay)) == f1);
  return r;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void lidar_sense_kernel(
    long B, int R, int n_env, int v_env, long env_stride, double lidar_range, double eps, int min_samples,
    int n_obs_max, int v_max, const double* __restrict__ state, const double* __restrict__ env_xy,
    const int32_t* __restrict__ env_nv, const double* __restrict__ ray_table, const double* __restrict__ noise,
    double* __restrict__ obs_xy, int32_t* __restrict__ obs_nv, double* __restrict__ c_eta, int32_t* __restrict__ n_inferred,
    int32_t* __restrict__ overflow, double* __restrict__ hits_out, int32_t* __restrict__ labels_out,
    int32_t* __restrict__ sched, int dbg_stop) {
  // LDS: 10.1 KB per wave = 16 waves per CU (the 160 KB of a CU are what caps this kernel's occupancy, not its registers:
  // the scan is latency-bound, and went from 8 to 16 resident waves per CU with this layout).  One array of points, everything
  // a lane owns of its own points (coordinates, cluster root) in registers, and the three small tables of the three phases on
  // one another.
  __shared__ __attribute__((aligned(16))) double pint_[2 * RMAX];   // ray phase: staged edges; then the readings (x, y), compacted in ray order
  double* const edge_ = pint_;                   // [ECAP][4]: g = b - a and f = robot - a of every staged edge ...
  double* const nua_ = pint_ + 4 * ECAP;         // [ECAP]:    ... and g x f, the ray-independent numerator of compute_intersection
  __shared__ __attribute__((aligned(16))) int comp_[RMAX];          // -1 = no reading; core: component root; else NO_ROOT; hull stage: cluster offsets
  __shared__ unsigned short cand_[RMAX];         // obstacles that can be hit from here, list order; then the ray of every reading; then member lists
  __shared__ unsigned short eoff_[66];           // first staged edge of the chunk's candidates
  __shared__ __attribute__((aligned(16))) double small_[NCC * 3];   // one phase's small table at a time:
  double (*const candc_)[3] = reinterpret_cast<double (*)[3]>(small_);            // rays: bounding circle (centre, radius) of the first NCC candidates
  double (*const bb16_)[4] = reinterpret_cast<double (*)[4]>(small_);             // neighbour rows: bounding box (x0, x1, y0, y1) of each run of 16 points
  int* const roots_ = reinterpret_cast<int*>(small_);                             // components on: cluster roots, ascending [64]
  unsigned short* const stagei_ = reinterpret_cast<unsigned short*>(small_ + 32); // hulls: [4][VSTAGE] vertices of the rings being marched, as point indices
  static_assert(NRUN * 4 * 8 <= NCC * 3 * 8 && 32 * 8 + 4 * VSTAGE * 2 <= NCC * 3 * 8, "the small tables share one area");

  const int lane = threadIdx.x;
  if ((long)blockIdx.x >= B) return;
  // Which robot this wave scans: the block index, or -- with an order buffer (include/lipmpc.h) -- the robot the order
  // kernel of THIS call put at this position: ranked by an estimate of its reading count (lidar_weight_kernel) and dealt out so
  // that the robots sharing a SIMD are a heavy one with light ones (lidar_order_kernel).  A scan's length varies 3x with the
  // number of readings, and with the whole batch resident the launch lasts as long as its most loaded SIMD: 155 us as the robots
  // come, 112 us ranked by the true counts, 123 us ranked by the estimate, ranking included (tools/lidar_order.py).  Any order
  // gives the same results.
  long b = blockIdx.x;
  if (sched && sched[SCHED_VALID] == (int)B) {
    const long r = sched[SCHED_ORDER + blockIdx.x];
    if (r >= 0 && r < B) b = r;
    // the robots of a SIMD come one from each round of `period` launch positions, the heaviest from the first: that one goes
    // first when the SIMD picks an instruction (the launch lasts as long as its longest scan)
    const int period = sched[SCHED_PERIOD];
    if (period > 0) {
      const long round = blockIdx.x / period;
      if (round == 0) __builtin_amdgcn_s_setprio(3);
      else if (round == 1) __builtin_amdgcn_s_setprio(1);
    }
  }
#ifdef LIPMPC_LIDAR_PHASES
  const unsigned long long t_enter = wall_clock64();
  if (dbg_stop == 8) {            // placement probe (tools/lidar_placement.py): where the dispatcher put launch position blockIdx.x
    if (lane == 0) {
      n_inferred[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
      overflow[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
    }
    for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127);                     // stay resident while the grid is placed
    return;
  }
#endif
  const double x0 = state[b * 5 + 0], y0 = state[b * 5 + 2];
  const double* exy = env_xy + b * env_stride * (long)n_env * v_env * 2;
  const int32_t* env = env_nv + b * env_stride * (long)n_env;
  int n_cand = 0;
  int in_ovf = 0;                 // inputs beyond what this kernel holds: more than RMAX obstacles in range, rings longer than v_env

  // ---- 1. ray casting (compute_lidar_readings) ---------------------------------------------------
  // candidate obstacles: those whose bounding circle comes within the range (conservative: an obstacle that is
  // skipped cannot hold a point closer than lidar_range), compacted once per robot in list order
  for (int j0 = 0; j0 < n_env; j0 += 64) {
    const int j = j0 + lane;
    bool keep = false;
    double mx = 0.0, my = 0.0, rad = 0.0;
    if (j < n_env) {
      int nv = env[j];
      if (nv > v_env) { nv = v_env; in_ovf = 1; }
      const double* ring = exy + (long)j * v_env * 2;
      double rad2 = 0.0;
      if (v_env <= VFAST) {          // (wave-uniform) the whole ring in one round of loads: a loop of dependent loads is a loop of misses
        double vx[VFAST], vy[VFAST];
#pragma unroll
        for (int e = 0; e < VFAST; ++e) { const int ee = e < nv ? e : 0; vx[e] = nv > 0 ? ring[2 * ee] : 0.0; vy[e] = nv > 0 ? ring[2 * ee + 1] : 0.0; }
#pragma unroll
        for (int e = 0; e < VFAST; ++e) if (e < nv) { mx += vx[e]; my += vy[e]; }
        if (nv > 0) { mx /= nv; my /= nv; }
#pragma unroll
        for (int e = 0; e < VFAST; ++e) if (e < nv) rad2 = fmax(rad2, (vx[e] - mx) * (vx[e] - mx) + (vy[e] - my) * (vy[e] - my));
      } else {
        for (int e = 0; e < nv; ++e) { mx += ring[2 * e]; my += ring[2 * e + 1]; }
        if (nv > 0) { mx /= nv; my /= nv; }
        for (int e = 0; e < nv; ++e) rad2 = fmax(rad2, (ring[2 * e] - mx) * (ring[2 * e] - mx) + (ring[2 * e + 1] - my) * (ring[2 * e + 1] - my));
      }
      if (nv > 0) {
        rad = sqrt(rad2) * (1.0 + 1e-12);                  // (any circle around the ring does: the margins below cover the rounding)
        const double reach = (lidar_range + rad) * (1.0 + 1e-9) + 1e-9;
        keep = (mx - x0) * (mx - x0) + (my - y0) * (my - y0) <= reach * reach;
      }
    }
    const unsigned long long ball = __ballot(keep);
    if (keep) {
      const int k = n_cand + __popcll(ball & ((1ull << lane) - 1ull));
      if (k < RMAX) cand_[k] = (unsigned short)j;
      if (k < NCC) { candc_[k][0] = mx; candc_[k][1] = my; candc_[k][2] = rad; }
    }
    n_cand += __popcll(ball);
  }
  // cand_ holds RMAX obstacles: the (RMAX+1)-th obstacle in range is dropped and the scan flagged (overflow), never
  // read past the list
  if (n_cand > RMAX) { n_cand = RMAX; in_ovf = 1; }
  in_ovf = __any(in_ovf) ? 1 : 0;
  __syncthreads();
  LIDAR_PHASE_END(6);
  // this lane's rays (i = lane + 64 p): direction b1 - a1, nearest hit so far.  A ray beyond the resolution has a zero
  // direction: every denominator is 0, it never hits.
  double rdx[WORDS], rdy[WORDS], bd[WORDS], hx[WORDS], hy[WORDS];
  const double inv_len2 = 1.0 / (lidar_range * lidar_range);     // 1 / |ray|^2 for the sector test (an estimate with a margin)
#pragma unroll
  for (int p = 0; p < WORDS; ++p) {
#pragma clang fp contract(off)
    const int i = p * 64 + lane;
    const bool on = i < R;
    const double ex = x0 + lidar_range * (on ? ray_table[2 * i] : 0.0), ey = y0 + lidar_range * (on ? ray_table[2 * i + 1] : 0.0);
    rdx[p] = on ? ex - x0 : 0.0; rdy[p] = on ? ey - y0 : 0.0;
    bd[p] = lidar_range; hx[p] = 0.0; hy[p] = 0.0;
  }
  // The candidates' edges go through LDS in chunks of at most 64 obstacles / ECAP edges (one chunk on ordinary maps):
  // lane c stages candidate jc0 + c -- edge vector g = b2 - a2 and offset f = a1 - a2 of the ray origin, the two operands
  // compute_intersection (Utils/obstacles.py:107-123) forms from the edge -- then every pass of 64 rays walks the staged
  // edges in list order.  Hits are kept across chunks in registers.
  for (int jc0 = 0; jc0 < n_cand;) {
    const bool mine = jc0 + lane < n_cand;
    const int j = mine ? cand_[jc0 + lane] : 0;
    const int nv = mine ? min(env[j], v_env) : 0;
    int incl = nv;                                   // inclusive prefix sum of the edge counts over the lanes
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, 64); if (lane >= m) incl += o; }
    const bool fits = mine && incl <= ECAP;          // a prefix of the lanes (incl is monotone)
    int nfit = __popcll(__ballot(fits));
    if (nfit == 0) { in_ovf = 1; nfit = 1; }         // one ring longer than ECAP edges: skipped and flagged
    if (fits) {
#pragma clang fp contract(off)
      const double* ring = exy + (long)j * v_env * 2;
      const int off = incl - nv;
      eoff_[lane] = off;
      auto put_edge = [&](int e, double ax, double ay, double bx, double by) {
#pragma clang fp contract(off)
        double* o = edge_ + 4 * (off + e);
        const double gx = bx - ax, gy = by - ay, fx = x0 - ax, fy = y0 - ay;
        o[0] = gx; o[1] = gy; o[2] = fx; o[3] = fy;
        nua_[off + e] = gx * fy - gy * fx;
      };
      if (v_env <= VFAST) {
        double vx[VFAST], vy[VFAST];
#pragma unroll
        for (int e = 0; e < VFAST; ++e) { const int ee = e < nv ? e : 0; vx[e] = nv > 0 ? ring[2 * ee] : 0.0; vy[e] = nv > 0 ? ring[2 * ee + 1] : 0.0; }
#pragma unroll
        for (int e = 0; e < VFAST; ++e)
          if (e < nv) { const bool last = e + 1 == nv; put_edge(e, vx[e], vy[e], last ? vx[0] : vx[(e + 1) % VFAST], last ? vy[0] : vy[(e + 1) % VFAST]); }
      } else {
        double ax = nv > 0 ? ring[0] : 0.0, ay = nv > 0 ? ring[1] : 0.0;
        const double fx0 = ax, fy0 = ay;
        for (int e = 0; e < nv; ++e) {
          const bool last = e + 1 == nv;
          const double bx = last ? fx0 : ring[2 * (e + 1)], by = last ? fy0 : ring[2 * (e + 1) + 1];
          put_edge(e, ax, ay, bx, by);
          ax = bx; ay = by;
        }
      }
      if (lane == nfit - 1) eoff_[nfit] = incl;
    } else if (lane == 0) { eoff_[0] = 0; eoff_[1] = 0; }      // (only when nothing fitted)
    __syncthreads();
    LIDAR_PHASE_END(7);
    for (int c = 0; c < nfit; ++c) {
      const int e0 = eoff_[c], e1 = eoff_[c + 1];
      const int jc = jc0 + c;
      double wx = 0.0, wy = 0.0, cr2 = INFINITY;
      if (jc < NCC) { wx = candc_[jc][0] - x0; wy = candc_[jc][1] - y0; const double cr = candc_[jc][2] + 1e-6; cr2 = cr * cr; }
#pragma unroll
      for (int p = 0; p < WORDS; ++p) {
        if (p * 64 >= R) continue;
        // sector test: the 64 rays of this pass span 64 degrees; an obstacle none of them comes near (distance from
        // its bounding circle's centre to the ray segment > radius, with a margin far above the rounding of this
        // estimate) is skipped by the whole wave — its edges could not have produced a hit for any of these rays
        {
          const double tt = fmin(1.0, fmax(0.0, (wx * rdx[p] + wy * rdy[p]) * inv_len2));
          const double ddx = wx - tt * rdx[p], ddy = wy - tt * rdy[p];
          if (!__any(ddx * ddx + ddy * ddy <= cr2)) continue;
        }
        for (int e = e0; e < e1; ++e) {
#pragma clang fp contract(off)
          const double gx = edge_[4 * e], gy = edge_[4 * e + 1], fx = edge_[4 * e + 2], fy = edge_[4 * e + 3];
          const double denom = gy * rdx[p] - gx * rdy[p];
          const double nua = nua_[e];
          const double nub = rdx[p] * fy - rdy[p] * fx;
          // 0 <= ua <= 1 and 0 <= ub <= 1 for ua = nua / denom, ub = nub / denom decided WITHOUT dividing: a correctly
          // rounded quotient is >= 0 exactly when the signs agree (or the numerator is +-0) and <= 1 exactly when
          // |numerator| <= |denominator| (a quotient above 1 is at least 1 + 2^-53 (1 + tiny) and rounds to 1 + 2^-52;
          // the one unreachable exception: a negative quotient below 5e-324 in magnitude, which rounds to -0.0 >= 0).
          // Only a ray that really hits the edge pays for the division that places the hit.
          const double ad = fabs(denom), sa = (denom > 0.0) ? nua : -nua, sb = (denom > 0.0) ? nub : -nub;
          if ((denom != 0.0) & (sa >= 0.0) & (sa <= ad) & (sb >= 0.0) & (sb <= ad)) {
            const double ua = nua / denom;
            const double qx = x0 + ua * rdx[p], qy = y0 + ua * rdy[p];
            const double dd = sqrt((qx - x0) * (qx - x0) + (qy - y0) * (qy - y0));
            if (dd < bd[p]) { bd[p] = dd; hx[p] = qx; hy[p] = qy; }      // strictly nearer: ties keep the earlier edge
          }
        }
      }
    }
    __syncthreads();
    jc0 += nfit;
  }
  in_ovf = __any(in_ovf) ? 1 : 0;
  // Readings (hit + noise) go from the registers straight into the list compacted in ray order (typically 110-200 of 360 rays
  // return one): clustering and hulls then sweep n points instead of RMAX slots.  Order is preserved, so "smallest core index"
  // numbering, border-point assignment and every index tie-break are those of the uncompacted scan.  (The staged edges are
  // dead: the ray loop ends on a barrier.)
  int n_pts = 0;
#pragma unroll
  for (int p = 0; p < WORDS; ++p) {
#pragma clang fp contract(off)
    const int i = p * 64 + lane;
    const bool have = bd[p] < lidar_range;           // bd starts at the range and only ever gets strictly smaller
    double qx = have ? hx[p] : 0.0, qy = have ? hy[p] : 0.0;
    if (have && noise) { qx = qx + noise[(b * R + i) * 2]; qy = qy + noise[(b * R + i) * 2 + 1]; }
    if (hits_out && i < R) { hits_out[(b * R + i) * 2] = have ? qx : NAN; hits_out[(b * R + i) * 2 + 1] = have ? qy : NAN; }
    const unsigned long long ball = __ballot(have);
    if (have) {
      const int k = n_pts + __popcll(ball & ((1ull << lane) - 1ull));
      pint_[2 * k] = qx; pint_[2 * k + 1] = qy; cand_[k] = (unsigned short)i;      // cand_ is free after the ray casting: ray of point k
    }
    n_pts += __popcll(ball);
  }
  __syncthreads();

  LIDAR_PHASE_END(1);
  // ---- 2. DBSCAN ------------------------------------------------------------------------------------
  // The launch lasts as long as its longest scan, and a scan with many readings (quadratically more pair tests, the longest hull)
  // shares its SIMD with three others: from here on it goes first when the SIMD picks an instruction.
  if (n_pts > 256) __builtin_amdgcn_s_setprio(3);
  else if (n_pts > 160) __builtin_amdgcn_s_setprio(2);
  else if (n_pts > 112) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
  const int NW = (n_pts + 63) >> 6;                      // words / passes actually in use (wave-uniform)
  const int npad = NW << 6;
  if (labels_out) for (int i = lane; i < R; i += 64) labels_out[b * R + i] = -2;      // -2 = no reading
  const double eps2 = eps * eps;
  unsigned long long vmask[WORDS];                  // which points exist
#pragma unroll
  for (int w = 0; w < WORDS; ++w) {
    const int left = n_pts - w * 64;
    vmask[w] = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
  }
  int touch[WORDS];                                  // smallest tree root point lane + 64 k touches (NO_ROOT: none)
#pragma unroll
  for (int w = 0; w < WORDS; ++w) touch[w] = NO_ROOT;
  // ---- 2a. the readings as chains ---------------------------------------------------------------------
  // Readings come in ray order: consecutive ones on one obstacle's outline lie centimetres apart, eps is 0.3 m.  Cut the list
  // where two consecutive readings are farther apart than eps.  A reading of a piece of at least min_samples readings is a core
  // point if it has min_samples - 1 neighbours among the readings one and two places from it; where that cheap count falls short
  // the neighbours are counted exactly (only a piece's free end may turn out not to be a core point: it is then a border point).
  // Pieces whose bounding boxes are farther apart than eps hold no neighbours of each other; pieces the boxes cannot separate are
  // decided by the distance test between their core points (one pair within eps: one cluster); the last piece and the first may
  // be joined across ray 0 by the pair (last, first).  A chained piece of core points is ONE cluster, numbered by its first core
  // point; a piece shorter than min_samples holds no core point and is noise; a border point goes to the cluster with the smallest
  // root among its core neighbours.  That is DBSCAN's answer without the neighbour rows; whenever any part of the proof fails
  // (more than 8 pieces, a non-core reading inside a piece, a short piece within eps of another, ...) the scan takes the general
  // route below.
  bool chains = false;
  if (n_pts >= 1) {
#pragma clang fp contract(off)
    constexpr int SEGMAX = 8;
    unsigned long long brk[WORDS];                   // bit l of brk[w]: reading w * 64 + l is the last of its piece
    bool far2[WORDS], back2[WORDS];                  // this lane's reading of word w has its second neighbour after / before it within eps
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      brk[w] = 0ull; far2[w] = false; back2[w] = false;
      if (w >= NW) continue;
      const int i = w * 64 + lane;
      const bool valid = i < n_pts;
      const int i1 = min(i + 1, n_pts - 1), i2 = min(i + 2, n_pts - 1), ib = max(min(i, n_pts - 1) - 2, 0);
      const double mx = pint_[2 * (valid ? i : 0)], my = pint_[2 * (valid ? i : 0) + 1];
      const double d1x = mx - pint_[2 * i1], d1y = my - pint_[2 * i1 + 1];
      const double d2x = mx - pint_[2 * i2], d2y = my - pint_[2 * i2 + 1];
      const double dbx = mx - pint_[2 * ib], dby = my - pint_[2 * ib + 1];
      const bool link1 = (i + 1 < n_pts) & (d1x * d1x + d1y * d1y <= eps2);
      brk[w] = __ballot(valid & !link1);
      far2[w] = valid & (i + 2 < n_pts) & (d2x * d2x + d2y * d2y <= eps2);
      back2[w] = valid & (i >= 2) & (dbx * dbx + dby * dby <= eps2);
    }
    int nseg = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) nseg += __popcll(brk[w]);
    // (last, first): the pair that may join the last piece to the first across ray 0
    const double wdx = pint_[2 * (n_pts - 1)] - pint_[0], wdy = pint_[2 * (n_pts - 1) + 1] - pint_[1];
    const bool wrap_close = n_pts >= 4 && wdx * wdx + wdy * wdy <= eps2;
    bool ok = nseg <= SEGMAX;
    if (ok) {
      // bounding box, first reading and length of every piece, piece s in lane s
      double bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
      int pstart = 0, plen = 0;
      int a = 0, s = 0;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        unsigned long long m = brk[w];               // wave-uniform
        while (m) {
          const int last = w * 64 + __ffsll((long long)m) - 1;
          m &= m - 1;
          double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
#pragma unroll
          for (int w2 = 0; w2 < WORDS; ++w2) {
            if (w2 >= NW || w2 * 64 > last || w2 * 64 + 63 < a) continue;
            const int i2 = w2 * 64 + lane;
            const bool in = (i2 >= a) & (i2 <= last);
            const double x = pint_[2 * (in ? i2 : a)], y = pint_[2 * (in ? i2 : a) + 1];
            x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
          }
          x0 = fmin(x0, lipmpc_dev::row_xor<1>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<1>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<1>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<1>(y1));
          x0 = fmin(x0, lipmpc_dev::row_xor<2>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<2>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<2>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<2>(y1));
          x0 = fmin(x0, lipmpc_dev::row_xor<4>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<4>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<4>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<4>(y1));
          x0 = fmin(x0, lipmpc_dev::row_xor<8>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<8>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<8>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<8>(y1));
          x0 = fmin(x0, wave_xor16(x0)); x1 = fmax(x1, wave_xor16(x1)); y0 = fmin(y0, wave_xor16(y0)); y1 = fmax(y1, wave_xor16(y1));
          x0 = fmin(x0, wave_xor32(x0)); x1 = fmax(x1, wave_xor32(x1)); y0 = fmin(y0, wave_xor32(y0)); y1 = fmax(y1, wave_xor32(y1));
          if (lane == s) { bx0 = x0; bx1 = x1; by0 = y0; by1 = y1; pstart = a; plen = last - a + 1; }
          a = last + 1; ++s;
        }
      }
      // Two pieces whose boxes lie farther apart than eps (eps with the margin of the run boxes) have no pair of neighbours.
      // A pair of pieces the boxes cannot separate (two sides of one obstacle make an L, and an L's box is large) is decided
      // by the distance test itself, the longer piece's readings in the lanes, the shorter one's coming one by one: no pair
      // within eps -- separate after all; some pair within eps -- the two pieces are one cluster, PROVIDED each is long enough
      // to be all core points on its own (a short piece next to a cluster would be border points: the general route's job).
      // Pieces that are one cluster share a label, the smallest piece number among them (lane s: label of piece s).
      const double epsx = eps * (1.0 + 1e-6) + 1e-9;
      const double ax0 = bx0 - epsx, ax1 = bx1 + epsx, ay0 = by0 - epsx, ay1 = by1 + epsx;
      const bool joined = wrap_close && nseg >= 2;     // the last piece and the first: joined across ray 0 by (last, first)
      auto wave_min = [&](int v) {
        v = min(v, lipmpc_dev::row_xor<1>(v)); v = min(v, lipmpc_dev::row_xor<2>(v)); v = min(v, lipmpc_dev::row_xor<4>(v));
        v = min(v, lipmpc_dev::row_xor<8>(v)); v = min(v, wave_xor16(v)); v = min(v, wave_xor32(v));
        return __builtin_amdgcn_readfirstlane(v);
      };
      // every reading's piece; the readings of long-enough pieces that the cheap count does NOT prove to be core points (the
      // ends of a piece whose second neighbour is farther than eps, mostly) get their neighbours counted exactly, one wave-wide
      // sweep each: core after all, or not -- then the reading is a border point of whatever cluster its core neighbours
      // belong to (assigned further down), it links no pieces, and it cannot be a cluster's first core point.
      int pcv[WORDS], stv[WORDS], env[WORDS];
      bool noncore[WORDS];
      unsigned long long susm[WORDS], ncm[WORDS];
      int before = 0, nsus = 0;
      const int len_first = __builtin_amdgcn_readlane(plen, 0), len_last = __builtin_amdgcn_readlane(plen, max(nseg - 1, 0));
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        pcv[w] = 0; stv[w] = 0; env[w] = 0; noncore[w] = false; susm[w] = 0ull; ncm[w] = 0ull;
        if (w >= NW) continue;
        const int i = w * 64 + lane;
        const bool valid = i < n_pts;
        pcv[w] = min(before + __popcll(brk[w] & ((1ull << lane) - 1ull)), nseg - 1);
        before += __popcll(brk[w]);
        stv[w] = __shfl(pstart, pcv[w], 64);
        const int pl = __shfl(plen, pcv[w], 64);
        env[w] = stv[w] + pl - 1;
        const int lenj = pl + ((joined && pcv[w] == 0) ? len_last : 0) + ((joined && pcv[w] == nseg - 1) ? len_first : 0);
        int cnt = (i > stv[w]) + (i < env[w]) + far2[w] + back2[w];
        if (wrap_close && (i == 0 || i == n_pts - 1)) ++cnt;
        susm[w] = __ballot(valid & (lenj >= min_samples) & (cnt < min_samples - 1));
        nsus += __popcll(susm[w]);
      }
      if (nsus > 8) ok = false;
      if (ok && nsus > 0) {
#pragma unroll
        for (int we = 0; we < WORDS; ++we) {
          unsigned long long m = susm[we];
          while (m) {
            const int le = __ffsll((long long)m) - 1, e = we * 64 + le;
            m &= m - 1;
            const double ex = pint_[2 * e], ey = pint_[2 * e + 1];
            int c = 0;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
              if (w >= NW) continue;
              const int i = w * 64 + lane;
              const double dx = pint_[2 * min(i, n_pts - 1)] - ex, dy = pint_[2 * min(i, n_pts - 1) + 1] - ey;
              c += __popcll(__ballot((i < n_pts) & (dx * dx + dy * dy <= eps2)));
            }
            if (c < min_samples) {
              // only the free END of a piece may fail to be a core point: anywhere else the chain of core points would be cut
              const int se = __builtin_amdgcn_readlane(stv[we], le), ee = __builtin_amdgcn_readlane(env[we], le);
              if ((e != se && e != ee) || (joined && (e == 0 || e == n_pts - 1))) ok = false;
              if (lane == le) noncore[we] = true;
            }
          }
        }
#pragma unroll
        for (int w = 0; w < WORDS; ++w) ncm[w] = __ballot(noncore[w]);
      }
      int lab = lane;
      auto relabel = [&](int p, int q2) {
        const int lp = __builtin_amdgcn_readlane(lab, p), lq = __builtin_amdgcn_readlane(lab, q2);
        const int lo = min(lp, lq), hi = max(lp, lq);
        lab = (lab == hi) ? lo : lab;
      };
      for (int t = 1; t < nseg && ok; ++t) {
        const double ox0 = lane_value(bx0, t), ox1 = lane_value(bx1, t), oy0 = lane_value(by0, t), oy1 = lane_value(by1, t);
        unsigned long long nmk = __ballot((lane < t) & (ox0 <= ax1) & (ox1 >= ax0) & (oy0 <= ay1) & (oy1 >= ay0));
        if (joined && t == nseg - 1) nmk &= ~1ull;                                  // (joined below)
        while (nmk && ok) {
          const int sp = __ffsll((long long)nmk) - 1;
          nmk &= nmk - 1;
          const int as = __builtin_amdgcn_readlane(pstart, sp), al = __builtin_amdgcn_readlane(plen, sp);
          const int bs = __builtin_amdgcn_readlane(pstart, t), bl = __builtin_amdgcn_readlane(plen, t);
          const bool a_longer = al > bl;
          const int vs = a_longer ? as : bs, vl = a_longer ? al : bl;               // in the lanes
          const int ls = a_longer ? bs : as, ll = a_longer ? bl : al;               // one by one
          // (a pair within eps has each reading inside the other piece's box grown by eps: of two walls meeting in a corner
          // only the readings near the corner take part)
          const int vi = a_longer ? sp : t, li = a_longer ? t : sp;
          const double vx0 = lane_value(bx0, vi) - epsx, vx1 = lane_value(bx1, vi) + epsx, vy0 = lane_value(by0, vi) - epsx, vy1 = lane_value(by1, vi) + epsx;
          const double lx0 = lane_value(bx0, li) - epsx, lx1 = lane_value(bx1, li) + epsx, ly0 = lane_value(by0, li) - epsx, ly1 = lane_value(by1, li) + epsx;
          // One pair within eps settles it, and where there is one it sits near the cut between the two pieces more often
          // than not: the one-by-one side is walked from its end nearer the other piece, and the walk stops at the first pair.
          const bool upwards = li > vi;
          bool hit = false, found = false;
#pragma unroll
          for (int wq = 0; wq < WORDS; ++wq) {
            const int wl = upwards ? wq : WORDS - 1 - wq;
            if (found || wl >= NW || wl * 64 > ls + ll - 1 || wl * 64 + 63 < ls) continue;
            const int il = wl * 64 + lane;
            const bool inl = (il >= ls) & (il < ls + ll) & !noncore[wl];
            const double qx = pint_[2 * (inl ? il : ls)], qy = pint_[2 * (inl ? il : ls) + 1];
            const unsigned long long cm = __ballot(inl & (qx >= vx0) & (qx <= vx1) & (qy >= vy0) & (qy <= vy1));
            if (!cm) continue;
#pragma unroll
            for (int w2 = 0; w2 < WORDS; ++w2) {
              if (w2 >= NW || w2 * 64 > vs + vl - 1 || w2 * 64 + 63 < vs) continue;
              const int i2 = w2 * 64 + lane;
              const bool inv = (i2 >= vs) & (i2 < vs + vl) & !noncore[w2];
              const double mx = pint_[2 * (inv ? i2 : vs)], my = pint_[2 * (inv ? i2 : vs) + 1];
              const bool in = inv & (mx >= lx0) & (mx <= lx1) & (my >= ly0) & (my <= ly1);
              if (found || !__any(in)) continue;
              unsigned long long c2 = cm;
              while (c2 && !found) {
#pragma unroll
                for (int rep4 = 0; rep4 < 4; ++rep4) {
                  if (!c2) break;
                  const int bit = upwards ? __ffsll((long long)c2) - 1 : 63 - __clzll((long long)c2);
                  c2 &= ~(1ull << bit);
                  const int j = wl * 64 + bit;
                  const double dx = mx - pint_[2 * j], dy = my - pint_[2 * j + 1];
                  hit |= in & (dx * dx + dy * dy <= eps2);
                }
                found = __any(hit);
              }
            }
          }
          if (found) {
            if (al < min_samples || bl < min_samples) ok = false;
            else relabel(sp, t);
          }
        }
      }
      if (joined) relabel(0, nseg - 1);
      int mlen = 0;                                    // lane L: readings of the cluster labelled L
      for (int t = 0; t < nseg; ++t) {
        const int lt = __builtin_amdgcn_readlane(lab, t), nt = __builtin_amdgcn_readlane(plen, t);
        if (lane == lt) mlen += nt;
      }
      // every reading: its cluster's label, whether it is a core point; a cluster's root = its first core point
      int lbv[WORDS];
      bool corev[WORDS];
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        lbv[w] = 0; corev[w] = false;
        if (w >= NW) continue;
        lbv[w] = __shfl(lab, pcv[w], 64);
        corev[w] = (w * 64 + lane < n_pts) & (__shfl(mlen, lbv[w], 64) >= min_samples) & !noncore[w];
      }
      int croot = NO_ROOT;                             // lane L: first core point of the cluster labelled L
      for (int t = 0; t < nseg && ok; ++t) {
        if (__builtin_amdgcn_readlane(lab, t) != t) continue;                       // (t is its cluster's label)
        if (__builtin_amdgcn_readlane(mlen, t) < min_samples) continue;             // (noise)
        int first = NO_ROOT;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) if (w < NW && corev[w] && lbv[w] == t) first = min(first, w * 64 + lane);
        first = wave_min(first);
        if (first == NO_ROOT) ok = false;                                           // (a long piece without a core point)
        if (lane == t) croot = first;
      }
      int rootv[WORDS];
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        rootv[w] = -1;
        if (w >= NW) continue;
        const int root = __shfl(croot, lbv[w], 64);
        rootv[w] = (w * 64 + lane < n_pts) ? (corev[w] ? root : NO_ROOT) : -1;
      }
      // the readings found not to be core points: border points of the cluster with the smallest root among their core
      // neighbours (noise if they have none)
      if (ok && nsus > 0) {
#pragma unroll
        for (int we = 0; we < WORDS; ++we) {
          unsigned long long m = ncm[we];
          while (m) {
            const int le = __ffsll((long long)m) - 1, e = we * 64 + le;
            m &= m - 1;
            const double ex = pint_[2 * e], ey = pint_[2 * e + 1];
            int best = NO_ROOT;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
              if (w >= NW) continue;
              const int i = w * 64 + lane;
              const double dx = pint_[2 * min(i, n_pts - 1)] - ex, dy = pint_[2 * min(i, n_pts - 1) + 1] - ey;
              if (corev[w] && dx * dx + dy * dy <= eps2) best = min(best, rootv[w]);
            }
            best = wave_min(best);
            if (lane == le) touch[we] = best;
          }
        }
      }
      if (ok) {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) comp_[w * 64 + lane] = rootv[w];
        chains = true;
      }
    }
  }
  if (chains) __syncthreads();
  if (!chains) {
  // row[k][w]: neighbour bits of point lane + 64 k against the 64 points of word w — kept in registers (the lane
  // that owns a point is the only one that reads its row).
  // All-pairs is 147 k distance tests for 384 readings (it was the longest phase of the scan), so the sweep is pruned
  // and each test made cheap:
  //  * readings come in ray order, so a RUN of 16 consecutive points is a short piece of one obstacle's outline with a
  //    small bounding box.  Word k is tested against run a of word w only if the run's box comes within eps of the box of one
  //    of k's four runs (a 24 x 24 bit matrix of run pairs, one lane per run, computed once); both box tests are conservative
  //    (eps with a margin far above any rounding), so no neighbour pair is ever dropped; and a pair of runs whose boxes lie
  //    within eps of each other corner to corner (a dense stretch of wall; a robot hemmed in) is all ones without a test;
  //  * a visited (word, run) tile is computed column by column: every lane holds its point of word k in registers, the
  //    run's point comes as one LDS broadcast read, and the compare's result goes into the lane's row word through the
  //    carry of an add (7 VALU instructions per 64 pair tests, nothing scalar in the chain).
  unsigned long long row[WORDS][WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
#pragma unroll
    for (int w = 0; w < WORDS; ++w) row[k][w] = 0ull;
  }
#pragma unroll
  for (int w = 0; w < WORDS; ++w) {                       // boxes of the runs (empty run: an empty box)
    if (w >= NW) continue;
    const bool vi = (vmask[w] >> lane) & 1ull;
    const double wx = pint_[2 * (w * 64 + lane)], wy = pint_[2 * (w * 64 + lane) + 1];     // (slots past n_pts: never used)
    double x0 = vi ? wx : INFINITY, x1 = vi ? wx : -INFINITY, y0 = vi ? wy : INFINITY, y1 = vi ? wy : -INFINITY;
    x0 = fmin(x0, lipmpc_dev::row_xor<1>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<1>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<1>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<1>(y1));
    x0 = fmin(x0, lipmpc_dev::row_xor<2>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<2>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<2>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<2>(y1));
    x0 = fmin(x0, lipmpc_dev::row_xor<4>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<4>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<4>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<4>(y1));
    x0 = fmin(x0, lipmpc_dev::row_xor<8>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<8>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<8>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<8>(y1));
    if ((lane & 15) == 0) { double* o = bb16_[w * 4 + (lane >> 4)]; o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1; }
  }
  __syncthreads();
  const double epsx = eps * (1.0 + 1e-6) + 1e-9;         // eps with a margin for the box tests
  // run a = lane: which runs b come within eps of it (bit b of nm: some pair may be neighbours), and which lie within eps of it
  // as a whole (bit b of fm: EVERY pair is -- the farthest corners of the two boxes pass the distance test itself, same
  // operations in the same order, and rounding is monotone, so every pair of points passes it too: no margin needed)
  unsigned nm = 0u, fm = 0u;
  {
#pragma clang fp contract(off)
    const double* me = bb16_[lane < NRUN ? lane : 0];
    const double mx0 = me[0], mx1 = me[1], my0 = me[2], my1 = me[3];
    const double ax0 = mx0 - epsx, ax1 = mx1 + epsx, ay0 = my0 - epsx, ay1 = my1 + epsx;
#pragma unroll 4
    for (int rb = 0; rb < 4 * NW; ++rb) {                 // (the runs of the words in use)
      const double* o = bb16_[rb];
      const double ox0 = o[0], ox1 = o[1], oy0 = o[2], oy1 = o[3];
      if ((ox0 <= ax1) & (ox1 >= ax0) & (oy0 <= ay1) & (oy1 >= ay0)) nm |= 1u << rb;
      const double dxm = fmax(fabs(mx1 - ox0), fabs(ox1 - mx0)), dym = fmax(fabs(my1 - oy0), fabs(oy1 - my0));
      if (dxm * dxm + dym * dym <= eps2) fm |= 1u << rb;
    }
  }
  // One block of the matrix = the lane's point of word k against the 64 points of word w, 16 columns (one run) at a time,
  // highest column first: the bit goes in through the carry, bits = 2 bits + (d2 <= eps2).
  auto shift_in = [&](unsigned& bits, double d2) {
    asm("v_cmp_ge_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "s"(eps2), "v"(d2) : "vcc");
  };
  for (int k = 0; k < NW; ++k) {                          // wave-uniform
    const double mx = pint_[2 * (k * 64 + lane)], my = pint_[2 * (k * 64 + lane) + 1];      // this lane's point of word k
    const int leftk = n_pts - k * 64;
    const bool own_k = lane < leftk;                      // ... exists
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      if (w >= NW) continue;
      unsigned piece[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        piece[a] = 0u;
        const unsigned nk = (__builtin_amdgcn_readlane(nm, w * 4 + a) >> (4 * k)) & 0xFu;
        const unsigned fk = (__builtin_amdgcn_readlane(fm, w * 4 + a) >> (4 * k)) & 0xFu;
        if (nk == 0u) continue;                           // run a of word w has no point near word k
        if (nk == fk) {                                   // ... or each run of word k has all of it or none of it within eps
          piece[a] = ((fk >> (lane >> 4)) & 1u) ? 0xFFFFu : 0u;
          continue;
        }
        const double* col = pint_ + 2 * (w * 64 + a * 16);
#pragma unroll
        for (int u4 = 3; u4 >= 0; --u4) {                 // four columns' distances, then their four bits
#pragma clang fp contract(off)
          double d2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double dx = mx - col[2 * (4 * u4 + j)], dy = my - col[2 * (4 * u4 + j) + 1];
            d2[j] = dx * dx + dy * dy;
          }
#pragma unroll
          for (int j = 3; j >= 0; --j) shift_in(piece[a], d2[j]);
        }
        __builtin_amdgcn_sched_barrier(0);                // (one run's loads in flight at a time: the scheduler would hoist them all)
      }
      const unsigned lo = piece[0] | (piece[1] << 16), hi = piece[2] | (piece[3] << 16);
      const unsigned long long bits = own_k ? (((((unsigned long long)hi) << 32) | lo) & vmask[w]) : 0ull;
#pragma unroll
      for (int k2 = 0; k2 < WORDS; ++k2) if (k2 == k) row[k2][w] = bits;
    }
  }
  // core points (>= min_samples neighbours, the point itself included) start as their own root
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
    const int i = k * 64 + lane;
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) cnt += __popcll(row[k][w]);
    comp_[i] = (i < n_pts) ? ((cnt >= min_samples) ? i : NO_ROOT) : -1;
  }
  __syncthreads();
  LIDAR_PHASE_END(2);
  // Connected components of the core points, label = smallest core index of the component.
  //  (1) forest: every core point points at its smallest core neighbour (lowest set bit of its neighbour row — no
  //      label reads), pointer jumping flattens the trees;
  //  (2) merge: per tree root r (ascending, a scalar loop) the membership mask M_r of its tree is one ballot per
  //      word, and "point i touches tree r" is (row_i & M_r) != 0 — register bit operations against wave-uniform
  //      masks instead of a sweep over the neighbours' labels.  The first tree a point touches is the smallest;
  //      a point whose tree root is larger hooks its root under it (atomicMin), pointer jumping, repeat until no
  //      hook happens (typically one or two rounds).
  // The last round also yields, for the non-core points, the smallest neighbouring cluster (border points).
  unsigned long long cmask[WORDS];
#pragma unroll
  for (int w = 0; w < WORDS; ++w) { const int cj = comp_[w * 64 + lane]; cmask[w] = __ballot(cj >= 0 && cj != NO_ROOT); }
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
    if (k >= NW) continue;
    const int i = k * 64 + lane;
    const int ci = comp_[i];
    if (ci >= 0 && ci != NO_ROOT) {
      int p = i;
      bool found = false;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        const unsigned long long bits = row[k][w] & cmask[w];
        if (!found && bits != 0ull) { p = w * 64 + __ffsll((long long)bits) - 1; found = true; }
      }
      comp_[i] = p;                                   // p <= i: a core point is its own neighbour
    }
  }
  __syncthreads();
  auto flatten = [&]() {
    for (int jump = 0; jump < 16; ++jump) {
      bool moved = false;
      for (int i = lane; i < npad; i += 64) {
        const int ci = comp_[i];
        if (ci >= 0 && ci != NO_ROOT) { const int cc = comp_[ci]; if (cc < ci) { comp_[i] = cc; moved = true; } }
      }
      __syncthreads();
      if (!__any(moved)) break;
    }
  };
  flatten();
  for (int round = 0; round < RMAX; ++round) {
    int cw[WORDS];
    unsigned long long rootmask[WORDS];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      cw[w] = comp_[w * 64 + lane];
      rootmask[w] = __ballot(cw[w] == w * 64 + lane);
      touch[w] = NO_ROOT;
    }
#pragma unroll
    for (int wr = 0; wr < WORDS; ++wr) {
      if (wr >= NW) continue;
      unsigned long long rm = rootmask[wr];            // wave-uniform
      while (rm) {
        const int r = wr * 64 + __ffsll((long long)rm) - 1;
        rm &= rm - 1;
        unsigned long long M[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; ++w) M[w] = __ballot(cw[w] == r);
#pragma unroll
        for (int k = 0; k < WORDS; ++k) {
          unsigned long long hit = 0ull;
#pragma unroll
          for (int w = 0; w < WORDS; ++w) hit |= row[k][w] & M[w];
          if (hit != 0ull && touch[k] == NO_ROOT) touch[k] = r;
        }
      }
    }
    bool changed = false;
#pragma unroll
    for (int k = 0; k < WORDS; ++k) {
      if (k >= NW) continue;
      const int own = cw[k];
      if (own >= 0 && own != NO_ROOT && touch[k] < own) { atomicMin(&comp_[own], touch[k]); changed = true; }
    }
    __syncthreads();
    if (!__any(changed)) break;
    flatten();
  }
  }      // (!chains)
  LIDAR_PHASE_END(4);
  // cluster root of every reading (of this lane's point of every word: nobody else asks for it): own component for cores,
  // smallest neighbouring core component for the rest
  int rootr[WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
    const int ci = (k < NW) ? comp_[k * 64 + lane] : -1;
    rootr[k] = (ci < 0) ? NO_ROOT : ((ci != NO_ROOT) ? ci : touch[k]);
  }
  LIDAR_PHASE_END(5);
  // roots in ascending order = cluster labels 0, 1, ...
  int n_clusters = 0;
  for (int w = 0; w < NW; ++w) {
    const int i = w * 64 + lane;
    const bool is_root = comp_[i] == i;
    const unsigned long long ball = __ballot(is_root);
    if (is_root) {
      const int k = n_clusters + __popcll(ball & ((1ull << lane) - 1ull));
      if (k < 64) roots_[k] = i;
    }
    n_clusters += __popcll(ball);
  }
  __syncthreads();
  if (labels_out) {
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      const int i = w * 64 + lane;
      if (i >= n_pts) continue;
      int lab = -1;                                           // -1 noise
      const int r = rootr[w];
      if (r != NO_ROOT) for (int k = 0; k < n_clusters && k < 64; ++k) if (roots_[k] == r) lab = k;
      labels_out[b * R + cand_[i]] = lab;
    }
  }

  LIDAR_PHASE_END(3);
  // ---- 3. convex hull per cluster (create_convex_hull) ------------------------------------------------
  // Four clusters at a time, one per 16-lane DPP row: a row walks its own cluster's member list (compacted below) and
  // every "best next vertex" reduction is four in-row DPP steps — no LDS crossbar, no cross-row traffic.  Vertices
  // are staged in LDS (the neighbour rows are dead by now) and committed in cluster order for proper polygons only.
  int n_out = 0, ovf = (n_clusters > 64 || in_ovf) ? 1 : 0;
  double* oxy = obs_xy ? obs_xy + b * (long)n_obs_max * v_max * 2 : nullptr;
  int32_t* onv = obs_nv ? obs_nv + b * (long)n_obs_max : nullptr;
  double* oce = c_eta ? c_eta + b * (long)n_obs_max * 4 : nullptr;
  if (onv) for (int k = lane; k < n_obs_max; k += 64) onv[k] = 0;
  if (oce) for (int k = lane; k < n_obs_max * 4; k += 64) oce[k] = 0.0;       // eta = (0, 0): empty slot
  const int nc = n_clusters < 64 ? n_clusters : 64;
  unsigned short* list_ = cand_;                        // member lists, cluster after cluster (labels are written)
  int* coff_ = comp_;                                   // coff_[k] .. coff_[k+1]: members of cluster k
  __syncthreads();
  {
    int off = 0;
    for (int k = 0; k < nc; ++k) {
      const int r = roots_[k];
      if (lane == 0) coff_[k] = off;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        if (w >= NW) continue;
        const bool m = rootr[w] == r;
        const unsigned long long ball = __ballot(m);
        if (m) list_[off + __popcll(ball & ((1ull << lane) - 1ull))] = (unsigned short)(w * 64 + lane);
        off += __popcll(ball);
      }
    }
    if (lane == 0) coff_[nc] = off;
  }
  __syncthreads();
  const int q = lane >> 4, l16 = lane & 15;
  auto row_best = [&](Cand& c, auto&& take_other) {     // butterfly over the 16 lanes of the row
    { Cand o; o.x = lipmpc_dev::row_xor<1>(c.x); o.y = lipmpc_dev::row_xor<1>(c.y); o.idx = lipmpc_dev::row_xor<1>(c.idx); if (take_other(c, o)) c = o; }
    { Cand o; o.x = lipmpc_dev::row_xor<2>(c.x); o.y = lipmpc_dev::row_xor<2>(c.y); o.idx = lipmpc_dev::row_xor<2>(c.idx); if (take_other(c, o)) c = o; }
    { Cand o; o.x = lipmpc_dev::row_xor<4>(c.x); o.y = lipmpc_dev::row_xor<4>(c.y); o.idx = lipmpc_dev::row_xor<4>(c.idx); if (take_other(c, o)) c = o; }
    { Cand o; o.x = lipmpc_dev::row_xor<8>(c.x); o.y = lipmpc_dev::row_xor<8>(c.y); o.idx = lipmpc_dev::row_xor<8>(c.idx); if (take_other(c, o)) c = o; }
  };
  auto wave_best = [&](Cand& c, auto&& take_other) {    // all 64 lanes: in-row DPP butterfly, then two cross-row steps
    row_best(c, take_other);
    { Cand o; o.x = wave_xor16(c.x); o.y = wave_xor16(c.y); o.idx = wave_xor16(c.idx); if (take_other(c, o)) c = o; }
    { Cand o; o.x = wave_xor32(c.x); o.y = wave_xor32(c.y); o.idx = wave_xor32(c.idx); if (take_other(c, o)) c = o; }
  };
  auto lex = [](const Cand& a, const Cand& o) {
    return (o.idx >= 0) & ((a.idx < 0) | (o.x < a.x) | ((o.x == a.x) & ((o.y < a.y) | ((o.y == a.y) & (o.idx < a.idx)))));
  };
  // One group of clusters: a lane fetches ITS members of its cluster once (point index from the member list, coordinates from
  // the point array: at most NJ of them) and every step of the march runs on registers and DPP alone -- the march is a chain
  // of dependent steps, and an LDS round trip per candidate and step was most of what a step cost.
  auto march_group = [&](auto nj_c, auto solo_c, int g, int ng) -> int {
    constexpr int NJ = decltype(nj_c)::value;
    constexpr bool SOLO = decltype(solo_c)::value;
    constexpr int W = SOLO ? 64 : 16;
    const int lw = SOLO ? lane : l16, qrow = SOLO ? 0 : q;
    const int k = g + qrow;
    const bool on = qrow < ng;
    const int beg = on ? coff_[k] : 0, end = on ? coff_[k + 1] : 0;
    // candidate slots in use (wave-uniform: the group's largest cluster decides): the unused ones cost nothing
    int njw = (end - beg + W - 1) / W;
    njw = max(njw, wave_xor16(njw)); njw = max(njw, wave_xor32(njw));
    const int nj = __builtin_amdgcn_readfirstlane(njw);
    Cand c[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int t = beg + lw + j * W;
      const bool have = t < end;
      const int i = have ? (int)list_[t] : 0;
      c[j].x = pint_[2 * i]; c[j].y = pint_[2 * i + 1]; c[j].idx = have ? i : -1;
    }
    // lexicographically smallest point of the cluster
    Cand st; st.idx = -1; st.x = 0.0; st.y = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) if (j < nj && lex(st, c[j])) st = c[j];
    if (SOLO) wave_best(st, lex); else row_best(st, lex);
    // Jarvis march (of the rows in lock step)
    double cxp = st.x, cyp = st.y;
    int cip = st.idx;
    int nvert = 0;
    bool done = !on;
    float ux = 0.0f, uy = -1.0f;       // the direction the march arrived along (the start is the lowest of the leftmost points)
    for (int step = 0; step <= v_max; ++step) {
      if (__all(done)) break;
      if (!done && nvert < VSTAGE && lw == 0) stagei_[qrow * VSTAGE + nvert] = (unsigned short)cip;
      if (!done) ++nvert;
      // the candidates as seen from the vertex the march stands on (points equal to it are never candidates): the offsets
      // are what every comparison of this step works on, and they are what travels through the reduction
      // The winner is first GUESSED: the candidate with the smallest turn from the edge the march came along, by a
      // single-precision key (s / (|s| + |t|), s and t the dot and cross product of that edge with the offset: decreasing in the
      // angle over [0, pi]) -- a reduction over two words per lane instead of five with a predicate at every stage -- and then
      // PROVED: no candidate of any lane beats it under the exact predicate.  The predicate is a strict total order with one
      // maximum, so a guess that passes IS the exact reduction's result; one that fails (candidates closer in angle than single
      // precision resolves, exact collinearity, duplicates) sends the wave through the exact reduction.
      Cand r[NJ];
      float key = -INFINITY;
      int kidx = -1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma clang fp contract(off)
        if (j >= nj) continue;
        r[j].x = c[j].x - cxp; r[j].y = c[j].y - cyp;
        r[j].idx = ((r[j].x == 0.0) & (r[j].y == 0.0)) ? -1 : c[j].idx;
        const float fx = (float)r[j].x, fy = (float)r[j].y;
        const float sdot = ux * fx + uy * fy, tcr = ux * fy - uy * fx;
        const float kj = r[j].idx < 0 ? -INFINITY : sdot * __builtin_amdgcn_rcpf(fabsf(sdot) + fabsf(tcr) + 1e-37f);
        if (kj > key) { key = kj; kidx = r[j].idx; }
      }
      auto key_step = [&](float ok, int oi) { const bool take = ok > key; key = take ? ok : key; kidx = take ? oi : kidx; };
      key_step(lipmpc_dev::row_xor<1>(key), lipmpc_dev::row_xor<1>(kidx));
      key_step(lipmpc_dev::row_xor<2>(key), lipmpc_dev::row_xor<2>(kidx));
      key_step(lipmpc_dev::row_xor<4>(key), lipmpc_dev::row_xor<4>(kidx));
      key_step(lipmpc_dev::row_xor<8>(key), lipmpc_dev::row_xor<8>(kidx));
      if (SOLO) { key_step(wave_xor16(key), wave_xor16(kidx)); key_step(wave_xor32(key), wave_xor32(kidx)); }
      // (lanes of a row may hold different guesses when keys tie: lane 0's is the row's)
      kidx = SOLO ? __builtin_amdgcn_readfirstlane(kidx) : lipmpc_dev::dpp0<0x150>(kidx);       // row_newbcast:0
      Cand best;
      {
#pragma clang fp contract(off)
        const int gi = kidx < 0 ? 0 : kidx;
        best.x = pint_[2 * gi] - cxp; best.y = pint_[2 * gi + 1] - cyp; best.idx = kidx;
        if (kidx < 0) { best.x = 0.0; best.y = 0.0; }
      }
      // (the proof, straight-line: a candidate strictly to the right of p -> guess beats it; one exactly in line with it --
      // other than the guess itself -- sends the wave to the full predicate)
      bool beaten = false, in_line = false;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma clang fp contract(off)
        if (j >= nj) continue;
        const double cr = best.x * r[j].y - best.y * r[j].x;
        const bool valid = r[j].idx >= 0;
        beaten |= valid & ((best.idx < 0) | (cr < 0.0));
        in_line |= valid & (cr == 0.0) & (r[j].idx != best.idx);
      }
      if (__any(in_line)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j < nj) beaten |= better_from(best, r[j]);
      }
      if (__any(beaten)) {                                               // the exact reduction
        best.idx = -1; best.x = 0.0; best.y = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j < nj && better_from(best, r[j])) best = r[j];
        auto right = [&](const Cand& a, const Cand& o) { return better_from(a, o); };
        if (SOLO) wave_best(best, right); else row_best(best, right);
      }
      const int bi = best.idx < 0 ? 0 : best.idx;
      const double bx = pint_[2 * bi], by = pint_[2 * bi + 1];           // where the winner is
      if (!done) {
        if (best.idx < 0) done = true;                                   // single (repeated) point
        else if (bx == st.x && by == st.y) done = true;                  // ring closed
        else { cxp = bx; cyp = by; cip = best.idx; ux = (float)best.x; uy = (float)best.y; }
      }
    }
    return nvert;
  };
  // Clusters are processed in order.  A large cluster (a wall seen over many rays) gets the whole wave, consecutive small
  // ones share a wave, one per 16-lane row.
  for (int g = 0; g < nc;) {
    const bool solo = coff_[g + 1] - coff_[g] >= SOLO_MIN;            // wave-uniform
    int ng = 1;
    if (!solo) while (ng < 4 && g + ng < nc && coff_[g + ng + 1] - coff_[g + ng] < SOLO_MIN) ++ng;
    const int nvert = solo ? march_group(std::integral_constant<int, WORDS>{}, std::true_type{}, g, 1)
                           : march_group(std::integral_constant<int, (SOLO_MIN + 14) / 16>{}, std::false_type{}, g, ng);
    __syncthreads();
    // < 3 extreme points = fewer than 3 unique points or a collinear cluster: the reference drops it (:70-76)
    for (int qq = 0; qq < ng; ++qq) {
      const int nv = __builtin_amdgcn_readlane(nvert, qq * 16);
      if (nv >= 3) {
        if (n_out >= n_obs_max || nv > v_max) ovf = 1;
        else {
          const unsigned short* ringi = stagei_ + qq * VSTAGE;
          if (oxy) for (int v = lane; v < nv * 2; v += 64) oxy[(long)n_out * v_max * 2 + v] = pint_[2 * ringi[v >> 1] + (v & 1)];
          if (onv && lane == 0) onv[n_out] = nv;
          if (oce) {
            // ---- 4. constraint assembly: closest point c and unit normal eta of this hull at the CoM, one edge per lane
            // (nv <= VSTAGE = 64); per edge the arithmetic of closest_point_normal, the nearest edge by a wave minimum with
            // the first edge winning ties (the sequential scan keeps the first strict minimum), inside = parity of the
            // crossing hits.  A zero-length edge or x == c is degenerate geometry: eta = NaN, the step reports DEGENERATE.
#pragma clang fp contract(off)
            const bool eon = lane < nv;
            const int ia = eon ? lane : 0, ib = (ia + 1 == nv) ? 0 : ia + 1, ip = (ia == 0) ? nv - 1 : ia - 1;
            const double* vp = pint_ + 2 * ringi[ip];
            const double* va = pint_ + 2 * ringi[ia];
            const double* vb = pint_ + 2 * ringi[ib];
            const EdgeCp ec = edge_closest(vp[0], vp[1], va[0], va[1], vb[0], vb[1], x0, y0);
            double dmin = eon ? ec.d : INFINITY;
            dmin = fmin(dmin, lipmpc_dev::row_xor<1>(dmin)); dmin = fmin(dmin, lipmpc_dev::row_xor<2>(dmin));
            dmin = fmin(dmin, lipmpc_dev::row_xor<4>(dmin)); dmin = fmin(dmin, lipmpc_dev::row_xor<8>(dmin));
            dmin = fmin(dmin, wave_xor16(dmin)); dmin = fmin(dmin, wave_xor32(dmin));
            const int sel = __ffsll((long long)__ballot(eon && ec.d == dmin)) - 1;      // >= 0: lane 0 is always an edge
            const double ccx = lane_value(ec.qx, sel), ccy = lane_value(ec.qy, sel);
            const bool inside = (__popcll(__ballot(eon && ec.hit)) & 1) != 0;
            bool degen = __any(eon && ec.degen);
            double nx = x0 - ccx, ny = y0 - ccy;
            const double nn = sqrt(nx * nx + ny * ny);
            if (!(nn > 0.0)) degen = true;
            nx = nx / nn; ny = ny / nn;
            if (inside) { nx = -nx; ny = -ny; }
            if (lane == 0) {
              double* o = oce + (long)n_out * 4;
              o[0] = ccx; o[1] = ccy; o[2] = degen ? NAN : nx; o[3] = degen ? NAN : ny;
            }
          }
          ++n_out;
        }
      }
    }
    __syncthreads();
    g += ng;
  }
  if (lane == 0) { n_inferred[b] = n_out; overflow[b] = ovf; }
#ifdef LIPMPC_LIDAR_PHASES
  if (dbg_stop == 10 && lane == 0) n_inferred[b] = chains ? 1 : 0;      // which route clustered this scan (tools/lidar_wave_times.py)
  if (dbg_stop == 9 && lane == 0) {      // wave timing (tools/lidar_wave_times.py): start and end on the 100 MHz wall clock, by robot
    n_inferred[b] = (int)(t_enter & 0x7fffffff);
    overflow[b] = (int)(wall_clock64() & 0x7fffffff);
  }
#endif
}

// Weight of every robot for the launch order of its scan: an ESTIMATE of its reading count -- every ray, if the robot stands
// INSIDE an obstacle (the heaviest scans there are: all 360 readings in one dense cluster); otherwise, per obstacle in range, the
// rays its bounding circle of radius r at distance d subtends, R / (2 pi) * 2 asin(r / d) (the asin by its argument: the estimate
// only ranks), scaled by the share of the circle inside the range -- summed and capped at R.  Correlation 0.91 with the true
// counts on a CROWDED-style map, 93 % of the heaviest tenth in its top fifth (tools/lidar_order.py): it costs one obstacle per
// lane instead of the scan itself.  One wave per robot.
__global__ __launch_bounds__(64) void lidar_weight_kernel(long B, int R, int n_env, int v_env, long env_stride, double lidar_range,
                                                          const double* __restrict__ state, const double* __restrict__ env_xy,
                                                          const int32_t* __restrict__ env_nv, int32_t* __restrict__ sched) {
  const long b = blockIdx.x;
  if (b >= B) return;
  const int lane = threadIdx.x;
  const double x0 = state[b * 5 + 0], y0 = state[b * 5 + 2];
  const double* exy = env_xy + b * env_stride * (long)n_env * v_env * 2;
  const int32_t* env = env_nv + b * env_stride * (long)n_env;
  double w = 0.0;
  for (int j = lane; j < n_env; j += 64) {
    const int nv = min(env[j], v_env);
    if (nv <= 0) continue;
    const double* ring = exy + (long)j * v_env * 2;
    double mx = 0.0, my = 0.0, rad2 = 0.0;
    bool within = false;             // the robot stands INSIDE this obstacle: every ray returns a reading (the heaviest scans there are)
    auto crosses = [&](double ax, double ay, double bx, double by) {      // edge a -> b against the +x ray from the robot
      return ((ay > y0) != (by > y0)) && (x0 - ax) * fabs(by - ay) < (bx - ax) * (y0 - ay) * ((by > ay) ? 1.0 : -1.0);
    };
    if (v_env <= VFAST) {            // the whole ring in one round of loads
      double vx[VFAST], vy[VFAST];
#pragma unroll
      for (int e = 0; e < VFAST; ++e) { const int ee = e < nv ? e : 0; vx[e] = ring[2 * ee]; vy[e] = ring[2 * ee + 1]; }
#pragma unroll
      for (int e = 0; e < VFAST; ++e) if (e < nv) { mx += vx[e]; my += vy[e]; }
      mx /= nv; my /= nv;
#pragma unroll
      for (int e = 0; e < VFAST; ++e)
        if (e < nv) {
          const double dx = vx[e] - mx, dy = vy[e] - my;
          rad2 = fmax(rad2, dx * dx + dy * dy);
          const bool last = e + 1 == nv;
          within ^= crosses(vx[e], vy[e], last ? vx[0] : vx[(e + 1) % VFAST], last ? vy[0] : vy[(e + 1) % VFAST]);
        }
    } else {
      for (int e = 0; e < nv; ++e) { mx += ring[2 * e]; my += ring[2 * e + 1]; }
      mx /= nv; my /= nv;
      for (int e = 0; e < nv; ++e) {
        const double dx = ring[2 * e] - mx, dy = ring[2 * e + 1] - my;
        rad2 = fmax(rad2, dx * dx + dy * dy);
        const int f = e + 1 == nv ? 0 : e + 1;
        within ^= crosses(ring[2 * e], ring[2 * e + 1], ring[2 * f], ring[2 * f + 1]);
      }
    }
    const double rad = sqrt(rad2), d = sqrt((mx - x0) * (mx - x0) + (my - y0) * (my - y0));
    if (d > lidar_range + rad) continue;
    const double inside = (d + rad <= lidar_range) ? 1.0 : fmin(1.0, fmax(0.0, (lidar_range + rad - d) / (2.0 * rad + 1e-300)));
    w += within ? (double)R : (double)R * (1.0 / M_PI) * fmin(1.0, rad / fmax(d, 1e-300)) * inside;
  }
  for (int m = 1; m < 64; m <<= 1) w += __shfl_xor(w, m, 64);
  if (lane == 0) sched[SCHED_ORDER + B + b] = (int32_t)fmin(w, (double)R);
}

// The launch order of the scans from the weights: robots by descending weight (counting sort, one workgroup; which of two
// equally heavy robots comes first is immaterial).
__global__ __launch_bounds__(1024) void lidar_order_kernel(long B, int period, int32_t* __restrict__ sched) {
  constexpr int NBIN = 512;                                // >= RMAX + 1 weights; one thread per bin in the scan
  static_assert(RMAX + 1 <= NBIN, "one bin per weight");
  __shared__ int cursor_[NBIN];
  __shared__ int scan_[NBIN];
  const int32_t* w = sched + SCHED_ORDER + B;
  int32_t* order = sched + SCHED_ORDER;
  for (int k = threadIdx.x; k < NBIN; k += blockDim.x) cursor_[k] = 0;
  __syncthreads();
  for (long i = threadIdx.x; i < B; i += blockDim.x) atomicAdd(&cursor_[RMAX - min(max(w[i], 0), RMAX)], 1);
  __syncthreads();
  // exclusive prefix sum of the bin counts (Hillis-Steele over the first NBIN threads; a serial loop of one thread was most of
  // this kernel's time)
  const int t = threadIdx.x;
  int mine = t < NBIN ? cursor_[t] : 0, acc = mine;
  for (int d = 1; d < NBIN; d <<= 1) {
    if (t < NBIN) scan_[t] = acc;
    __syncthreads();
    if (t < NBIN && t >= d) acc += scan_[t - d];
    __syncthreads();
  }
  if (t < NBIN) cursor_[t] = acc - mine;
  __syncthreads();
  // Rank -> launch position.  While every wave of the grid is resident at once, launch positions `period` apart share a SIMD
  // (period = 4 x compute units: the dispatcher deals single-wave blocks round-robin, tools/lidar_placement.py), so the ranks go
  // out boustrophedon -- forwards on even rounds of `period` positions, backwards on odd ones: a SIMD's robots are one from each
  // quantile, the heaviest with the lightest.  (Beyond what is resident the dispatcher takes blocks as slots free up, and plain
  // heaviest-first is the right order there: the last, partial round stays as ranked.)
  for (long i = threadIdx.x; i < B; i += blockDim.x) {
    const long r = atomicAdd(&cursor_[RMAX - min(max(w[i], 0), RMAX)], 1);
    const long q = r / period, s2 = r % period;
    const long pos = ((q + 1) * period <= B && (q & 1)) ? q * period + (period - 1 - s2) : r;
    order[pos] = (int32_t)i;
  }
  if (threadIdx.x == 0) { sched[SCHED_VALID] = (int32_t)B; sched[SCHED_PERIOD] = period; }
}

}  // namespace

// SIMDs of the device (4 per compute unit), asked once per device
static int simd_count(int device) {
  static int cached[64];
  if (device < 0 || device >= 64) return 1024;
  if (cached[device] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    cached[device] = 4 * cus;
  }
  return cached[device];
}

static int lidar_launch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env, int32_t env_shared,
                        double lidar_range, double eps, int32_t min_samples, int32_t n_obs_max, int32_t v_max,
                        const double* state, const double* env_xy, const int32_t* env_nv, const double* ray_table,
                        const double* noise, double* obs_xy, int32_t* obs_nv, double* c_eta, int32_t* n_inferred,
                        int32_t* overflow, double* hits, int32_t* labels, int32_t* schedule, void* hip_stream) {
  if (B < 0 || resolution < 1 || resolution > RMAX || n_env < 0 || v_env < 1 || n_obs_max < 1 || v_max < 3 || v_max > VSTAGE) return LIPMPC_E_ARG;
  if (n_env > 65535) return LIPMPC_E_UNSUPPORTED;      // obstacle indices are kept as 16 bits in LDS
  if (B == 0) return LIPMPC_OK;
  if (!state || !ray_table || !n_inferred || !overflow || (n_env > 0 && (!env_xy || !env_nv)) || (!obs_xy != !obs_nv) ||
      (!obs_xy && !c_eta))
    return LIPMPC_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return LIPMPC_E_HIP;
#ifdef LIPMPC_LIDAR_PHASES
  // profiling build only (make CXXFLAGS+=-DLIPMPC_LIDAR_PHASES, tools/lidar_phases.py): LIPMPC_LIDAR_STOP=1..7 ends the
  // kernel after that phase; outputs are then undefined.  The shipped library has no such knob.
  static const int dbg_stop = getenv("LIPMPC_LIDAR_STOP") ? atoi(getenv("LIPMPC_LIDAR_STOP")) : 0;
#else
  const int dbg_stop = 0;
#endif
  if (schedule && n_env > 0) {
    // rank the robots first: estimate of the reading counts -> launch positions, a heavy robot with light ones on every SIMD
    hipLaunchKernelGGL(lidar_weight_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)hip_stream, (long)B, resolution, n_env, v_env,
                       (long)(env_shared ? 0 : 1), lidar_range, state, env_xy, env_nv, schedule);
    hipLaunchKernelGGL(lidar_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, (long)B, simd_count(device), schedule);
  } else {
    schedule = nullptr;
  }
  hipLaunchKernelGGL(lidar_sense_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)hip_stream, (long)B, resolution, n_env,
                     v_env, (long)(env_shared ? 0 : 1), lidar_range, eps, min_samples, n_obs_max, v_max, state, env_xy, env_nv,
                     ray_table, noise, obs_xy, obs_nv, c_eta, n_inferred, overflow, hits, labels, schedule, dbg_stop);
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}

extern "C" int lipmpc_lidar_sense_batch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                                        int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                                        int32_t n_obs_max, int32_t v_max, const double* state, const double* env_xy,
                                        const int32_t* env_nv, const double* ray_table, const double* noise,
                                        double* obs_xy, int32_t* obs_nv, int32_t* n_inferred, int32_t* overflow,
                                        double* hits, int32_t* labels, void* hip_stream) {
  if (!obs_xy || !obs_nv) return LIPMPC_E_ARG;
  return lidar_launch(device, B, resolution, n_env, v_env, env_shared, lidar_range, eps, min_samples, n_obs_max, v_max, state,
                      env_xy, env_nv, ray_table, noise, obs_xy, obs_nv, nullptr, n_inferred, overflow, hits, labels, nullptr, hip_stream);
}

extern "C" int lipmpc_lidar_c_eta_batch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                                        int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                                        int32_t n_obs_max, int32_t v_max, const double* state, const double* env_xy,
                                        const int32_t* env_nv, const double* ray_table, const double* noise,
                                        double* c_eta, int32_t* n_inferred, int32_t* overflow, double* obs_xy,
                                        int32_t* obs_nv, double* hits, int32_t* labels, int32_t* schedule,
                                        void* hip_stream) {
  if (!c_eta) return LIPMPC_E_ARG;
  if (schedule && B > 0x3fffffff) return LIPMPC_E_UNSUPPORTED;
  return lidar_launch(device, B, resolution, n_env, v_env, env_shared, lidar_range, eps, min_samples, n_obs_max, v_max, state,
                      env_xy, env_nv, ray_table, noise, obs_xy, obs_nv, c_eta, n_inferred, overflow, hits, labels, schedule, hip_stream);
}

extern "C" int64_t lipmpc_lidar_schedule_words(int64_t B) { return B < 0 ? LIPMPC_E_ARG : SCHED_ORDER + 2L * B; }
