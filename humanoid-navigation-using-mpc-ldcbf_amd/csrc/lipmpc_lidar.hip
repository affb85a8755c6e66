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
  // number of readings, and with the whole batch resident the launch lasts as long as its most loaded SIMD: 94 us as the robots
  // come, 74 us ranked by the true counts, 85 us ranked by the estimate, ranking included (tools/lidar_order.py).  Any order
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

  // 1. ray casting -> n_pts readings (hit + noise) compacted in ray order in pint_, the ray of reading k in cand_[k]
#include "lipmpc_lidar_rays.inc"

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
  // 2a. clustering by chains of consecutive readings, where that is provably DBSCAN's answer -> chains, comp_, touch
#include "lipmpc_lidar_chains.inc"
  if (chains) __syncthreads();
  if (!chains) {
  // 2. the general route: neighbour rows, core flags, connected components -> comp_, touch
#include "lipmpc_lidar_rows.inc"
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
  // 3 + 4. hull per cluster, constraint assembly -> obs_xy / obs_nv / c_eta, n_out, ovf
#include "lipmpc_lidar_hulls.inc"
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
  // exclusive prefix sum of the bin counts: each of the first NBIN / 64 waves scans its 64 bins in registers, the waves' totals
  // go through LDS once (two barriers; a serial loop of one thread, then a Hillis-Steele scan with a barrier pair per step,
  // were most of this kernel's time)
  const int t = threadIdx.x;
  {
    const int mine = t < NBIN ? cursor_[t] : 0;
    int acc = mine;                                   // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(acc, d, 64); if ((t & 63) >= d) acc += o; }
    if (t < NBIN && (t & 63) == 63) scan_[t >> 6] = acc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < NBIN / 64; ++k) if (k < (t >> 6)) base += scan_[k];
    __syncthreads();
    if (t < NBIN) cursor_[t] = base + acc - mine;
  }
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
