// lipmpc_lidar.hip — unknown-environment front end (BASELINE config 5): per robot and MPC step, a 2-D LiDAR scan of the
// true map -> noisy readings -> DBSCAN clusters -> convex hull per cluster, i.e. the obstacle rings the step kernel
// consumes.  Restates
//   range_finder / compute_lidar_readings / retrieve_clusters / build_local_obstacles
//       HumanoidNavigation/RangeFinder/range_finder_wth_polygons_dbscan.py:26-63, 65-83, 100-126, 157-180
//   line_polygon_intersection (compute_intersection)      HumanoidNavigation/Utils/obstacles.py:95-139
//   the call site                                          HumanoidNavigation/MPC/HumanoidMPCVariants/HumanoidMPCUnknownEnvironment.py:30-68
// One wavefront (64 lanes) per robot; everything between the ray casting and the rings stays in LDS:
//   1. rays: lane l owns rays l, l+64, ...; every ray walks obstacles in list order and edges in ring order and keeps
//      the nearest hit strictly inside the range (contraction off: the hit points are bit-identical to the reference's)
//   2. DBSCAN(eps, min_samples) by its order-free characterisation (oracle/lidar_oracle.py), on the readings
//      compacted in ray order: neighbour bit rows, core flags, connected components of the core points (forest of
//      "smallest core neighbour" pointers + pointer jumping, then merging trees through ballot masks of tree
//      membership — bit operations, no sweeps over neighbours' labels), clusters numbered by their smallest core
//      index, border points to the smallest neighbouring cluster
//   3. hull per cluster: Jarvis march from the lexicographically smallest point, farthest point on collinear ties
//      (= the CCW ring of extreme points Qhull / monotone chain return, same rotation as np.unique + monotone chain);
//      four clusters march at once, one per 16-lane DPP row, over compacted member lists
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "lipmpc_kernel.hpp"      // DPP row exchanges (lipmpc_dev::row_xor)

namespace {

constexpr int RMAX = 384;            // rays per scan (reference: 360)
constexpr int WORDS = RMAX / 64;     // neighbour bit row
constexpr int NO_ROOT = 0x7fffffff;
constexpr int SOLO_MIN = 48;        // a cluster of at least this many points gets the whole wave in the hull stage
constexpr int NCC = 64;             // candidates whose bounding circle is kept for the per-pass sector test
constexpr int VSTAGE = 64;          // hull vertices staged per cluster (v_max <= VSTAGE)
static_assert(4 * VSTAGE * 2 <= 2 * RMAX, "hull staging reuses the point arrays");

struct Cand { double x, y; int idx; };


// is candidate b a better "next hull vertex" than a when standing on p?  (b strictly to the right of p->a, or
// collinear and farther; a.idx < 0 = no candidate yet; points equal to p are never candidates)
__device__ __forceinline__ bool better(double px, double py, const Cand& a, const Cand& b) {
#pragma clang fp contract(off)
  if (b.idx < 0) return false;
  if (a.idx < 0) return true;
  const double cr = (a.x - px) * (b.y - py) - (a.y - py) * (b.x - px);
  if (cr < 0.0) return true;
  if (cr > 0.0) return false;
  const double da = (a.x - px) * (a.x - px) + (a.y - py) * (a.y - py);
  const double db = (b.x - px) * (b.x - px) + (b.y - py) * (b.y - py);
  return db > da || (db == da && b.idx < a.idx);
}

__global__ __launch_bounds__(64) void lidar_sense_kernel(
    long B, int R, int n_env, int v_env, long env_stride, double lidar_range, double eps, int min_samples,
    int n_obs_max, int v_max, const double* __restrict__ state, const double* __restrict__ env_xy,
    const int32_t* __restrict__ env_nv, const double* __restrict__ ray_table, const double* __restrict__ noise,
    double* __restrict__ obs_xy, int32_t* __restrict__ obs_nv, int32_t* __restrict__ n_inferred,
    int32_t* __restrict__ overflow, double* __restrict__ hits_out, int32_t* __restrict__ labels_out, int dbg_stop) {
  __shared__ double pxy_[2 * RMAX];
  double* const px_ = pxy_;
  double* const py_ = pxy_ + RMAX;
  __shared__ __attribute__((aligned(16))) int comp_[RMAX];                    // -1 = no reading; core: component root; else NO_ROOT
  __shared__ int root_[RMAX];                    // cluster root of every reading (NO_ROOT = noise)
  __shared__ double cx_[RMAX], cy_[RMAX];       // points in member-list order (hull stage)
  __shared__ int roots_[64];
  __shared__ double candc_[NCC][3];              // bounding circle (centre, radius) of the first NCC candidate obstacles
  __shared__ double bb_[WORDS][4];               // bounding box (x0, x1, y0, y1) of the 64 points of each word
  __shared__ int cand_[RMAX];                    // obstacles that can be hit from here, list order

  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  if (b >= B) return;
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
      for (int e = 0; e < nv; ++e) { mx += ring[2 * e]; my += ring[2 * e + 1]; }
      if (nv > 0) {
        mx /= nv; my /= nv;
        for (int e = 0; e < nv; ++e) rad = fmax(rad, hypot(ring[2 * e] - mx, ring[2 * e + 1] - my));
        keep = hypot(mx - x0, my - y0) <= (lidar_range + rad) * (1.0 + 1e-9) + 1e-9;
      }
    }
    const unsigned long long ball = __ballot(keep);
    if (keep) {
      const int k = n_cand + __popcll(ball & ((1ull << lane) - 1ull));
      if (k < RMAX) cand_[k] = j;
      if (k < NCC) { candc_[k][0] = mx; candc_[k][1] = my; candc_[k][2] = rad; }
    }
    n_cand += __popcll(ball);
  }
  // cand_ holds RMAX obstacles: the (RMAX+1)-th obstacle in range is dropped and the scan flagged (overflow), never
  // read past the list
  if (n_cand > RMAX) { n_cand = RMAX; in_ovf = 1; }
  in_ovf = __any(in_ovf) ? 1 : 0;
  __syncthreads();
  for (int i = lane; i < RMAX; i += 64) {
    bool have = false;
    double hx = 0.0, hy = 0.0;
    if (i < R) {
#pragma clang fp contract(off)
      const double ex = x0 + lidar_range * ray_table[2 * i], ey = y0 + lidar_range * ray_table[2 * i + 1];
      const double rdx = ex - x0, rdy = ey - y0;               // b1 - a1
      double best_d = lidar_range;
      const double inv_len2 = __builtin_amdgcn_rcp(rdx * rdx + rdy * rdy);
      for (int jc = 0; jc < n_cand; ++jc) {
        // sector test: the 64 rays of this pass span 64 degrees; an obstacle none of them comes near (distance from
        // its bounding circle's centre to the ray segment > radius, with a margin far above the rounding of this
        // estimate) is skipped by the whole wave — its edges could not have produced a hit for any of these rays
        if (jc < NCC) {
          const double wx = candc_[jc][0] - x0, wy = candc_[jc][1] - y0, cr = candc_[jc][2] + 1e-6;
          const double tt = fmin(1.0, fmax(0.0, (wx * rdx + wy * rdy) * inv_len2));
          const double ddx = wx - tt * rdx, ddy = wy - tt * rdy;
          if (!__any(ddx * ddx + ddy * ddy <= cr * cr)) continue;
        }
        const int j = cand_[jc];
        const int nv = min(env[j], v_env);
        const double* ring = exy + (long)j * v_env * 2;
        bool chave = false;
        double cx = 0.0, cy = 0.0, cd = lidar_range;
        for (int e = 0; e < nv; ++e) {
          const double a2x = ring[2 * e], a2y = ring[2 * e + 1];
          const int e1 = (e + 1 == nv) ? 0 : e + 1;
          const double b2x = ring[2 * e1], b2y = ring[2 * e1 + 1];
          const double denom = (b2y - a2y) * rdx - (b2x - a2x) * rdy;
          if (denom == 0.0) continue;
          const double nua = (b2x - a2x) * (y0 - a2y) - (b2y - a2y) * (x0 - a2x);
          const double nub = rdx * (y0 - a2y) - rdy * (x0 - a2x);
          // cheap conservative prefilter (no division): clearly outside [0,1] -> next edge; the reference's exact
          // division test decides everything that survives
          const double ad = fabs(denom), sa = (denom > 0.0) ? nua : -nua, sb = (denom > 0.0) ? nub : -nub;
          const double slack = ad * 1e-12;
          if (sa < -slack || sb < -slack || sa > ad + slack || sb > ad + slack) continue;
          const double ua = nua / denom;
          const double ub = nub / denom;
          if (ua >= 0.0 && ua <= 1.0 && ub >= 0.0 && ub <= 1.0) {
            const double qx = x0 + ua * rdx, qy = y0 + ua * rdy;
            const double dd = sqrt((qx - x0) * (qx - x0) + (qy - y0) * (qy - y0));
            if (dd < cd) { cd = dd; cx = qx; cy = qy; chave = true; }
          }
        }
        if (chave && cd <= lidar_range && cd < best_d) { best_d = cd; hx = cx; hy = cy; have = true; }
      }
      if (have && noise) { hx = hx + noise[(b * R + i) * 2]; hy = hy + noise[(b * R + i) * 2 + 1]; }
    }
    px_[i] = hx; py_[i] = hy;
    comp_[i] = have ? NO_ROOT : -1;
    if (hits_out && i < R) { hits_out[(b * R + i) * 2] = have ? hx : NAN; hits_out[(b * R + i) * 2 + 1] = have ? hy : NAN; }
  }
  __syncthreads();

#ifdef LIPMPC_LIDAR_PHASES
#define LIDAR_PHASE_END(n) if (dbg_stop == (n)) return
#else
#define LIDAR_PHASE_END(n)
#endif
  LIDAR_PHASE_END(1);
  // ---- 2. DBSCAN ------------------------------------------------------------------------------------
  // Readings are first compacted in ray order (typically 110-200 of 360 rays return one): clustering and hulls then
  // sweep n points instead of RMAX slots.  Order is preserved, so "smallest core index" numbering, border-point
  // assignment and every index tie-break are those of the uncompacted scan.  In place: slot k <= i always, one
  // word of 64 rays is read by the whole wave before its survivors are written back.
  int n_pts = 0;
  for (int w = 0; w < WORDS; ++w) {
    const int i = w * 64 + lane;
    const bool have = comp_[i] >= 0;
    const double hx = px_[i], hy = py_[i];
    const unsigned long long ball = __ballot(have);
    __syncthreads();
    if (have) {
      const int k = n_pts + __popcll(ball & ((1ull << lane) - 1ull));
      px_[k] = hx; py_[k] = hy; cand_[k] = i;               // cand_ is free after the ray casting: ray of point k
    }
    n_pts += __popcll(ball);
    __syncthreads();
  }
  const int NW = (n_pts + 63) >> 6;                      // words / passes actually in use (wave-uniform)
  const int npad = NW << 6;
  for (int i = lane; i < RMAX; i += 64) comp_[i] = (i < n_pts) ? NO_ROOT : -1;
  if (labels_out) for (int i = lane; i < R; i += 64) labels_out[b * R + i] = -2;      // -2 = no reading
  __syncthreads();
  const double eps2 = eps * eps;
  unsigned long long vmask[WORDS];                  // which points exist
#pragma unroll
  for (int w = 0; w < WORDS; ++w) {
    const int left = n_pts - w * 64;
    vmask[w] = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
  }
  // row[k][w]: neighbour bits of point lane + 64 k against the 64 points of word w — kept in registers (the lane
  // that computes a row is the only one that reads it), which keeps the LDS footprint at 19 KB = 8 waves per CU
  unsigned long long row[WORDS][WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
#pragma unroll
    for (int w = 0; w < WORDS; ++w) row[k][w] = 0ull;
  }
  // bounding boxes of the words: two words whose boxes are more than eps apart along an axis hold no neighbour pair,
  // and the whole 64 x 64 block is skipped by the wave (clusters are contiguous runs of rays, so most off-diagonal
  // blocks go)
  for (int w = 0; w < NW; ++w) {
    const int i = w * 64 + lane;
    const bool vi = comp_[i] >= 0;
    double x0 = vi ? px_[i] : INFINITY, x1 = vi ? px_[i] : -INFINITY, y0 = vi ? py_[i] : INFINITY, y1 = vi ? py_[i] : -INFINITY;
    for (int m = 1; m < 64; m <<= 1) {
      x0 = fmin(x0, __shfl_xor(x0, m, 64)); x1 = fmax(x1, __shfl_xor(x1, m, 64));
      y0 = fmin(y0, __shfl_xor(y0, m, 64)); y1 = fmax(y1, __shfl_xor(y1, m, 64));
    }
    if (lane == 0) { bb_[w][0] = x0; bb_[w][1] = x1; bb_[w][2] = y0; bb_[w][3] = y1; }
  }
  __syncthreads();
  for (int k = 0; k < NW; ++k) {
    const int i = k * 64 + lane;
    int cnt = 0;
    const bool vi = comp_[i] >= 0;
    const double xi = px_[i], yi = py_[i];
    const double kx0 = bb_[k][0], kx1 = bb_[k][1], ky0 = bb_[k][2], ky1 = bb_[k][3];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      if (w >= NW) continue;
      if (bb_[w][0] - kx1 > eps || kx0 - bb_[w][1] > eps || bb_[w][2] - ky1 > eps || ky0 - bb_[w][3] > eps) continue;   // wave-uniform
      // counted, unrolled sweep over all 64 slots of the word (loads pipeline; absent points are masked after)
      unsigned long long bits = 0ull;
#pragma unroll 16
      for (int kk = 0; kk < 64; ++kk) {
#pragma clang fp contract(off)
        const double dx = xi - px_[w * 64 + kk], dy = yi - py_[w * 64 + kk];
        bits |= (unsigned long long)(dx * dx + dy * dy <= eps2) << kk;
      }
      bits = vi ? (bits & vmask[w]) : 0ull;
      cnt += __popcll(bits);
#pragma unroll
      for (int k2 = 0; k2 < WORDS; ++k2) if (k2 == k) row[k2][w] = bits;     // k is wave-uniform
    }
    root_[i] = (vi && cnt >= min_samples) ? i : NO_ROOT;      // core points start as their own root
  }
  __syncthreads();
  for (int i = lane; i < npad; i += 64) if (comp_[i] >= 0) comp_[i] = root_[i];
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
  int touch[WORDS];                                  // smallest tree root point lane + 64 k touches (NO_ROOT: none)
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
  LIDAR_PHASE_END(4);
  // cluster root of every reading: own component for cores, smallest neighbouring core component for the rest
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
    if (k >= NW) continue;
    const int i = k * 64 + lane;
    const int ci = comp_[i];
    root_[i] = (ci < 0) ? NO_ROOT : ((ci != NO_ROOT) ? ci : touch[k]);
  }
  __syncthreads();
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
    for (int i = lane; i < n_pts; i += 64) {
      int lab = -1;                                           // -1 noise
      const int r = root_[i];
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
  double* oxy = obs_xy + b * (long)n_obs_max * v_max * 2;
  int32_t* onv = obs_nv + b * (long)n_obs_max;
  for (int k = lane; k < n_obs_max; k += 64) onv[k] = 0;
  const int nc = n_clusters < 64 ? n_clusters : 64;
  int* list_ = cand_;                                   // member lists, cluster after cluster (labels are written)
  int* coff_ = comp_;                                   // coff_[k] .. coff_[k+1]: members of cluster k
  double* stage_ = pxy_;                                // [4][VSTAGE][2]: the points live on in cx_/cy_ from here
  __syncthreads();
  {
    int off = 0;
    for (int k = 0; k < nc; ++k) {
      const int r = roots_[k];
      if (lane == 0) coff_[k] = off;
      for (int w = 0; w < NW; ++w) {
        const int i = w * 64 + lane;
        const bool m = root_[i] == r;
        const unsigned long long ball = __ballot(m);
        if (m) {
          const int pos = off + __popcll(ball & ((1ull << lane) - 1ull));
          list_[pos] = i; cx_[pos] = px_[i]; cy_[pos] = py_[i];
        }
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
    { Cand o; o.x = __shfl_xor(c.x, 16, 64); o.y = __shfl_xor(c.y, 16, 64); o.idx = __shfl_xor(c.idx, 16, 64); if (take_other(c, o)) c = o; }
    { Cand o; o.x = __shfl_xor(c.x, 32, 64); o.y = __shfl_xor(c.y, 32, 64); o.idx = __shfl_xor(c.idx, 32, 64); if (take_other(c, o)) c = o; }
  };
  // Clusters are processed in order.  A large cluster (a wall seen over many rays) gets the whole wave — its member
  // scan is what a march step costs —, consecutive small ones share a wave, one per 16-lane row.
  for (int g = 0; g < nc;) {
    const bool solo = coff_[g + 1] - coff_[g] >= SOLO_MIN;            // wave-uniform
    int ng = 1;
    if (!solo) while (ng < 4 && g + ng < nc && coff_[g + ng + 1] - coff_[g + ng] < SOLO_MIN) ++ng;
    const int W = solo ? 64 : 16;
    const int lw = solo ? lane : l16, qrow = solo ? 0 : q;
    const int k = g + qrow;
    const bool on = qrow < ng;
    const int beg = on ? coff_[k] : 0, end = on ? coff_[k + 1] : 0;
    // lexicographically smallest point of the cluster
    Cand st; st.idx = -1; st.x = 0.0; st.y = 0.0;
    for (int t = beg + lw; t < end; t += W) {
      const int i = list_[t];
      const double x = cx_[t], y = cy_[t];
      if (st.idx < 0 || x < st.x || (x == st.x && (y < st.y || (y == st.y && i < st.idx)))) { st.x = x; st.y = y; st.idx = i; }
    }
    auto lex = [](const Cand& a, const Cand& o) {
      return o.idx >= 0 && (a.idx < 0 || o.x < a.x || (o.x == a.x && (o.y < a.y || (o.y == a.y && o.idx < a.idx))));
    };
    if (solo) wave_best(st, lex); else row_best(st, lex);
    // Jarvis march (of the rows in lock step)
    double cxp = st.x, cyp = st.y;
    int nvert = 0;
    bool done = !on;
    for (int step = 0; step <= v_max; ++step) {
      if (__all(done)) break;
      if (!done && nvert < VSTAGE && lw == 0) { stage_[(qrow * VSTAGE + nvert) * 2] = cxp; stage_[(qrow * VSTAGE + nvert) * 2 + 1] = cyp; }
      if (!done) ++nvert;
      Cand best; best.idx = -1; best.x = 0.0; best.y = 0.0;
      for (int t = beg + lw; t < end; t += 2 * W) {               // two independent candidates per trip
        const int t2 = t + W;
        const bool two = t2 < end;
        Cand c1, c2;
        c1.x = cx_[t]; c1.y = cy_[t]; c1.idx = list_[t];
        c2.x = cx_[two ? t2 : t]; c2.y = cy_[two ? t2 : t]; c2.idx = two ? list_[t2] : -1;
        if (!(c1.x == cxp && c1.y == cyp) && better(cxp, cyp, best, c1)) best = c1;
        if (two && !(c2.x == cxp && c2.y == cyp) && better(cxp, cyp, best, c2)) best = c2;
      }
      auto right = [&](const Cand& a, const Cand& o) { return better(cxp, cyp, a, o); };
      if (solo) wave_best(best, right); else row_best(best, right);
      if (!done) {
        if (best.idx < 0) done = true;                                   // single (repeated) point
        else if (best.x == st.x && best.y == st.y) done = true;          // ring closed
        else { cxp = best.x; cyp = best.y; }
      }
    }
    __syncthreads();
    // < 3 extreme points = fewer than 3 unique points or a collinear cluster: the reference drops it (:70-76)
    for (int qq = 0; qq < ng; ++qq) {
      const int nv = __shfl(nvert, qq * 16, 64);
      if (nv >= 3) {
        if (n_out >= n_obs_max || nv > v_max) ovf = 1;
        else {
          for (int v = lane; v < nv * 2; v += 64) oxy[(long)n_out * v_max * 2 + v] = stage_[qq * VSTAGE * 2 + v];
          if (lane == 0) onv[n_out] = nv;
          ++n_out;
        }
      }
    }
    __syncthreads();
    g += ng;
  }
  if (lane == 0) { n_inferred[b] = n_out; overflow[b] = ovf; }
}

}  // namespace

extern "C" int lipmpc_lidar_sense_batch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                                        int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                                        int32_t n_obs_max, int32_t v_max, const double* state, const double* env_xy,
                                        const int32_t* env_nv, const double* ray_table, const double* noise,
                                        double* obs_xy, int32_t* obs_nv, int32_t* n_inferred, int32_t* overflow,
                                        double* hits, int32_t* labels, void* hip_stream) {
  if (B < 0 || resolution < 1 || resolution > RMAX || n_env < 0 || v_env < 1 || n_obs_max < 1 || v_max < 3 || v_max > VSTAGE) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state || !ray_table || !obs_xy || !obs_nv || !n_inferred || !overflow || (n_env > 0 && (!env_xy || !env_nv)))
    return LIPMPC_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return LIPMPC_E_HIP;
#ifdef LIPMPC_LIDAR_PHASES
  // profiling build only (make CXXFLAGS+=-DLIPMPC_LIDAR_PHASES, tools/lidar_phases.py): LIPMPC_LIDAR_STOP=1..5 ends the
  // kernel after that phase; outputs are then undefined.  The shipped library has no such knob.
  static const int dbg_stop = getenv("LIPMPC_LIDAR_STOP") ? atoi(getenv("LIPMPC_LIDAR_STOP")) : 0;
#else
  const int dbg_stop = 0;
#endif
  hipLaunchKernelGGL(lidar_sense_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)hip_stream, (long)B, resolution, n_env,
                     v_env, (long)(env_shared ? 0 : 1), lidar_range, eps, min_samples, n_obs_max, v_max, state, env_xy, env_nv,
                     ray_table, noise, obs_xy, obs_nv, n_inferred, overflow, hits, labels, dbg_stop);
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}
