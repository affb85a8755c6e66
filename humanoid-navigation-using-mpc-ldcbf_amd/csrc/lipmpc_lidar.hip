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
//   2. DBSCAN(eps, min_samples) by its order-free characterisation (oracle/lidar_oracle.py): neighbour bit rows,
//      core flags, connected components of core points by min-label propagation + pointer jumping, clusters numbered
//      by their smallest core index, border points to the smallest neighbouring cluster
//   3. hull per cluster: Jarvis march from the lexicographically smallest point, farthest point on collinear ties
//      (= the CCW ring of extreme points Qhull / monotone chain return, same rotation as np.unique + monotone chain)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/lipmpc.h"

namespace {

constexpr int RMAX = 384;            // rays per scan (reference: 360)
constexpr int WORDS = RMAX / 64;     // neighbour bit row
constexpr int NO_ROOT = 0x7fffffff;

struct Cand { double x, y; int idx; };

__device__ __forceinline__ double shfl_d(double v, int m) { return __shfl_xor(v, m, 64); }

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
    int32_t* __restrict__ overflow, double* __restrict__ hits_out, int32_t* __restrict__ labels_out) {
  __shared__ double px_[RMAX], py_[RMAX];
  __shared__ __attribute__((aligned(16))) int comp_[RMAX];                    // -1 = no reading; core: component root; else NO_ROOT
  __shared__ int root_[RMAX];                    // cluster root of every reading (NO_ROOT = noise)
  __shared__ unsigned long long nb_[RMAX][WORDS];
  __shared__ int roots_[64];
  __shared__ int cand_[RMAX];                    // obstacles that can be hit from here, list order

  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  if (b >= B) return;
  const double x0 = state[b * 5 + 0], y0 = state[b * 5 + 2];
  const double* exy = env_xy + b * env_stride * (long)n_env * v_env * 2;
  const int32_t* env = env_nv + b * env_stride * (long)n_env;
  int n_cand = 0;

  // ---- 1. ray casting (compute_lidar_readings) ---------------------------------------------------
  // candidate obstacles: those whose bounding circle comes within the range (conservative: an obstacle that is
  // skipped cannot hold a point closer than lidar_range), compacted once per robot in list order
  for (int j0 = 0; j0 < n_env; j0 += 64) {
    const int j = j0 + lane;
    bool keep = false;
    if (j < n_env) {
      const int nv = env[j];
      const double* ring = exy + (long)j * v_env * 2;
      double mx = 0.0, my = 0.0;
      for (int e = 0; e < nv; ++e) { mx += ring[2 * e]; my += ring[2 * e + 1]; }
      if (nv > 0) {
        mx /= nv; my /= nv;
        double rad = 0.0;
        for (int e = 0; e < nv; ++e) rad = fmax(rad, hypot(ring[2 * e] - mx, ring[2 * e + 1] - my));
        keep = hypot(mx - x0, my - y0) <= (lidar_range + rad) * (1.0 + 1e-9) + 1e-9;
      }
    }
    const unsigned long long ball = __ballot(keep);
    if (keep) { const int k = n_cand + __popcll(ball & ((1ull << lane) - 1ull)); if (k < RMAX) cand_[k] = j; }
    n_cand += __popcll(ball);
  }
  __syncthreads();
  for (int i = lane; i < RMAX; i += 64) {
    bool have = false;
    double hx = 0.0, hy = 0.0;
    if (i < R) {
#pragma clang fp contract(off)
      const double ex = x0 + lidar_range * ray_table[2 * i], ey = y0 + lidar_range * ray_table[2 * i + 1];
      const double rdx = ex - x0, rdy = ey - y0;               // b1 - a1
      double best_d = lidar_range;
      for (int jc = 0; jc < n_cand; ++jc) {
        const int j = cand_[jc];
        const int nv = env[j];
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

  // ---- 2. DBSCAN ------------------------------------------------------------------------------------
  const double eps2 = eps * eps;
  unsigned long long vmask[WORDS];                  // which readings exist, one ballot per 64 rays
#pragma unroll
  for (int w = 0; w < WORDS; ++w) vmask[w] = __ballot(comp_[w * 64 + lane] >= 0);
  for (int i = lane; i < RMAX; i += 64) {
    int cnt = 0;
    const bool vi = comp_[i] >= 0;
    const double xi = px_[i], yi = py_[i];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      // counted, unrolled sweep over all 64 slots of the word (loads pipeline; absent readings are masked after)
      unsigned long long bits = 0ull;
#pragma unroll 16
      for (int k = 0; k < 64; ++k) {
#pragma clang fp contract(off)
        const double dx = xi - px_[w * 64 + k], dy = yi - py_[w * 64 + k];
        bits |= (unsigned long long)(dx * dx + dy * dy <= eps2) << k;
      }
      bits = vi ? (bits & vmask[w]) : 0ull;
      cnt += __popcll(bits);
      nb_[i][w] = bits;
    }
    root_[i] = (vi && cnt >= min_samples) ? i : NO_ROOT;      // core points start as their own root
  }
  __syncthreads();
  for (int i = lane; i < RMAX; i += 64) if (comp_[i] >= 0) comp_[i] = root_[i];
  __syncthreads();
  // connected components of the core points (hook: min over core neighbours; then pointer jumping to the root;
  // repeat until a hook round changes nothing — a handful of rounds instead of one per hop of the longest chain)
  unsigned long long cmask[WORDS];
#pragma unroll
  for (int w = 0; w < WORDS; ++w) { const int cj = comp_[w * 64 + lane]; cmask[w] = __ballot(cj >= 0 && cj != NO_ROOT); }
  for (int round = 0; round < RMAX; ++round) {
    bool changed = false;
    for (int i = lane; i < RMAX; i += 64) {
      const int ci = comp_[i];
      bool mine = false;
      if (ci >= 0 && ci != NO_ROOT) {
        mine = true;
      }
      // lanes of one pass own rays i = lane + 64 k: their neighbours sit in the same few words, so a word is
      // swept by the whole wave with independent (pipelined, broadcast) LDS reads or skipped by the whole wave
      int m = mine ? ci : NO_ROOT;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        const unsigned long long bits = mine ? (nb_[i][w] & cmask[w]) : 0ull;
        if (__any(bits != 0ull)) {
#pragma unroll 4
          for (int k4 = 0; k4 < 16; ++k4) {
            const int4 c4 = *reinterpret_cast<const int4*>(&comp_[w * 64 + k4 * 4]);
            const unsigned nib = (unsigned)(bits >> (k4 * 4)) & 0xfu;
            m = (nib & 1u) ? min(m, c4.x) : m;
            m = (nib & 2u) ? min(m, c4.y) : m;
            m = (nib & 4u) ? min(m, c4.z) : m;
            m = (nib & 8u) ? min(m, c4.w) : m;
          }
        }
      }
      if (mine && m < ci) { comp_[i] = m; changed = true; }
    }
    __syncthreads();
    for (int jump = 0; jump < 4; ++jump) {
      for (int i = lane; i < RMAX; i += 64) {
        const int ci = comp_[i];
        if (ci >= 0 && ci != NO_ROOT) { const int cc = comp_[ci]; if (cc < ci) comp_[i] = cc; }
      }
      __syncthreads();
    }
    if (!__any(changed)) break;
  }
  // cluster root of every reading: own component for cores, smallest neighbouring core component for the rest
  for (int i = lane; i < RMAX; i += 64) {
    const int ci = comp_[i];
    int r = NO_ROOT;
    if (ci >= 0) {
      if (ci != NO_ROOT) r = ci;
      else {
        for (int w = 0; w < WORDS; ++w) {
          unsigned long long bits = nb_[i][w];
          while (bits) {
            const int k = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const int cj = comp_[w * 64 + k];
            if (cj != NO_ROOT && cj < r) r = cj;
          }
        }
      }
    }
    root_[i] = r;
  }
  __syncthreads();
  // roots in ascending order = cluster labels 0, 1, ...
  int n_clusters = 0;
  for (int w = 0; w < WORDS; ++w) {
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
    for (int i = lane; i < R; i += 64) {
      int lab = -2;                                           // -2 no reading, -1 noise
      if (comp_[i] >= 0) {
        lab = -1;
        const int r = root_[i];
        if (r != NO_ROOT) for (int k = 0; k < n_clusters && k < 64; ++k) if (roots_[k] == r) lab = k;
      }
      labels_out[b * R + i] = lab;
    }
  }

  // ---- 3. convex hull per cluster (create_convex_hull) ------------------------------------------------
  int n_out = 0, ovf = (n_clusters > 64) ? 1 : 0;
  double* oxy = obs_xy + b * (long)n_obs_max * v_max * 2;
  int32_t* onv = obs_nv + b * (long)n_obs_max;
  for (int k = lane; k < n_obs_max; k += 64) onv[k] = 0;
  for (int k = 0; k < n_clusters && k < 64; ++k) {
    const int r = roots_[k];
    // lexicographically smallest point of the cluster
    Cand st; st.idx = -1; st.x = 0.0; st.y = 0.0;
    for (int i = lane; i < RMAX; i += 64) {
      if (root_[i] == r) {
        const double x = px_[i], y = py_[i];
        if (st.idx < 0 || x < st.x || (x == st.x && (y < st.y || (y == st.y && i < st.idx)))) { st.x = x; st.y = y; st.idx = i; }
      }
    }
    for (int m = 1; m < 64; m <<= 1) {
      Cand o; o.x = shfl_d(st.x, m); o.y = shfl_d(st.y, m); o.idx = __shfl_xor(st.idx, m, 64);
      const bool take = o.idx >= 0 && (st.idx < 0 || o.x < st.x || (o.x == st.x && (o.y < st.y || (o.y == st.y && o.idx < st.idx))));
      if (take) st = o;
    }
    // Jarvis march; vertices are buffered in the output slot n_out and committed only for a proper polygon
    double cxp = st.x, cyp = st.y;
    int nvert = 0;
    const bool room = n_out < n_obs_max;
    for (int step = 0; step <= v_max; ++step) {
      if (room && nvert < v_max && lane == 0) { oxy[((long)n_out * v_max + nvert) * 2] = cxp; oxy[((long)n_out * v_max + nvert) * 2 + 1] = cyp; }
      ++nvert;
      Cand best; best.idx = -1; best.x = 0.0; best.y = 0.0;
      for (int i = lane; i < RMAX; i += 64) {
        if (root_[i] == r) {
          Cand cnd; cnd.x = px_[i]; cnd.y = py_[i]; cnd.idx = i;
          if (!(cnd.x == cxp && cnd.y == cyp) && better(cxp, cyp, best, cnd)) best = cnd;
        }
      }
      for (int m = 1; m < 64; m <<= 1) {
        Cand o; o.x = shfl_d(best.x, m); o.y = shfl_d(best.y, m); o.idx = __shfl_xor(best.idx, m, 64);
        if (better(cxp, cyp, best, o)) best = o;
      }
      if (best.idx < 0) break;                                   // single (repeated) point
      if (best.x == st.x && best.y == st.y) break;               // ring closed
      cxp = best.x; cyp = best.y;
    }
    // < 3 extreme points = fewer than 3 unique points or a collinear cluster: the reference drops it (:70-76)
    if (nvert >= 3) {
      if (!room || nvert > v_max) ovf = 1;
      else { if (lane == 0) onv[n_out] = nvert; ++n_out; }
    }
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
  if (B < 0 || resolution < 1 || resolution > RMAX || n_env < 0 || v_env < 1 || n_obs_max < 1 || v_max < 3) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state || !ray_table || !obs_xy || !obs_nv || !n_inferred || !overflow || (n_env > 0 && (!env_xy || !env_nv)))
    return LIPMPC_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return LIPMPC_E_HIP;
  hipLaunchKernelGGL(lidar_sense_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)hip_stream, (long)B, resolution, n_env,
                     v_env, (long)(env_shared ? 0 : 1), lidar_range, eps, min_samples, n_obs_max, v_max, state, env_xy, env_nv,
                     ray_table, noise, obs_xy, obs_nv, n_inferred, overflow, hits, labels);
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}
