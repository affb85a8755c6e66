// lipmpc_lidar.hip — unknown-environment front end (BASELINE config 5): per robot and MPC step, a 2-D LiDAR scan of the
// true map -> noisy readings -> DBSCAN clusters -> convex hull per cluster, i.e. the obstacle rings the step kernel
// consumes.  Restates
//   range_finder / compute_lidar_readings / retrieve_clusters / build_local_obstacles
//       HumanoidNavigation/RangeFinder/range_finder_wth_polygons_dbscan.py:26-63, 65-83, 100-126, 157-180
//   line_polygon_intersection (compute_intersection)      HumanoidNavigation/Utils/obstacles.py:95-139
//   the call site                                          HumanoidNavigation/MPC/HumanoidMPCVariants/HumanoidMPCUnknownEnvironment.py:30-68
// One wavefront (64 lanes) per robot; everything between the ray casting and the half-spaces stays in LDS / registers:
//   1. rays: lane l owns rays l, l+64, ...; the edges of the obstacles within range are staged once per robot in LDS
//      (edge vector and the robot's offset from the edge's first vertex: the operands of compute_intersection, no
//      global / scalar load left in the ray loops) and every ray walks them in list order, keeping the nearest hit
//      strictly inside the range (contraction off: the hit points are bit-identical to the reference's)
//   2. DBSCAN(eps, min_samples) by its order-free characterisation (oracle/lidar_oracle.py), on the readings
//      compacted in ray order: neighbour bit rows, core flags, connected components of the core points (forest of
//      "smallest core neighbour" pointers + pointer jumping, then merging trees through ballot masks of tree
//      membership — bit operations, no sweeps over neighbours' labels), clusters numbered by their smallest core
//      index, border points to the smallest neighbouring cluster
//   3. hull per cluster: Jarvis march from the lexicographically smallest point, farthest point on collinear ties
//      (= the CCW ring of extreme points Qhull / monotone chain return, same rotation as np.unique + monotone chain);
//      four clusters march at once, one per 16-lane DPP row, over compacted member lists
//   4. constraint assembly (HumanoidMPCUnknownEnvironment.py:54-62 -> ObstaclesUtils.py:60-109): closest point c and
//      unit normal eta of every hull at the robot's CoM, one hull edge per lane, from the hull still staged in LDS --
//      the (c, eta) rows are what the step solver consumes (lipmpc_plan_step_batch_c_eta); the rings themselves go to
//      HBM only when the caller asks for them
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
// order buffer (int32, scratch of ONE call): [B the order is valid for, -, order[B] (robot at launch position i), weight[B]]
constexpr int SCHED_VALID = 0, SCHED_ORDER = 2;
constexpr int NRUN = RMAX / 16;      // runs of 16 consecutive readings
constexpr int ECAP = 2 * RMAX / 4;  // edges staged per chunk of the ray phase (4 doubles each, in the hull stage's point arrays)
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

__global__ __launch_bounds__(64) void lidar_sense_kernel(
    long B, int R, int n_env, int v_env, long env_stride, double lidar_range, double eps, int min_samples,
    int n_obs_max, int v_max, const double* __restrict__ state, const double* __restrict__ env_xy,
    const int32_t* __restrict__ env_nv, const double* __restrict__ ray_table, const double* __restrict__ noise,
    double* __restrict__ obs_xy, int32_t* __restrict__ obs_nv, double* __restrict__ c_eta, int32_t* __restrict__ n_inferred,
    int32_t* __restrict__ overflow, double* __restrict__ hits_out, int32_t* __restrict__ labels_out,
    int32_t* __restrict__ sched, int dbg_stop) {
  __shared__ __attribute__((aligned(16))) double pxy_[2 * RMAX];
  double* const px_ = pxy_;
  double* const py_ = pxy_ + RMAX;
  __shared__ __attribute__((aligned(16))) int comp_[RMAX];                    // -1 = no reading; core: component root; else NO_ROOT
  __shared__ int root_[RMAX];                    // cluster root of every reading (NO_ROOT = noise)
  __shared__ __attribute__((aligned(16))) double cxy_[2 * RMAX];   // ray phase: staged edges; hull stage: points in member-list order
  double* const cx_ = cxy_;
  double* const cy_ = cxy_ + RMAX;
  double* const edge_ = cxy_;                    // [ECAP][4]: (b - a) and (robot - a) of every staged edge
  __shared__ int roots_[64];
  __shared__ int eoff_[65];                      // first staged edge of the chunk's candidates
  __shared__ double candc_[NCC][3];              // bounding circle (centre, radius) of the first NCC candidate obstacles
  __shared__ double bb16_[NRUN][4];              // bounding box (x0, x1, y0, y1) of each run of 16 points
  __shared__ unsigned nearm_[NRUN];              // runs within eps of run a, as bits
  __shared__ int cand_[RMAX];                    // obstacles that can be hit from here, list order

  const int lane = threadIdx.x;
  if ((long)blockIdx.x >= B) return;
  // Which robot this wave scans: the block index, or -- with an order buffer (include/lipmpc.h) -- the robot the order
  // kernel of THIS call put at this position: heaviest first, by an estimate of its reading count (lidar_weight_kernel).  A
  // scan's length varies 3x with the number of readings (all-pairs clustering), 4096 robots run in two rounds on the 2048 wave
  // slots, and a heavy robot started late sets the launch time: 241 us as the robots come, 159 us heaviest first by the true
  // counts (tools/lidar_order.py).  Any order gives the same results.
  long b = blockIdx.x;
  if (sched && sched[SCHED_VALID] == (int)B) {
    const long r = sched[SCHED_ORDER + blockIdx.x];
    if (r >= 0 && r < B) b = r;
  }
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
  // this lane's rays (i = lane + 64 p): direction b1 - a1, nearest hit so far.  A ray beyond the resolution has a zero
  // direction: every denominator is 0, it never hits.
  double rdx[WORDS], rdy[WORDS], inv_len2[WORDS], bd[WORDS], hx[WORDS], hy[WORDS];
#pragma unroll
  for (int p = 0; p < WORDS; ++p) {
#pragma clang fp contract(off)
    const int i = p * 64 + lane;
    const bool on = i < R;
    const double ex = x0 + lidar_range * (on ? ray_table[2 * i] : 0.0), ey = y0 + lidar_range * (on ? ray_table[2 * i + 1] : 0.0);
    rdx[p] = on ? ex - x0 : 0.0; rdy[p] = on ? ey - y0 : 0.0;
    inv_len2[p] = __builtin_amdgcn_rcp(rdx[p] * rdx[p] + rdy[p] * rdy[p]);
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
      double ax = nv > 0 ? ring[0] : 0.0, ay = nv > 0 ? ring[1] : 0.0;
      const double fx0 = ax, fy0 = ay;
      for (int e = 0; e < nv; ++e) {
        const bool last = e + 1 == nv;
        const double bx = last ? fx0 : ring[2 * (e + 1)], by = last ? fy0 : ring[2 * (e + 1) + 1];
        double* o = edge_ + 4 * (off + e);
        o[0] = bx - ax; o[1] = by - ay; o[2] = x0 - ax; o[3] = y0 - ay;
        ax = bx; ay = by;
      }
      if (lane == nfit - 1) eoff_[nfit] = incl;
    } else if (lane == 0) { eoff_[0] = 0; eoff_[1] = 0; }      // (only when nothing fitted)
    __syncthreads();
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
          const double tt = fmin(1.0, fmax(0.0, (wx * rdx[p] + wy * rdy[p]) * inv_len2[p]));
          const double ddx = wx - tt * rdx[p], ddy = wy - tt * rdy[p];
          if (!__any(ddx * ddx + ddy * ddy <= cr2)) continue;
        }
        for (int e = e0; e < e1; ++e) {
#pragma clang fp contract(off)
          const double gx = edge_[4 * e], gy = edge_[4 * e + 1], fx = edge_[4 * e + 2], fy = edge_[4 * e + 3];
          const double denom = gy * rdx[p] - gx * rdy[p];
          if (denom == 0.0) continue;
          const double nua = gx * fy - gy * fx;
          const double nub = rdx[p] * fy - rdy[p] * fx;
          // 0 <= ua <= 1 and 0 <= ub <= 1 for ua = nua / denom, ub = nub / denom decided WITHOUT dividing: a correctly
          // rounded quotient is >= 0 exactly when the signs agree (or the numerator is +-0) and <= 1 exactly when
          // |numerator| <= |denominator| (a quotient above 1 is at least 1 + 2^-53 (1 + tiny) and rounds to 1 + 2^-52;
          // the one unreachable exception: a negative quotient below 5e-324 in magnitude, which rounds to -0.0 >= 0).
          // Only a ray that really hits the edge pays for the division that places the hit.
          const double ad = fabs(denom), sa = (denom > 0.0) ? nua : -nua, sb = (denom > 0.0) ? nub : -nub;
          if (sa >= 0.0 && sa <= ad && sb >= 0.0 && sb <= ad) {
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
#pragma unroll
  for (int p = 0; p < WORDS; ++p) {
#pragma clang fp contract(off)
    const int i = p * 64 + lane;
    const bool have = bd[p] < lidar_range;           // bd starts at the range and only ever gets strictly smaller
    double qx = have ? hx[p] : 0.0, qy = have ? hy[p] : 0.0;
    if (have && noise) { qx = qx + noise[(b * R + i) * 2]; qy = qy + noise[(b * R + i) * 2 + 1]; }
    px_[i] = qx; py_[i] = qy;
    comp_[i] = have ? NO_ROOT : -1;
    if (hits_out && i < R) { hits_out[(b * R + i) * 2] = have ? qx : NAN; hits_out[(b * R + i) * 2 + 1] = have ? qy : NAN; }
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
  // that owns a point is the only one that reads its row), which keeps the LDS footprint at 20 KB = 8 waves per CU.
  // All-pairs is 147 k distance tests for 384 readings (it was the longest phase of the scan), so the sweep is pruned
  // and each test made cheap:
  //  * readings come in ray order, so a RUN of 16 consecutive points is a short piece of one obstacle's outline with a
  //    small bounding box.  Run a is tested against word w only if its box comes within eps of the box of one of w's
  //    four runs (a 24 x 24 bit matrix of run pairs, one lane per run, computed once); both box tests are conservative
  //    (eps with a margin far above any rounding), so no neighbour pair is ever dropped;
  //  * a visited (run, word) tile is computed COLUMN by column: every lane holds one point of word w in registers, the
  //    run's point kk comes as one LDS broadcast read, one compare gives the 64 bits "points of word w within eps of
  //    point kk" as a wave mask -- by symmetry row[k][w] of point kk -- and v_writelane drops it into lane kk:
  //    8 VALU instructions per 64 pair tests, no bit insertion.  dx^2 + dy^2 does not depend on which of the two
  //    points is subtracted from which, so the bits are those of the reference's distance test.
  unsigned long long row[WORDS][WORDS];
  double wxr[WORDS], wyr[WORDS];                         // this lane's point of every word
  double* const pint_ = cxy_;                            // [RMAX][2] interleaved copy of the points (the staged edges are dead)
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
    wxr[k] = px_[k * 64 + lane]; wyr[k] = py_[k * 64 + lane];
    pint_[2 * (k * 64 + lane)] = wxr[k]; pint_[2 * (k * 64 + lane) + 1] = wyr[k];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) row[k][w] = 0ull;
  }
#pragma unroll
  for (int w = 0; w < WORDS; ++w) {                       // boxes of the runs (empty run: an empty box)
    const bool vi = (vmask[w] >> lane) & 1ull;
    double x0 = vi ? wxr[w] : INFINITY, x1 = vi ? wxr[w] : -INFINITY, y0 = vi ? wyr[w] : INFINITY, y1 = vi ? wyr[w] : -INFINITY;
    x0 = fmin(x0, lipmpc_dev::row_xor<1>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<1>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<1>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<1>(y1));
    x0 = fmin(x0, lipmpc_dev::row_xor<2>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<2>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<2>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<2>(y1));
    x0 = fmin(x0, lipmpc_dev::row_xor<4>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<4>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<4>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<4>(y1));
    x0 = fmin(x0, lipmpc_dev::row_xor<8>(x0)); x1 = fmax(x1, lipmpc_dev::row_xor<8>(x1)); y0 = fmin(y0, lipmpc_dev::row_xor<8>(y0)); y1 = fmax(y1, lipmpc_dev::row_xor<8>(y1));
    if ((lane & 15) == 0) { double* o = bb16_[w * 4 + (lane >> 4)]; o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1; }
  }
  __syncthreads();
  const double epsx = eps * (1.0 + 1e-6) + 1e-9;         // eps with a margin for the box tests
  {   // run a = lane: which runs b come within eps of it (bit b)
    unsigned nm = 0u;
    const double* me = bb16_[lane < NRUN ? lane : 0];
    const double ax0 = me[0] - epsx, ax1 = me[1] + epsx, ay0 = me[2] - epsx, ay1 = me[3] + epsx;
#pragma unroll 4
    for (int rb = 0; rb < NRUN; ++rb) {
      const double* o = bb16_[rb];
      if (o[0] <= ax1 && o[1] >= ax0 && o[2] <= ay1 && o[3] >= ay0) nm |= 1u << rb;
    }
    if (lane < NRUN) nearm_[lane] = nm;
  }
  __syncthreads();
  // (lo, hi) with lane l (wave-uniform) replaced by the wave-uniform 64-bit mask.  v_writelane_b32 takes ONE scalar
  // operand besides m0 (constant bus), so the lane select goes through m0, saved and restored around the pair.
  auto wrlane64 = [](int& lo, int& hi, unsigned long long mask, int l) {
    int keep;
    asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
        : "+v"(lo), "+v"(hi), "=&s"(keep)
        : "s"((int)(unsigned)mask), "s"((int)(unsigned)(mask >> 32)), "s"(l));
  };
  // Only the blocks w >= k are computed: in a block with w > k every lane also notes, per column, whether ITS point (of
  // word w) is within eps of the column's point (of word k) -- bit kk of its own row[w][k], the transposed block, for
  // two more instructions per column instead of a second pass.
  for (int k = 0; k < NW; ++k) {                          // wave-uniform
    unsigned nmk[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) nmk[a] = __builtin_amdgcn_readfirstlane(nearm_[k * 4 + a]);
    const int leftk = n_pts - k * 64;
    const unsigned long long vmk = leftk >= 64 ? ~0ull : ((1ull << leftk) - 1ull);        // vmask[k] (leftk >= 1 for k < NW)
    const bool own_k = (vmk >> lane) & 1ull;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
      if (w >= NW || w < k) continue;
      const bool off = w != k;                            // wave-uniform: an off-diagonal block also fills its transpose
      int lo = 0, hi = 0;                                 // the row words being assembled: lane kk gets the mask of column kk
      unsigned tlo = 0u, thi = 0u;                        // this lane's point (word w) against the columns (word k)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (((nmk[a] >> (4 * w)) & 0xFu) == 0u) continue; // run a of word k has no point near word w
        const double* col = pint_ + 2 * (k * 64 + a * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma clang fp contract(off)
          const double dx = wxr[w] - col[2 * u], dy = wyr[w] - col[2 * u + 1];
          const bool near = dx * dx + dy * dy <= eps2;
          const unsigned long long mask = __ballot(near);
          wrlane64(lo, hi, mask, a * 16 + u);
          if (off) {
            constexpr unsigned one = 1u;
            if (a < 2) tlo |= near ? (one << ((a * 16 + u) & 31)) : 0u;
            else thi |= near ? (one << ((a * 16 + u) & 31)) : 0u;
          }
        }
      }
      const unsigned long long bits = ((((unsigned long long)(unsigned)hi) << 32) | (unsigned)lo) & vmask[w];
      const unsigned long long tbits = ((((unsigned long long)thi) << 32) | tlo) & vmk;
      const bool own_w = (vmask[w] >> lane) & 1ull;
#pragma unroll
      for (int k2 = 0; k2 < WORDS; ++k2) {
        if (k2 == k) {                                     // k is wave-uniform
          row[k2][w] = own_k ? bits : 0ull;
          if (off) row[w][k2] = own_w ? tbits : 0ull;
        }
      }
    }
  }
  // core points (>= min_samples neighbours, the point itself included) start as their own root
#pragma unroll
  for (int k = 0; k < WORDS; ++k) {
    if (k >= NW) continue;
    const int i = k * 64 + lane;
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) cnt += __popcll(row[k][w]);
    root_[i] = (comp_[i] >= 0 && cnt >= min_samples) ? i : NO_ROOT;
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
  double* oxy = obs_xy ? obs_xy + b * (long)n_obs_max * v_max * 2 : nullptr;
  int32_t* onv = obs_nv ? obs_nv + b * (long)n_obs_max : nullptr;
  double* oce = c_eta ? c_eta + b * (long)n_obs_max * 4 : nullptr;
  if (onv) for (int k = lane; k < n_obs_max; k += 64) onv[k] = 0;
  if (oce) for (int k = lane; k < n_obs_max * 4; k += 64) oce[k] = 0.0;       // eta = (0, 0): empty slot
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
          const double* ring = stage_ + qq * VSTAGE * 2;
          if (oxy) for (int v = lane; v < nv * 2; v += 64) oxy[(long)n_out * v_max * 2 + v] = ring[v];
          if (onv && lane == 0) onv[n_out] = nv;
          if (oce) {
            // ---- 4. constraint assembly: closest point c and unit normal eta of this hull at the CoM, one edge per lane
            // (nv <= VSTAGE = 64); per edge the arithmetic of closest_point_normal, the nearest edge by a wave minimum with
            // the first edge winning ties (the sequential scan keeps the first strict minimum), inside = parity of the
            // crossing hits.  A zero-length edge or x == c is degenerate geometry: eta = NaN, the step reports DEGENERATE.
#pragma clang fp contract(off)
            const bool eon = lane < nv;
            const int ia = eon ? lane : 0, ib = (ia + 1 == nv) ? 0 : ia + 1, ip = (ia == 0) ? nv - 1 : ia - 1;
            const EdgeCp ec = edge_closest(ring[2 * ip], ring[2 * ip + 1], ring[2 * ia], ring[2 * ia + 1], ring[2 * ib],
                                           ring[2 * ib + 1], x0, y0);
            double dmin = eon ? ec.d : INFINITY;
            for (int m = 1; m < 64; m <<= 1) dmin = fmin(dmin, __shfl_xor(dmin, m, 64));
            const int sel = __ffsll((long long)__ballot(eon && ec.d == dmin)) - 1;      // >= 0: lane 0 is always an edge
            const double ccx = __shfl(ec.qx, sel, 64), ccy = __shfl(ec.qy, sel, 64);
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
}

// Weight of every robot for the launch order of its scan: an ESTIMATE of its reading count from the bounding circles of the
// obstacles in range -- the rays a circle of radius r at distance d subtends, R / (2 pi) * 2 asin(r / d) (the asin by its
// argument: the estimate only ranks), scaled by the share of the circle inside the range, summed and capped at R.  Correlation
// 0.76 with the true counts on a CROWDED-style map, 27 of the 30 heaviest robots in its top 60: enough for
// "heaviest first", and it costs one obstacle per lane instead of the scan itself.  One wave per robot.
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
    for (int e = 0; e < nv; ++e) { mx += ring[2 * e]; my += ring[2 * e + 1]; }
    mx /= nv; my /= nv;
    for (int e = 0; e < nv; ++e) { const double dx = ring[2 * e] - mx, dy = ring[2 * e + 1] - my; rad2 = fmax(rad2, dx * dx + dy * dy); }
    const double rad = sqrt(rad2), d = sqrt((mx - x0) * (mx - x0) + (my - y0) * (my - y0));
    if (d > lidar_range + rad) continue;
    const double inside = (d + rad <= lidar_range) ? 1.0 : fmin(1.0, fmax(0.0, (lidar_range + rad - d) / (2.0 * rad + 1e-300)));
    w += (d <= rad) ? (double)R : (double)R * (1.0 / M_PI) * fmin(1.0, rad / d) * inside;
  }
  for (int m = 1; m < 64; m <<= 1) w += __shfl_xor(w, m, 64);
  if (lane == 0) sched[SCHED_ORDER + B + b] = (int32_t)fmin(w, (double)R);
}

// The launch order of the scans from the weights: robots by descending weight (counting sort, one workgroup; which of two
// equally heavy robots comes first is immaterial).
__global__ __launch_bounds__(1024) void lidar_order_kernel(long B, int32_t* __restrict__ sched) {
  __shared__ int cursor_[RMAX + 1];
  const int32_t* w = sched + SCHED_ORDER + B;
  int32_t* order = sched + SCHED_ORDER;
  for (int k = threadIdx.x; k <= RMAX; k += blockDim.x) cursor_[k] = 0;
  __syncthreads();
  for (long i = threadIdx.x; i < B; i += blockDim.x) atomicAdd(&cursor_[RMAX - min(max(w[i], 0), RMAX)], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k <= RMAX; ++k) { const int c = cursor_[k]; cursor_[k] = run; run += c; }
  }
  __syncthreads();
  for (long i = threadIdx.x; i < B; i += blockDim.x) order[atomicAdd(&cursor_[RMAX - min(max(w[i], 0), RMAX)], 1)] = (int32_t)i;
  if (threadIdx.x == 0) sched[SCHED_VALID] = (int32_t)B;
}

}  // namespace

static int lidar_launch(int device, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env, int32_t env_shared,
                        double lidar_range, double eps, int32_t min_samples, int32_t n_obs_max, int32_t v_max,
                        const double* state, const double* env_xy, const int32_t* env_nv, const double* ray_table,
                        const double* noise, double* obs_xy, int32_t* obs_nv, double* c_eta, int32_t* n_inferred,
                        int32_t* overflow, double* hits, int32_t* labels, int32_t* schedule, void* hip_stream) {
  if (B < 0 || resolution < 1 || resolution > RMAX || n_env < 0 || v_env < 1 || n_obs_max < 1 || v_max < 3 || v_max > VSTAGE) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state || !ray_table || !n_inferred || !overflow || (n_env > 0 && (!env_xy || !env_nv)) || (!obs_xy != !obs_nv) ||
      (!obs_xy && !c_eta))
    return LIPMPC_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return LIPMPC_E_HIP;
#ifdef LIPMPC_LIDAR_PHASES
  // profiling build only (make CXXFLAGS+=-DLIPMPC_LIDAR_PHASES, tools/lidar_phases.py): LIPMPC_LIDAR_STOP=1..5 ends the
  // kernel after that phase; outputs are then undefined.  The shipped library has no such knob.
  static const int dbg_stop = getenv("LIPMPC_LIDAR_STOP") ? atoi(getenv("LIPMPC_LIDAR_STOP")) : 0;
#else
  const int dbg_stop = 0;
#endif
  if (schedule && n_env > 0) {
    // rank the robots first: estimate of the reading counts -> order, heaviest first; the scans then start in that order
    hipLaunchKernelGGL(lidar_weight_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)hip_stream, (long)B, resolution, n_env, v_env,
                       (long)(env_shared ? 0 : 1), lidar_range, state, env_xy, env_nv, schedule);
    hipLaunchKernelGGL(lidar_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, (long)B, schedule);
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
