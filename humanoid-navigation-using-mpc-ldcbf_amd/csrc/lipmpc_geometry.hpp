// lipmpc_geometry.hpp -- closest point on a convex ring, unit normal, inside flip (ObstaclesUtils.py:50-109), bit-compatible with the oracles
// Part of the MI355X-native batched LIP-MPC / LDCBF step solver (csrc/lipmpc_kernel.hpp includes the parts in order).
#pragma once
#include "lipmpc_types.hpp"

namespace lipmpc_dev {

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


}  // namespace lipmpc_dev
