/* lipmpc_oracle.c — CPU restatement (plain C, float64, dense) of the reference's per-timestep
 * LIP-MPC / LDCBF step.  TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY: nothing in the shipped
 * package links or loads this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * It is a line-for-line C port of oracle/lipmpc_oracle.py (same start point, step rule, stop test,
 * active-set finish), deliberately DENSE: the constraint matrix G (m x n) is materialised and
 * K = 2I + G^T D G is formed with m n (n+1)/2 multiply-adds, i.e. exactly the algorithmic work
 * SURVEY.md §8(d) prices (F_iter) on the rows that are in the solve -- all of them with
 * LIPMPC_FLAG_NO_PRESOLVE, the rows the presolve keeps otherwise (lipmpc_oracle.py: presolve_ldcbf; the
 * ballast is one weighted zero row here).  The HIP kernel never forms G; agreement between the two is
 * therefore a check of the kernel's structured operators, not a tautology.
 *
 * Reference lines followed (HumanoidNavigation/...):
 *   theta/omega           MPC/HumanoidMpc.py:137-160
 *   closest point, eta    Utils/ObstaclesUtils.py:50-109
 *   rows                  MPC/HumanoidMpc.py:183-249 (reach, manoeuvrability, walking velocity),
 *                         :252-294 + MPC/HumanoidMPCVariants/HumanoidMPCCustomLCBF.py:30-31 (LDCBF)
 *   cost                  MPC/HumanoidMpc.py:321-333
 *   dynamics              MPC/HumanoidMpc.py:34-48
 * The solve itself (CasADi Opti + IPOPT, HumanoidMpc.py:97-100,417) is third-party and absent
 * offline: PARITY UNPINNED for it; see the header of lipmpc_oracle.py.
 *
 * Build: make -C oracle   (gcc -O3 -march=native -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/lipmpc.h"

#define NMAXV 32   /* 2 * 16 */
#define MMAX 944   /* 9*16 + 16*50 */

static const double IPM_S_FLOOR = 0.1, IPM_Z0 = 30.0, IPM_STEP_FRAC = 0.995, IPM_Z_DIVERGE = 1e13;
static const double IPM_STALL_TOL = 1e-6;
static const int IPM_SLOW_FROM = 8;
static const double IPM_SLOW_RATIO = 0.9, IPM_SLOW_SIGMA = 0.5;   /* no-progress safeguard, see lipmpc_oracle.py */
static const double FIN_RHO = 1e10, FIN_EPS = 1e-9, FIN_INNER_TOL = 1e-11, FIN_IDENT = 1e5, FIN_STALL = 0.5;
static const double FIN_GD_MIN = 1e-14, FIN_DUAL_REL = 1e-14, FIN_RHO_POLISH = 1e12, FIN_POLISH_TOL = 1e-10;
static const double SCREEN_MARGIN = 1e-3;     /* presolve: see lipmpc_oracle.py */
enum { FIN_ROUNDS = 8, FIN_ROUNDS_LONG = 16, FIN_INNER = 6 };

/* ---- geometry (ObstaclesUtils.py:50-109) ------------------------------------------------ */
static void closest_point_normal(const double* ring, int nv, double px, double py, double* cx, double* cy,
                                 double* ex, double* ey, int* degenerate) {
  double best = INFINITY;
  *cx = NAN; *cy = NAN; *degenerate = 0;
  int inside = 0;
  double x0v = ring[2 * (nv - 1)], y0v = ring[2 * (nv - 1) + 1];
  int f0 = y0v >= py;
  for (int i = 0; i < nv; ++i) {
    double ax = ring[2 * i], ay = ring[2 * i + 1];
    int i1 = (i + 1 == nv) ? 0 : i + 1;
    double bx = ring[2 * i1], by = ring[2 * i1 + 1];
    double dx = bx - ax, dy = by - ay;
    double nrm = sqrt(dx * dx + dy * dy);
    double den = nrm * nrm;
    if (den == 0.0) {
      *degenerate = 1;
    } else {
      double t = ((px - ax) * dx + (py - ay) * dy) / den;
      t = fmax(0.0, fmin(1.0, t));
      double qx = ax + t * dx, qy = ay + t * dy;
      double ux = qx - px, uy = qy - py;
      double d = sqrt(ux * ux + uy * uy);
      if (d < best) { best = d; *cx = qx; *cy = qy; }
    }
    int f1 = ay >= py;
    if (f0 != f1) {
      int hit = ((ay - py) * (x0v - ax) >= (ax - px) * (y0v - ay)) == f1;
      if (hit) inside = !inside;
    }
    x0v = ax; y0v = ay; f0 = f1;
  }
  double nx = px - *cx, ny = py - *cy;
  double nn = sqrt(nx * nx + ny * ny);
  if (!(nn > 0.0)) { *degenerate = 1; *ex = 0.0; *ey = 0.0; return; }
  nx /= nn; ny /= nn;
  if (inside) { nx = -nx; ny = -ny; }
  *ex = nx; *ey = ny;
}

/* ---- dense linear algebra ---------------------------------------------------------------- */
/* Square-root-free factorisation K = L D L^T, right-looking, in place (lower triangle: L below the diagonal, D on it),
 * and the two substitutions.  Same pivots as a Cholesky factorisation in exact arithmetic; in floating point it is the
 * form the kernel uses, and it keeps going on badly conditioned K (cond ~ 1e16 in the last iterations of the N = 16 /
 * 50-obstacle class) where the square-root form reports a non-positive pivot a step earlier -- with the square-root form
 * here 5 of 4096 such problems came out INFEASIBLE that the kernel solves and certifies.  Returns 0 on a pivot <= 0. */
static int cholesky(double* K, int n) {
  for (int j = 0; j < n; ++j) {
    double pj = K[j * n + j];
    if (!(pj > 0.0)) return 0;
    double ip = 1.0 / pj;
    for (int i = j + 1; i < n; ++i) {
      double f = K[i * n + j] * ip;                 /* L_ij */
      for (int c = j + 1; c <= i; ++c) K[i * n + c] -= f * K[c * n + j];
    }
    for (int i = j + 1; i < n; ++i) K[i * n + j] *= ip;
  }
  return 1;
}
static void chol_solve(const double* L, int n, double* b) {
  for (int j = 0; j < n; ++j) {                      /* L w = b (unit lower) */
    double s = b[j];
    for (int c = 0; c < j; ++c) s -= L[j * n + c] * b[c];
    b[j] = s;
  }
  for (int j = 0; j < n; ++j) b[j] /= L[j * n + j]; /* D */
  for (int j = n - 1; j >= 0; --j) {                 /* L^T x = w */
    double s = b[j];
    for (int i = j + 1; i < n; ++i) s -= L[i * n + j] * b[i];
    b[j] = s;
  }
}
static void form_K(const double* G, const double* d, int m, int n, double* K) {
  memset(K, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) K[i * n + i] = 2.0;
  for (int r = 0; r < m; ++r) {
    double dr = d[r];
    if (dr == 0.0) continue;
    const double* g = G + (size_t)r * n;
    for (int i = 0; i < n; ++i) {
      double gi = dr * g[i];
      if (gi == 0.0) continue;
      for (int j = 0; j <= i; ++j) K[i * n + j] += gi * g[j];
    }
  }
}
static void mat_vec(const double* G, const double* x, int m, int n, double* y) { /* y = G x */
  for (int r = 0; r < m; ++r) {
    double s = 0.0;
    const double* g = G + (size_t)r * n;
    for (int i = 0; i < n; ++i) s += g[i] * x[i];
    y[r] = s;
  }
}
static void matT_vec(const double* G, const double* w, int m, int n, double* y) { /* y = G^T w */
  for (int i = 0; i < n; ++i) y[i] = 0.0;
  for (int r = 0; r < m; ++r) {
    double wr = w[r];
    if (wr == 0.0) continue;
    const double* g = G + (size_t)r * n;
    for (int i = 0; i < n; ++i) y[i] += g[i] * wr;
  }
}
static double max_step(const double* v, const double* dv, int m) {
  double a = INFINITY;
  for (int i = 0; i < m; ++i)
    if (dv[i] < 0.0) { double t = -v[i] / dv[i]; if (t < a) a = t; }
  return a;
}

typedef struct {
  double G[MMAX * NMAXV], h[MMAX];
  double s[MMAX], z[MMAX], rp[MMAX], d[MMAX], w[MMAX], rc[MMAX], ds[MMAX], dz[MMAX], dsa[MMAX], dza[MMAX], t[MMAX];
  double y[MMAX], slack[MMAX];
  int act[MMAX], canon[MMAX];
  double K[NMAXV * NMAXV];
} work_t;

/* One problem.  Layouts as in include/lipmpc.h. */
static void plan_one(const lipmpc_params* P0, const double* bnd, work_t* W, const double* st, const double* goal, int foot0, double delta,
                     const double* obs_xy, const int32_t* obs_nv, double* U, double* X, double* theta, double* omega,
                     double* obj, int32_t* status_out, int32_t* iters_out, uint64_t* active, uint64_t* working, double* c_eta,
                     double* diag, const double* c_eta_in) {
  lipmpc_params Pl = *P0;   /* per-problem (V_MAX_x, V_MAX_y, ALPHA, OMEGA_MAX) overrides, bounds_tuning.py:17-26 */
  if (bnd) { Pl.v_max_xy[0] = bnd[0]; Pl.v_max_xy[1] = bnd[1]; Pl.alpha = bnd[2]; Pl.omega_max = bnd[3]; }
  const lipmpc_params* P = &Pl;
  const int N = P->N, n = 2 * N, n_obs = P->n_obs_max;
  const double beta = sqrt(P->g / P->h_com), ch = cosh(beta * P->dt), sh = sinh(beta * P->dt);
  const double kap = beta * sh / (ch - 1.0);
  const int m_tot = 9 * N + (N + 1) * n_obs, words = (m_tot + 63) / 64;
  const double p0[2] = {st[0], st[2]}, v0[2] = {st[1], st[3]};
  for (int i = 0; i < words; ++i) { active[i] = 0; if (working) working[i] = 0; }
  for (int i = 0; i < N * 2; ++i) U[i] = NAN;
  for (int i = 0; i < (N + 1) * 4; ++i) X[i] = NAN;
  *obj = NAN; *iters_out = 0;
  if (diag) { memset(diag, 0, sizeof(double) * LIPMPC_DIAG_WORDS); diag[2] = INFINITY; }
  /* theta / omega (HumanoidMpc.py:137-160) */
  const double psi = atan2(goal[1] - p0[1], goal[0] - p0[0]);
  theta[0] = st[4];
  for (int k = 0; k < N; ++k) {
    double w = fmin(fmax(psi - theta[k], -P->omega_max), P->omega_max);
    omega[k] = w;
    theta[k + 1] = theta[k] + w * P->sampling_time;
  }
  double sv[17];
  for (int i = 0; i <= N; ++i) sv[i] = (i % 2 == 0) ? (double)foot0 : -(double)foot0;
  /* c, eta per obstacle (HumanoidMpc.py:296-319) */
  double ex[50], ey[50], bb[50], h0v[50];
  int present[50];
  int flag = 0;
  for (int j = 0; j < n_obs; ++j) {
    double cx = 0, cy = 0;
    ex[j] = ey[j] = bb[j] = 0.0;
    int dg = 0;
    if (c_eta_in) {            /* given half-spaces (lipmpc_plan_step_batch_c_eta): eta = (0,0) = empty slot */
      cx = c_eta_in[4 * j]; cy = c_eta_in[4 * j + 1]; ex[j] = c_eta_in[4 * j + 2]; ey[j] = c_eta_in[4 * j + 3];
      present[j] = (ex[j] != 0.0) || (ey[j] != 0.0);
    } else {
      int nv = obs_nv[j];
      present[j] = nv > 0;
      if (nv > 0) closest_point_normal(obs_xy + (size_t)j * P->v_max * 2, nv, p0[0], p0[1], &cx, &cy, &ex[j], &ey[j], &dg);
    }
    if (present[j]) {
      double ec = ex[j] * cx + ey[j] * cy;
      bb[j] = ec + delta;
      double h0 = (ex[j] * p0[0] + ey[j] * p0[1]) - ec - delta;
      h0v[j] = h0;
      if (dg) flag |= 2; else if (h0 < -P->k0_tol) flag |= 1;
    }
    if (c_eta) { c_eta[4 * j] = cx; c_eta[4 * j + 1] = cy; c_eta[4 * j + 2] = ex[j]; c_eta[4 * j + 3] = ey[j]; }
  }
  if (flag & 2) { *status_out = LIPMPC_STATUS_DEGENERATE; return; }
  if (flag & 1) { *status_out = LIPMPC_STATUS_INFEASIBLE; return; }

  /* affine maps p_k = Pm[k] q + pc[k], v_k = Vm[k] q + vc[k]  (x+ = A x + B u eliminated in u) */
  static __thread double Vm[17][2][NMAXV];
  double vc[17][2];
  memset(Vm, 0, sizeof(Vm));
  vc[0][0] = v0[0]; vc[0][1] = v0[1];
  for (int k = 1; k <= N; ++k)
    for (int a = 0; a < 2; ++a) {
      for (int i = 0; i < n; ++i) Vm[k][a][i] = -Vm[k - 1][a][i];
      Vm[k][a][2 * (k - 1) + a] += kap;
      if (k >= 2) Vm[k][a][2 * (k - 2) + a] -= kap;
      vc[k][a] = -vc[k - 1][a] - ((k == 1) ? kap * p0[a] : 0.0);   /* p_0 is the only constant position */
    }
  /* rows in canonical order, k=0 LDCBF rows skipped (constants) */
  double* G = W->G; double* h = W->h;
  int m = 0;
#define NEWROW(ci) do { memset(G + (size_t)m * n, 0, sizeof(double) * n); W->canon[m] = (ci); } while (0)
  for (int k = 0; k < N; ++k) {           /* reach: upper x,y then lower x,y */
    double c_ = cos(theta[k]), s_ = sin(theta[k]);
    double R[2][2] = {{c_, s_}, {-s_, c_}};
    for (int lo = 0; lo < 2; ++lo)
      for (int a = 0; a < 2; ++a) {
        NEWROW(4 * k + 2 * lo + a);
        double sg = lo ? -1.0 : 1.0;
        double dc = (a == 1) ? sv[k] * P->ell : 0.0;
        for (int cc = 0; cc < 2; ++cc) {
          G[(size_t)m * n + 2 * k + cc] += sg * R[a][cc];
          if (k >= 1) G[(size_t)m * n + 2 * (k - 1) + cc] -= sg * R[a][cc];
          else dc -= R[a][cc] * p0[cc];
        }
        h[m] = lo ? -(P->l_min[a] - dc) : (P->l_max[a] - dc);
        ++m;
      }
  }
  for (int k = 0; k < N; ++k) {           /* manoeuvrability: state k+1, theta_{k+1}, omega_k */
    double r[2] = {cos(theta[k + 1]), sin(theta[k + 1])};
    NEWROW(4 * N + k);
    double cst = 0.0;
    for (int cc = 0; cc < 2; ++cc) {
      for (int i = 0; i < n; ++i) G[(size_t)m * n + i] += r[cc] * Vm[k + 1][cc][i];
      cst += r[cc] * vc[k + 1][cc];
    }
    h[m] = P->v_max_xy[0] - (P->alpha / M_PI) * fabs(omega[k]) - cst;
    ++m;
  }
  for (int k = 1; k <= N; ++k) {          /* walking velocity: upper long,lat then lower long,lat */
    double c_ = cos(theta[k]), s_ = sin(theta[k]);
    double Wv[2][2] = {{c_, s_}, {-s_, c_ * sv[k]}};
    for (int lo = 0; lo < 2; ++lo)
      for (int a = 0; a < 2; ++a) {
        NEWROW(5 * N + 4 * (k - 1) + 2 * lo + a);
        double sg = lo ? -1.0 : 1.0, cst = 0.0;
        for (int cc = 0; cc < 2; ++cc) {
          for (int i = 0; i < n; ++i) G[(size_t)m * n + i] += sg * Wv[a][cc] * Vm[k][cc][i];
          cst += Wv[a][cc] * vc[k][cc];
        }
        h[m] = lo ? -(P->v_min[a] - cst) : (P->v_max_xy[a] - cst);
        ++m;
      }
  }
  /* presolve (lipmpc_oracle.py: presolve_ldcbf): the LDCBF rows the leg-reach rows make redundant leave the problem; n_d
   * copies of one ballast row 0.q <= s_bar keep their averaging effect on the interior-point iteration */
  const int presolve = !(P->flags & (LIPMPC_FLAG_INTERIOR | LIPMPC_FLAG_WARM_START | LIPMPC_FLAG_NO_PRESOLVE));
  double es[50];
  int n_ballast = 0;
  double s_ballast = 0.0;
  if (presolve) {
    const double dx = fmax(fabs(P->l_max[0]), fabs(P->l_min[0])), dy = fmax(fabs(P->l_max[1]), fabs(P->l_min[1])) + fabs(P->ell);
    const double step = sqrt(dx * dx + dy * dy);
    for (int j = 0; j < n_obs; ++j) {
      if (!present[j]) continue;
      es[j] = sqrt(ex[j] * ex[j] + ey[j] * ey[j]) * step;
      int nd_j = 0;
      for (int k = 1; k <= N; ++k) if (h0v[j] > es[j] * k + SCREEN_MARGIN) ++nd_j;
      n_ballast += nd_j;
      s_ballast += nd_j * h0v[j];
    }
    if (n_ballast) s_ballast /= n_ballast;
  }
  for (int k = 1; k <= N; ++k)            /* LDCBF k = 1..N: -eta.p_k <= -(delta + eta.c) */
    for (int j = 0; j < n_obs; ++j) {
      if (!present[j]) continue;
      if (presolve && h0v[j] > es[j] * k + SCREEN_MARGIN) continue;
      NEWROW(9 * N + k * n_obs + j);
      G[(size_t)m * n + 2 * (k - 1)] = -ex[j];
      G[(size_t)m * n + 2 * (k - 1) + 1] = -ey[j];
      h[m] = -bb[j];
      ++m;
    }
  /* ballast: ONE zero row 0.q <= s_bar standing for n_d identical copies (they evolve identically: weight n_d in the two
   * sums a row's complementarity product enters, mu and mu_aff); interior-point phase only */
  const int m_real = m;
  if (n_ballast > 0) { NEWROW(-1); h[m] = s_ballast; ++m; }
  const double m_count = (double)(m_real + n_ballast);
  const double w_ball = (double)n_ballast;
#undef NEWROW
  double g[NMAXV], q[NMAXV], rd[NMAXV], dq[NMAXV], tmp[NMAXV];
  for (int k = 0; k < N; ++k) { g[2 * k] = goal[0]; g[2 * k + 1] = goal[1]; q[2 * k] = p0[0]; q[2 * k + 1] = p0[1]; }

  /* ---- Mehrotra predictor-corrector on the normal equations --------------------------------- */
  double *s = W->s, *z = W->z, *rp = W->rp, *d = W->d, *w = W->w, *rc = W->rc, *ds = W->ds, *dz = W->dz;
  double *dsa = W->dsa, *dza = W->dza, *t = W->t, *K = W->K;
  mat_vec(G, q, m, n, t);
  for (int i = 0; i < m; ++i) { s[i] = fmax(h[i] - t[i], IPM_S_FLOOR); z[i] = IPM_Z0; }
  int status = LIPMPC_STATUS_MAX_ITER, it = 0;
  double mu = 0.0, mu_prev = INFINITY;
  const double tol = (P->flags & LIPMPC_FLAG_INTERIOR) ? P->tol_interior : P->tol;
  for (it = 0; it <= P->max_iter; ++it) {
    mat_vec(G, q, m, n, t);
    double rpmax = 0.0, zmax = 0.0, qmax = 0.0;
    mu = 0.0;
    for (int i = 0; i < m; ++i) {
      rp[i] = t[i] + s[i] - h[i];
      rpmax = fmax(rpmax, fabs(rp[i]));
      mu += ((i >= m_real) ? w_ball : 1.0) * (s[i] * z[i]);
      zmax = fmax(zmax, z[i]);
    }
    mu /= m_count;
    for (int i = 0; i < n; ++i) qmax = fmax(qmax, fabs(q[i]));
    if (rpmax <= tol && mu <= tol) { status = LIPMPC_STATUS_SOLVED; break; }
    if (it == P->max_iter) break;
    if (!(zmax < IPM_Z_DIVERGE) || !(qmax < 1e300)) { status = LIPMPC_STATUS_INFEASIBLE; break; }
    for (int i = 0; i < m; ++i) d[i] = z[i] / s[i];
    form_K(G, d, m, n, K);
    if (!cholesky(K, n)) {
      status = (rpmax <= IPM_STALL_TOL && mu <= IPM_STALL_TOL) ? LIPMPC_STATUS_SOLVED : LIPMPC_STATUS_INFEASIBLE;
      break;
    }
    matT_vec(G, z, m, n, rd);
    for (int i = 0; i < n; ++i) rd[i] += 2.0 * (q[i] - g[i]);
    /* predictor: rc = s z */
    for (int i = 0; i < m; ++i) w[i] = (z[i] * rp[i] - s[i] * z[i]) / s[i];
    matT_vec(G, w, m, n, tmp);
    for (int i = 0; i < n; ++i) dq[i] = -rd[i] - tmp[i];
    chol_solve(K, n, dq);
    mat_vec(G, dq, m, n, t);
    for (int i = 0; i < m; ++i) { dsa[i] = -rp[i] - t[i]; dza[i] = -(s[i] * z[i] + z[i] * dsa[i]) / s[i]; }
    double a_aff = fmin(1.0, fmin(max_step(s, dsa, m), max_step(z, dza, m)));
    double mu_aff = 0.0;
    for (int i = 0; i < m; ++i) mu_aff += ((i >= m_real) ? w_ball : 1.0) * ((s[i] + a_aff * dsa[i]) * (z[i] + a_aff * dza[i]));
    mu_aff /= m_count;
    double ratio = mu_aff / mu, sigma = ratio * ratio * ratio;
    if (it >= IPM_SLOW_FROM) {
      double ramp = fmin(1.0, fmax(0.0, (mu / mu_prev - IPM_SLOW_RATIO) * (1.0 / (1.0 - IPM_SLOW_RATIO))));
      sigma = fmax(sigma, IPM_SLOW_SIGMA * ramp);
    }
    mu_prev = mu;
    double sigma_mu = sigma * mu;
    for (int i = 0; i < m; ++i) { rc[i] = s[i] * z[i] + dsa[i] * dza[i] - sigma_mu; w[i] = (z[i] * rp[i] - rc[i]) / s[i]; }
    matT_vec(G, w, m, n, tmp);
    for (int i = 0; i < n; ++i) dq[i] = -rd[i] - tmp[i];
    chol_solve(K, n, dq);
    mat_vec(G, dq, m, n, t);
    for (int i = 0; i < m; ++i) { ds[i] = -rp[i] - t[i]; dz[i] = -(rc[i] + z[i] * ds[i]) / s[i]; }
    double a = fmin(1.0, IPM_STEP_FRAC * fmin(max_step(s, ds, m), max_step(z, dz, m)));
    for (int i = 0; i < n; ++i) q[i] += a * dq[i];
    for (int i = 0; i < m; ++i) { s[i] += a * ds[i]; z[i] += a * dz[i]; }
  }
  *iters_out = it;
  *status_out = status;
  if (status != LIPMPC_STATUS_SOLVED) return;
  m = m_real;                               /* the finish and the outputs see the real rows only */
  double margin = INFINITY;
  for (int i = 0; i < m; ++i) margin = fmin(margin, fabs(log(z[i] / (FIN_IDENT * s[i]))));
  if (diag) { diag[2] = margin; diag[3] = 0.0; }

  /* ---- certified active-set finish ------------------------------------------------------------- */
  int* act = W->act;
  double* y = W->y; double* slack = W->slack;
  for (int i = 0; i < m; ++i) act[i] = z[i] > FIN_IDENT * s[i];   /* under-estimate: see lipmpc_oracle.py */
  if (!(P->flags & LIPMPC_FLAG_INTERIOR)) {
    /* primal active-set rounds from the interior-point iterate x (see finish_active_set in lipmpc_oracle.py):
     * equality solve on A -> ratio test along d = x_A - x (a blocked step adds the blocking row) -> else drop the negative
     * multipliers -> else certified */
    double qf[NMAXV], xf[NMAXV], dd[NMAXV];
    memcpy(xf, q, sizeof(double) * n);
    for (int i = 0; i < m; ++i) y[i] = act[i] ? z[i] : 0.0;
    int certified = 0, rounds = 0;
    double eres = INFINITY, rho = FIN_RHO;
    const int fin_rounds = P->finish_rounds > 0 ? P->finish_rounds : (P->N <= 8 ? FIN_ROUNDS : FIN_ROUNDS_LONG);
    for (int rnd = 1; rnd <= fin_rounds; ++rnd) {
      rounds = rnd;
      for (int i = 0; i < m; ++i) d[i] = act[i] ? rho : 0.0;
      form_K(G, d, m, n, K);
      int fok = cholesky(K, n);
      memcpy(qf, xf, sizeof(double) * n);
      eres = INFINITY;
      double rdmax = 0.0, rmax = 0.0;
      for (int in = 0; in <= FIN_INNER; ++in) {
        mat_vec(G, qf, m, n, t);
        rmax = 0.0;
        for (int i = 0; i < m; ++i) { w[i] = act[i] ? (t[i] - h[i]) : 0.0; rmax = fmax(rmax, fabs(w[i])); }
        matT_vec(G, y, m, n, rd);
        rdmax = 0.0;
        for (int i = 0; i < n; ++i) { rd[i] += 2.0 * (qf[i] - g[i]); rdmax = fmax(rdmax, fabs(rd[i])); }
        const double eprev = eres;
        eres = fmax(rdmax, rmax);
        if (eres <= FIN_INNER_TOL || in == FIN_INNER || (eres <= FIN_EPS && eres > FIN_STALL * eprev)) break;
        for (int i = 0; i < m; ++i) rc[i] = rho * w[i];
        matT_vec(G, rc, m, n, tmp);
        for (int i = 0; i < n; ++i) dq[i] = -rd[i] - tmp[i];
        chol_solve(K, n, dq);
        mat_vec(G, dq, m, n, ds);
        for (int i = 0; i < n; ++i) qf[i] += dq[i];
        for (int i = 0; i < m; ++i) if (act[i]) y[i] += rho * (ds[i] + w[i]);
      }
      /* (t holds G qf: every pass of the loop above evaluates it before it decides to stop) */
      for (int i = 0; i < n; ++i) dd[i] = qf[i] - xf[i];
      mat_vec(G, dd, m, n, ds);                          /* g_i . d */
      double rbest = INFINITY;
      int blk = -1;
      for (int i = 0; i < m; ++i) {
        slack[i] = h[i] - t[i];                           /* slack at x_A */
        if (!act[i] && ds[i] > FIN_GD_MIN) {
          const double r = fmax(slack[i] + ds[i], 0.0) / ds[i];     /* slack at x over the approach rate */
          if (r < rbest) { rbest = r; blk = i; }
        }
      }
      if (blk >= 0 && rbest < 1.0) {
        for (int i = 0; i < n; ++i) xf[i] += rbest * dd[i];
        act[blk] = 1;
        continue;
      }
      memcpy(xf, qf, sizeof(double) * n);
      double ymin = INFINITY, ymax = 0.0, smin = INFINITY;
      for (int i = 0; i < m; ++i) {
        if (act[i]) { ymin = fmin(ymin, y[i]); ymax = fmax(ymax, y[i]); }
        else smin = fmin(smin, slack[i]);
      }
      if (ymin < -FIN_EPS) {                             /* every negative multiplier leaves at once */
        for (int i = 0; i < m; ++i) if (act[i] && y[i] < -FIN_EPS) { act[i] = 0; y[i] = 0.0; }
        continue;
      }
      /* polish: an equality solve left above FIN_POLISH_TOL gets one more round on the same set at the stiffer penalty */
      if (eres > FIN_POLISH_TOL && rho == FIN_RHO && rnd < fin_rounds) { rho = FIN_RHO_POLISH; continue; }
      double qmax = 0.0;
      for (int i = 0; i < n; ++i) qmax = fmax(qmax, fabs(qf[i]));
      certified = fok && rmax <= FIN_EPS && rdmax <= FIN_EPS + FIN_DUAL_REL * ymax && smin >= -FIN_EPS && qmax < 1e300;
      if (diag) diag[3] = fmin(ymin, smin);
      break;
    }
    if (diag) { diag[0] = rounds; diag[1] = eres; }
    if (certified) memcpy(q, qf, sizeof(double) * n);
    else { status = LIPMPC_STATUS_UNCERTIFIED; for (int i = 0; i < m; ++i) act[i] = z[i] > FIN_IDENT * s[i]; }
  }
  *status_out = status;
  /* `working`: the set the finish certified with (interior mode / uncertified: the estimate z > 1e5 s);
   * `active`: the rows TIGHT at the returned point, slack <= LIPMPC_TIGHT_TOL (lipmpc_oracle.py: tight_set) -- unique
   * because the minimiser is -- and the tightness margin */
  if (working)
    for (int i = 0; i < m; ++i)
      if (act[i]) working[W->canon[i] >> 6] |= (uint64_t)1 << (W->canon[i] & 63);
  mat_vec(G, q, m, n, t);
  double tmargin = INFINITY;
  for (int i = 0; i < m; ++i) {
    const double sl = h[i] - t[i];
    if (sl <= LIPMPC_TIGHT_TOL) active[W->canon[i] >> 6] |= (uint64_t)1 << (W->canon[i] & 63);
    tmargin = fmin(tmargin, fabs(sl - LIPMPC_TIGHT_TOL));
  }
  if (diag) diag[4] = tmargin;

  /* ---- outputs: X*, U* (recover footsteps), objective incl. the k=0 term ------------------------ */
  double p[17][2], v[17][2];
  p[0][0] = p0[0]; p[0][1] = p0[1]; v[0][0] = v0[0]; v[0][1] = v0[1];
  double ob = 0.0;
  for (int k = 0; k <= N; ++k) {
    if (k >= 1) { p[k][0] = q[2 * (k - 1)]; p[k][1] = q[2 * (k - 1) + 1]; }
    ob += (p[k][0] - goal[0]) * (p[k][0] - goal[0]) + (p[k][1] - goal[1]) * (p[k][1] - goal[1]);
  }
  for (int k = 0; k < N; ++k)
    for (int a = 0; a < 2; ++a) {
      v[k + 1][a] = -v[k][a] + kap * (p[k + 1][a] - p[k][a]);
      U[2 * k + a] = (p[k + 1][a] - ch * p[k][a] - (sh / beta) * v[k][a]) / (1.0 - ch);
    }
  for (int k = 0; k <= N; ++k) { X[4 * k] = p[k][0]; X[4 * k + 1] = v[k][0]; X[4 * k + 2] = p[k][1]; X[4 * k + 3] = v[k][1]; }
  *obj = ob;
}

/* Host-pointer twin of lipmpc_plan_step_batch (same layouts), OpenMP over problems. */
int lipmpc_oracle_plan_step_batch(const lipmpc_params* P, int64_t B, const double* state, const double* goal,
                                  const int8_t* first_foot, const double* delta, const double* obs_xy,
                                  const int32_t* obs_nv, double* U, double* X, double* theta, double* omega,
                                  double* obj, int32_t* status, int32_t* iters, uint64_t* active, uint64_t* working,
                                  double* c_eta, double* diag, const double* bounds, const double* c_eta_in, int n_threads) {
  if (!P || P->N < 1 || P->N > 16 || P->n_obs_max < 0 || P->n_obs_max > 50) return LIPMPC_E_UNSUPPORTED;
  const int N = P->N, n_obs = P->n_obs_max;
  const int64_t words = (9 * N + (N + 1) * n_obs + 63) / 64;
  int nt = n_threads > 0 ? n_threads : 1;
  int err = 0;
#pragma omp parallel num_threads(nt)
  {
    work_t* W = (work_t*)malloc(sizeof(work_t));
    if (!W) {
#pragma omp atomic write
      err = 1;
    }
#pragma omp barrier
    if (!err) {
#pragma omp for schedule(dynamic, 8)
      for (int64_t b = 0; b < B; ++b) {
        plan_one(P, bounds ? bounds + b * 4 : NULL, W, state + b * 5, goal + b * 2, (int)first_foot[b], delta ? delta[b] : 0.0,
                 obs_xy ? obs_xy + (size_t)b * n_obs * P->v_max * 2 : NULL, obs_nv ? obs_nv + b * n_obs : NULL,
                 U + b * N * 2, X + b * (N + 1) * 4, theta + b * (N + 1), omega + b * N, obj + b, status + b, iters + b,
                 active + b * words, working ? working + b * words : NULL, c_eta ? c_eta + (size_t)b * n_obs * 4 : NULL,
                 diag ? diag + b * LIPMPC_DIAG_WORDS : NULL,
                 c_eta_in ? c_eta_in + (size_t)b * n_obs * 4 : NULL);
      }
    }
    free(W);
  }
  return err ? LIPMPC_E_NOMEM : LIPMPC_OK;
}

int lipmpc_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
