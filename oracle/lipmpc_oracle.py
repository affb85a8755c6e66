"""CPU oracle (numpy, float64) for the per-timestep LIP-MPC / LDCBF step QP.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

It restates, in plain numpy, the algorithm of the reference's hot path.  Citations are
``file:line`` relative to the reference checkout (``HumanoidNavigation/...``):

* LIP dynamics ``x+ = A x + B u``                      MPC/HumanoidMpc.py:34-48, 335-343
* heading preprocessing (theta / omega)                 MPC/HumanoidMpc.py:137-160
* closest point / unit normal / inside flip             Utils/ObstaclesUtils.py:50-109
* LDCBF rows  eta^T (p_k - c) - delta >= 0, k = 0..N    MPC/HumanoidMpc.py:252-294,
                                                        MPC/HumanoidMPCVariants/HumanoidMPCCustomLCBF.py:30-31
* leg reachability rows                                 MPC/HumanoidMpc.py:183-202, 232-236
* manoeuvrability rows                                  MPC/HumanoidMpc.py:204-219, 238-243
* walking-velocity rows                                 MPC/HumanoidMpc.py:162-181, 245-249
* cost  sum_{k=0..N} |p_k - goal|^2                     MPC/HumanoidMpc.py:321-333
* closed-loop driver                                    MPC/HumanoidMpc.py:380-459
* constants                                             config.yml:2-17, MPC/HumanoidMpc.py:20-22, :200

Parity pinning status
---------------------
* geometry (c, eta, inside flag): PINNED bit-for-bit/1e-12 against the reference's own
  ``ObstaclesUtils`` imported in the build container (fixtures in ``tests/golden``).
* the QP solve: the reference hands the problem to CasADi/IPOPT (``HumanoidMpc.py:97-100,417``,
  ``casadi`` unpinned in requirements.txt) which is not installable offline, and the
  reference has no tests or numeric fixtures -> **parity unpinned** for the solve itself.
  The restated problem is a strictly convex QP with a unique minimiser, so the oracle is
  defined as that exact minimiser; it is cross-checked here by two independent methods
  (a Mehrotra IPM + active-set polish, and Lawson-Hanson NNLS on the least-distance dual)
  and softly against closed-loop trajectories recovered from the reference's committed
  result PDFs (``Assets/ReportResults/*/evolutions``; IPOPT tol=1e-5 leaves those 1e-4..1e-3
  from the optimum, so that check is a regression, not a parity pin).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------------------
# constants (config.yml:2-17; HumanoidMpc.py:20-22; ell at HumanoidMpc.py:200)
# --------------------------------------------------------------------------------------


@dataclass
class Params:
    N: int = 3
    dt: float = 0.4                 # DELTA_T
    g: float = 9.81                 # GRAVITY_CONST
    h_com: float = 1.0              # COM_HEIGHT
    alpha: float = 3.6              # ALPHA
    l_max: tuple = (0.10, 0.10)     # L_MAX_X, L_MAX_Y
    l_min: tuple = (-0.1, -0.1)     # L_MIN_X, L_MIN_Y
    v_min: tuple = (-0.1, 0.1)      # V_MIN
    v_max: tuple = (0.8, 0.4)       # V_MAX
    omega_max: float = 0.156 * math.pi
    ell: float = 0.05
    sampling_time: float = 0.4
    # solver knobs (build-defined)
    tol: float = 1e-11
    tol_interior: float = 1e-9      # stop tolerance of the interior (non-finished) mode, see include/lipmpc.h
    max_iter: int = 60
    finish_rounds: int = 0      # 0 = FIN_ROUNDS
    k0_tol: float = 1e-5            # tolerated violation of the constant k=0 LDCBF rows
                                    # (= the reference's IPOPT constr_viol_tol, HumanoidMpc.py:99)
    presolve: bool = True           # exact mode: drop the LDCBF rows the leg-reach rows make redundant (presolve_ldcbf);
                                    # False = LIPMPC_FLAG_NO_PRESOLVE
    warm_start: bool = False        # LIPMPC_FLAG_WARM_START: a closed loop seeds every step with the previous one's result;
                                    # like the interior mode it keeps every row (ONE rule, also in the kernel and the C port:
                                    # presolve <=> exact and presolve and not warm_start -- whether or not this particular
                                    # step has a previous result to read)

    @property
    def beta(self):
        return math.sqrt(self.g / self.h_com)

    @property
    def ch(self):
        return math.cosh(self.beta * self.dt)

    @property
    def sh(self):
        return math.sinh(self.beta * self.dt)

    @property
    def kappa(self):
        return self.beta * self.sh / (self.ch - 1.0)


STATUS_SOLVED, STATUS_MAX_ITER, STATUS_INFEASIBLE, STATUS_DEGENERATE, STATUS_UNCERTIFIED = 0, 1, 2, 3, 4
TIGHT_TOL = 1e-7       # LIPMPC_TIGHT_TOL: a row is ACTIVE (canonical active set) when its slack at the returned point is below this
SCREEN_MARGIN = 1e-3   # presolve: an LDCBF row is dropped when the leg-reach rows keep it this far from active


def reach_step(P: Params):
    """Largest CoM displacement |p_{k+1} - p_k| the leg-reach rows allow (HumanoidMpc.py:183-202, 232-236:
    L_MIN <= R(theta_k)(p_{k+1} - p_k) + (0, s_k ell) <= L_MAX, a rotation of the displacement)."""
    dx = max(abs(P.l_max[0]), abs(P.l_min[0]))
    dy = max(abs(P.l_max[1]), abs(P.l_min[1])) + abs(P.ell)
    return math.sqrt(dx * dx + dy * dy)


def presolve_ldcbf(x0, cs, etas, delta, P: Params):
    """Which LDCBF rows (stage k = 1..N, obstacle j) the leg-reach rows make REDUNDANT: every feasible p_k lies within
    k * reach_step of p_0, so  eta_j.(p_k - c_j) - delta >= h0_j - |eta_j| k reach_step  with h0_j the row's value at p_0;
    where that bound exceeds SCREEN_MARGIN the row can never be active or violated and leaves the problem -- same feasible
    set, same minimiser, same active set (measured on the BASELINE fields: 2.8 of 80 rows stay at N = 8 / 10 obstacles, 14 of
    800 at N = 16 / 50 obstacles).  What the dropped rows did do in the interior-point iteration is average: their
    complementarity products pull mu = s.z / m and Mehrotra's sigma towards the well-centred value, and without them the
    iteration TAIL grows (26 -> 35 iterations at most on the bench batch).  That part is kept: the n_d dropped rows are
    replaced by n_d copies of ONE ballast row  0.q <= s_bar  (s_bar = their mean slack at p_0), whose step is a scalar
    recurrence (no direction, no entry in K or in the right-hand sides) -- with it the iteration counts are those of the
    full problem (mean 13.1 / max 27 against 13.05 / 26; at N = 16 / 50 obstacles 16.4 against 18.6).
    Returns (redundant [N, n_obs] bool, n_d, s_bar); the kernel and the C oracle evaluate the same expressions."""
    N, n_obs = P.N, len(cs)
    red = np.zeros((N, n_obs), bool)
    step = reach_step(P)
    px, py = float(x0[0]), float(x0[2])
    n_d, ssum = 0, 0.0
    for j in range(n_obs):
        ex, ey, cx, cy = float(etas[j][0]), float(etas[j][1]), float(cs[j][0]), float(cs[j][1])
        h0 = ((ex * px + ey * py) - (ex * cx + ey * cy)) - delta
        es = math.sqrt(ex * ex + ey * ey) * step
        nd_j = 0
        for k in range(1, N + 1):
            if h0 > es * k + SCREEN_MARGIN:
                red[k - 1, j] = True
                nd_j += 1
        n_d += nd_j
        ssum += nd_j * h0
    return red, n_d, (ssum / n_d if n_d else 0.0)


def lip_matrices(P: Params):
    """A_l, B_l of HumanoidMpc.py:34-48 (state (px,vx,py,vy), input = stance foot)."""
    b, ch, sh = P.beta, P.ch, P.sh
    Ad = np.array([[ch, sh / b], [sh * b, ch]])
    Bd = np.array([1.0 - ch, -b * sh])
    A = np.zeros((4, 4))
    A[:2, :2] = Ad
    A[2:, 2:] = Ad
    B = np.zeros((4, 2))
    B[:2, 0] = Bd
    B[2:, 1] = Bd
    return A, B


def precompute_theta_omega(x0, theta0, goal, P: Params):
    """HumanoidMpc.py:137-160 — the *current* position is used for every k; the multiplier is
    sampling_time; no angle wrapping."""
    theta = [float(theta0)]
    omega = []
    for _ in range(P.N):
        target = math.atan2(goal[1] - x0[2], goal[0] - x0[0]) - theta[-1]
        w = min(max(target, -P.omega_max), P.omega_max)
        omega.append(w)
        theta.append(theta[-1] + w * P.sampling_time)
    return np.array(theta), np.array(omega)


# --------------------------------------------------------------------------------------
# geometry (ObstaclesUtils.py:50-109)
# --------------------------------------------------------------------------------------

def point_in_ring(x, ring):
    """Crossing-number inside test on the closed CCW vertex ring; restates what
    ``matplotlib.path.Path(hull_vertices).contains_point(x)`` (radius 0) decides
    (ObstaclesUtils.py:50-57): a +X ray from x, edge endpoints classified by ``y >= x_y``,
    crossing counted when the edge/ray intersection lies at or to the right of x."""
    tx, ty = float(x[0]), float(x[1])
    V = len(ring)
    inside = False
    x0v, y0v = float(ring[V - 1][0]), float(ring[V - 1][1])
    f0 = y0v >= ty
    for i in range(V):
        x1v, y1v = float(ring[i][0]), float(ring[i][1])
        f1 = y1v >= ty
        if f0 != f1:
            if ((y1v - ty) * (x0v - x1v) >= (x1v - tx) * (y0v - y1v)) == f1:
                inside = not inside
        x0v, y0v, f0 = x1v, y1v, f1
    return inside


def closest_point_and_normal(x, ring):
    """ObstaclesUtils.py:60-109 on the CCW ring ``points[vertices]`` (edge i = ring[i] ->
    ring[i+1]).  Returns c(2), eta(2), inside flag, degenerate flag.

    t = clip((x-A).(B-A) / |B-A|^2, 0, 1) with |B-A|^2 formed as sqrt-then-square (:81),
    c = A + t (B-A) (:85), strictly-smaller distance wins (:92), eta = (x-c)/|x-c| (:97-104),
    negated when x is inside (:106-107)."""
    V = len(ring)
    px, py = float(x[0]), float(x[1])
    best = math.inf
    cx = cy = math.nan
    degenerate = False
    for i in range(V):
        ax, ay = float(ring[i][0]), float(ring[i][1])
        bx, by = float(ring[(i + 1) % V][0]), float(ring[(i + 1) % V][1])
        ex, ey = bx - ax, by - ay
        nrm = math.sqrt(ex * ex + ey * ey)
        den = nrm * nrm
        if den == 0.0:
            degenerate = True
            continue
        t = ((px - ax) * ex + (py - ay) * ey) / den
        t = max(0.0, min(1.0, t))
        qx, qy = ax + t * ex, ay + t * ey
        dx, dy = qx - px, qy - py
        d = math.sqrt(dx * dx + dy * dy)
        if d < best:
            best, cx, cy = d, qx, qy
    nx, ny = px - cx, py - cy
    nn = math.sqrt(nx * nx + ny * ny)
    if not (nn > 0.0):
        return np.array([cx, cy]), np.array([0.0, 0.0]), False, True
    nx, ny = nx / nn, ny / nn
    inside = point_in_ring((px, py), ring)
    if inside:
        nx, ny = -nx, -ny
    return np.array([cx, cy]), np.array([nx, ny]), inside, degenerate


def list_c_and_eta(x0, obstacles):
    """HumanoidMpc.py:296-319: one (c, eta) per obstacle, all at the *current* CoM."""
    cs, etas, degen = [], [], False
    for ring in obstacles:
        c, eta, _, dg = closest_point_and_normal((x0[0], x0[2]), ring)
        cs.append(c)
        etas.append(eta)
        degen = degen or dg
    if cs:
        return np.array(cs), np.array(etas), degen
    return np.zeros((0, 2)), np.zeros((0, 2)), degen


# --------------------------------------------------------------------------------------
# the step QP, two algebraically identical builders
# --------------------------------------------------------------------------------------

def tight_set(G, h, q):
    """The canonical active set of a point q: the rows with slack h_i - g_i.q <= TIGHT_TOL, and the tightness margin
    min_i |slack_i - TIGHT_TOL| (how far the nearest row is from changing sides).  The minimiser of the strictly convex step
    QP is unique, so this set is a function of the problem alone -- which the finish's WORKING set (rows with a positive
    multiplier in its certificate) is not at a degenerate vertex, where linearly dependent tight rows leave the multipliers
    non-unique.  This is what "active-constraint indices bit-exact" (BASELINE north_star) is checked on."""
    if G.shape[0] == 0:
        return np.zeros(0, bool), math.inf
    slack = h - G @ q
    return slack <= TIGHT_TOL, float(np.min(np.abs(slack - TIGHT_TOL)))


def n_rows(N, n_obs):
    """canonical inequality count: reach 4N | manoeuvr N | vel 4N | LDCBF (N+1) n_obs."""
    return 9 * N + (N + 1) * n_obs


def build_qp_position_form(x0, theta, omega, goal, s_v, cs, etas, delta, P: Params):
    """Position form: variables q = (p_1..p_N) in R^{2N}; rows G q <= h in
    the canonical order reach | manoeuvr | vel | LDCBF(k=0..N).  The k=0 LDCBF rows have a
    zero G row and h = eta.(p0 - c) - delta (they are constants of the step).

    v_k = -v_{k-1} + kappa (p_k - p_{k-1}) follows from eliminating u_k between the position
    and velocity rows of x+ = A x + B u (HumanoidMpc.py:34-48)."""
    N = P.N
    n = 2 * N
    kap = P.kappa
    n_obs = len(cs)
    p0 = np.array([x0[0], x0[2]])
    v0 = np.array([x0[1], x0[3]])
    # affine maps  p_k = Pm[k] q + pc[k],  v_k = Vm[k] q + vc[k]
    Pm = np.zeros((N + 1, 2, n))
    pc = np.zeros((N + 1, 2))
    pc[0] = p0
    for k in range(1, N + 1):
        Pm[k, 0, 2 * (k - 1)] = 1.0
        Pm[k, 1, 2 * (k - 1) + 1] = 1.0
    Vm = np.zeros((N + 1, 2, n))
    vc = np.zeros((N + 1, 2))
    vc[0] = v0
    for k in range(1, N + 1):
        Vm[k] = -Vm[k - 1] + kap * (Pm[k] - Pm[k - 1])
        vc[k] = -vc[k - 1] + kap * (pc[k] - pc[k - 1])
    rows, rhs = [], []
    # leg reachability (HumanoidMpc.py:183-202, 232-236): upper (x,y) then lower (x,y)
    for k in range(N):
        c_, s_ = math.cos(theta[k]), math.sin(theta[k])
        R = np.array([[c_, s_], [-s_, c_]])
        Dm = R @ (Pm[k + 1] - Pm[k])
        dc = R @ (pc[k + 1] - pc[k]) + np.array([0.0, s_v[k] * P.ell])
        for a in range(2):
            rows.append(Dm[a]); rhs.append(P.l_max[a] - dc[a])
        for a in range(2):
            rows.append(-Dm[a]); rhs.append(-(P.l_min[a] - dc[a]))
    # manoeuvrability (HumanoidMpc.py:204-219, 238-243): state k+1, theta_{k+1}, omega_k
    for k in range(N):
        c_, s_ = math.cos(theta[k + 1]), math.sin(theta[k + 1])
        r = np.array([c_, s_])
        rows.append(r @ Vm[k + 1])
        rhs.append(P.v_max[0] - (P.alpha / math.pi) * abs(omega[k]) - r @ vc[k + 1])
    # walking velocities (HumanoidMpc.py:162-181, 245-249): k = 1..N, s_v[k] on the cos*vy term
    for k in range(1, N + 1):
        c_, s_ = math.cos(theta[k]), math.sin(theta[k])
        W = np.array([[c_, s_], [-s_, c_ * s_v[k]]])
        Wm = W @ Vm[k]
        wc = W @ vc[k]
        for a in range(2):
            rows.append(Wm[a]); rhs.append(P.v_max[a] - wc[a])
        for a in range(2):
            rows.append(-Wm[a]); rhs.append(-(P.v_min[a] - wc[a]))
    # LDCBF (HumanoidMpc.py:252-294; CustomLCBF.py:30-31): k = 0..N, obstacle order
    for k in range(N + 1):
        for j in range(n_obs):
            rows.append(-(etas[j] @ Pm[k]))
            rhs.append(etas[j] @ (pc[k] - cs[j]) - delta)
    G = np.array(rows).reshape(-1, n)
    h = np.array(rhs)
    g = np.tile(np.asarray(goal, float), N)
    return G, h, g


def build_qp_reference_form(x0, theta, omega, goal, s_v, cs, etas, delta, P: Params):
    """The same rows written the way the reference writes them — on (X, U) through A_l, B_l
    (HumanoidMpc.py:221-249, 284-292, 321-333) — then condensed in U by propagating the
    dynamics.  Returns G_U, h, H_U, f_U, and the affine map p = T u + t0 to positions.  Used
    only by tests, to prove the position form is the same problem."""
    N = P.N
    A, B = lip_matrices(P)
    nu = 2 * N
    # X_k = Xm[k] U + xc[k]
    Xm = np.zeros((N + 1, 4, nu))
    xc = np.zeros((N + 1, 4))
    xc[0] = np.asarray(x0[:4], float)
    for k in range(N):
        Sel = np.zeros((2, nu)); Sel[0, 2 * k] = 1.0; Sel[1, 2 * k + 1] = 1.0
        Xm[k + 1] = A @ Xm[k] + B @ Sel
        xc[k + 1] = A @ xc[k]
    rows, rhs = [], []
    for k in range(N):
        c_, s_ = math.cos(theta[k]), math.sin(theta[k])
        dpm = np.stack([Xm[k + 1][0] - Xm[k][0], Xm[k + 1][2] - Xm[k][2]])
        dpc = np.array([xc[k + 1][0] - xc[k][0], xc[k + 1][2] - xc[k][2]])
        lm = np.stack([c_ * dpm[0] + s_ * dpm[1], -s_ * dpm[0] + c_ * dpm[1]])
        lc = np.array([c_ * dpc[0] + s_ * dpc[1], -s_ * dpc[0] + c_ * dpc[1] + s_v[k] * P.ell])
        for a in range(2):
            rows.append(lm[a]); rhs.append(P.l_max[a] - lc[a])
        for a in range(2):
            rows.append(-lm[a]); rhs.append(-(P.l_min[a] - lc[a]))
    for k in range(N):
        c_, s_ = math.cos(theta[k + 1]), math.sin(theta[k + 1])
        rows.append(c_ * Xm[k + 1][1] + s_ * Xm[k + 1][3])
        rhs.append(P.v_max[0] - (P.alpha / math.pi) * abs(omega[k]) - (c_ * xc[k + 1][1] + s_ * xc[k + 1][3]))
    for k in range(1, N + 1):
        c_, s_ = math.cos(theta[k]), math.sin(theta[k])
        lm = np.stack([c_ * Xm[k][1] + s_ * Xm[k][3], -s_ * Xm[k][1] + c_ * s_v[k] * Xm[k][3]])
        lc = np.array([c_ * xc[k][1] + s_ * xc[k][3], -s_ * xc[k][1] + c_ * s_v[k] * xc[k][3]])
        for a in range(2):
            rows.append(lm[a]); rhs.append(P.v_max[a] - lc[a])
        for a in range(2):
            rows.append(-lm[a]); rhs.append(-(P.v_min[a] - lc[a]))
    for k in range(N + 1):
        pm = np.stack([Xm[k][0], Xm[k][2]])
        pcst = np.array([xc[k][0], xc[k][2]])
        for j in range(len(cs)):
            rows.append(-(etas[j] @ pm)); rhs.append(etas[j] @ (pcst - cs[j]) - delta)
    G = np.array(rows).reshape(-1, nu)
    h = np.array(rhs)
    T = np.zeros((2 * N, nu)); t0 = np.zeros(2 * N)
    for k in range(1, N + 1):
        T[2 * (k - 1)] = Xm[k][0]; T[2 * (k - 1) + 1] = Xm[k][2]
        t0[2 * (k - 1)] = xc[k][0]; t0[2 * (k - 1) + 1] = xc[k][2]
    gg = np.tile(np.asarray(goal, float), N)
    H = 2.0 * T.T @ T
    f = 2.0 * T.T @ (t0 - gg)
    return G, h, H, f, T, t0


# --------------------------------------------------------------------------------------
# solvers
# --------------------------------------------------------------------------------------

@dataclass
class QPResult:
    q: np.ndarray
    z: np.ndarray
    s: np.ndarray
    status: int
    iters: int
    active: np.ndarray = field(default=None)
    margin: float = math.inf
    rounds: int = 0
    cert_margin: float = 0.0     # min(smallest active multiplier, smallest inactive slack) of the certificate


# solver constants shared (by value) with oracle/lipmpc_oracle.c and the HIP kernel
IPM_S_FLOOR = 0.1      # initial slack floor
IPM_Z0 = 30.0          # initial multiplier
IPM_STEP_FRAC = 0.995  # fraction to the boundary
IPM_Z_DIVERGE = 1e13   # multiplier blow-up => infeasible
IPM_STALL_TOL = 1e-6   # Cholesky breakdown below this (r_p, mu) counts as converged
IPM_SLOW_FROM = 8      # from this iteration on (earlier, mu legitimately stalls while infeasibility is traded in):
IPM_SLOW_RATIO = 0.9   # mu / previous mu between 0.9 and 1 = the iteration made (almost) no progress ...
IPM_SLOW_SIGMA = 0.5   # ... then centre at least 0 .. this much, linearly (breaks the limit cycles of plain Mehrotra steps;
                       # a ramp, not a switch: a threshold would let two implementations part ways on a rounding)
FIN_RHO = 1e10         # penalty of the active-set equality solve
FIN_EPS = 1e-9         # sign / violation threshold of the certificate
FIN_ROUNDS = 8         # default cap on active-set rounds (Params.finish_rounds = 0) for N <= 8
FIN_ROUNDS_LONG = 16   # ... and for longer horizons
FIN_IDENT = 1e5        # initial working set: z_i > FIN_IDENT * s_i (see finish_active_set)
FIN_INNER = 6          # max multiplier iterations per equality solve
FIN_INNER_TOL = 1e-11
FIN_STALL = 0.5        # a correction that leaves more than this share of the residual has stalled (stop if already <= FIN_EPS)
FIN_RHO_POLISH = 1e12  # penalty of the polish round (see finish_active_set, 5.)
FIN_POLISH_TOL = 1e-10 # an equality solve left above this is polished before it may certify
FIN_GD_MIN = 1e-14     # ratio test: a direction component below this does not run into its row
FIN_DUAL_REL = 1e-14   # stationarity tolerance of the certificate: FIN_EPS + this x largest multiplier (rounding floor of G_A^T y)
WARM_Z_MIN, WARM_Z_MAX = 3.0, 100.0   # warm start of a closed loop: previous multipliers, shifted by one stage, clipped to this band


def solve_qp_ipm(G, h, g, q0, tol=1e-11, max_iter=60, z0=None):
    """min |q-g|^2 s.t. G q <= h by Mehrotra predictor-corrector on the normal equations
    K = 2I + G^T diag(z/s) G (slack form G q + s = h, s,z > 0).  Start: q0 (the caller passes
    "stand still", p_k = p_0), s = max(h - G q0, 0.1), z = 30; sigma = (mu_aff/mu)^3, raised towards 0.5 after an
    iteration (from the 8th on) that left mu above 0.9 of its previous value (linear ramp up to ratio 1).  Stop when max|r_p| <= tol and
    mu = s.z/m <= tol.  The dual residual is not part of the test: on the normal equations it
    stalls near cond(K)*eps, and the active-set finish below recomputes q exactly anyway."""
    m, n = G.shape
    q = np.array(q0, float).copy()
    if m == 0:
        return QPResult(q=g.copy(), z=np.zeros(0), s=np.zeros(0), status=STATUS_SOLVED, iters=0)
    s = np.maximum(h - G @ q, IPM_S_FLOOR)
    z = np.full(m, IPM_Z0) if z0 is None else np.array(z0, float).copy()      # z0: warm start (shift_warm_start)
    status = STATUS_MAX_ITER
    it = 0
    mu_prev = math.inf
    for it in range(max_iter + 1):
        rd = 2.0 * (q - g) + G.T @ z
        rp = G @ q + s - h
        mu = float(s @ z) / m
        if np.max(np.abs(rp)) <= tol and mu <= tol:
            status = STATUS_SOLVED
            break
        if it == max_iter:
            break
        if not (np.max(z) < IPM_Z_DIVERGE) or not np.all(np.isfinite(q)):
            status = STATUS_INFEASIBLE
            break
        d = z / s
        K = 2.0 * np.eye(n) + (G.T * d) @ G
        try:
            L = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:
            # K = 2I + G^T D G stops being numerically positive definite once max(z/s) ~ 1e15.
            # Close to the solution that is "converged to working precision" (the finish takes
            # over); anywhere else it is the signature of an infeasible problem.
            near = np.max(np.abs(rp)) <= IPM_STALL_TOL and mu <= IPM_STALL_TOL
            status = STATUS_SOLVED if near else STATUS_INFEASIBLE
            break

        def kkt_solve(rc):
            # Newton step for  2dq + G^T dz = -rd ; G dq + ds = -rp ; z ds + s dz = -rc
            w = (z * rp - rc) / s
            dq = np.linalg.solve(L.T, np.linalg.solve(L, -rd - G.T @ w))
            ds = -rp - G @ dq
            dz = -(rc + z * ds) / s
            return dq, ds, dz

        dq_a, ds_a, dz_a = kkt_solve(s * z)
        a_aff = min(1.0, _max_step(s, ds_a), _max_step(z, dz_a))
        mu_aff = float((s + a_aff * ds_a) @ (z + a_aff * dz_a)) / m
        sigma = (mu_aff / mu) ** 3
        # Safeguard: plain Mehrotra steps can fall into a limit cycle (two complementarity pairs trading places,
        # mu constant, steps of 0.6) — seen once per ~4000 problems on LiDAR-inferred obstacle sets.  An iteration
        # that did not reduce mu by 10 % is followed by one that centres at least half way.
        if it >= IPM_SLOW_FROM:
            ramp = min(1.0, max(0.0, (mu / mu_prev - IPM_SLOW_RATIO) * (1.0 / (1.0 - IPM_SLOW_RATIO))))
            sigma = max(sigma, IPM_SLOW_SIGMA * ramp)
        mu_prev = mu
        dq, ds, dz = kkt_solve(s * z + ds_a * dz_a - sigma * mu)
        a = min(1.0, IPM_STEP_FRAC * min(_max_step(s, ds), _max_step(z, dz)))
        q = q + a * dq
        s = s + a * ds
        z = z + a * dz
    return QPResult(q=q, z=z, s=s, status=status, iters=it)


def _max_step(v, dv):
    """largest a with v + a dv >= 0 (inf when nothing blocks)"""
    neg = dv < 0
    if not np.any(neg):
        return math.inf
    return float(np.min(-v[neg] / dv[neg]))


def eqp_multiplier_method(G, h, g, active, q, y_full, rho=None):
    """min |q-g|^2 s.t. G_A q = h_A by the method of multipliers in residual-correction form,
    warm-started at the current point: K_A = 2I + rho G_A^T G_A (one Cholesky), then repeat
      rd = 2(q-g) + G_A^T y ; r = G_A q - h_A ; dq = -K_A^{-1}(rd + rho G_A^T r) ;
      q += dq ; y += rho (G_A dq + r)
    until max(|rd|,|r|) <= 1e-11 (at most 6 times, or until a correction stalls below 1e-9).  Dependent active rows are
    harmless.  Returns q, y, max|rd|, max|r| of the last evaluation."""
    n = G.shape[1]
    rho = FIN_RHO if rho is None else rho
    GA, hA = G[active], h[active]
    y = y_full[active].copy()
    q = q.copy()
    if GA.shape[0] == 0:
        return g.copy(), y, 0.0, 0.0
    K = 2.0 * np.eye(n) + rho * GA.T @ GA
    L = np.linalg.cholesky(K)
    res = math.inf
    for _ in range(FIN_INNER + 1):
        rd = 2.0 * (q - g) + GA.T @ y
        r = GA @ q - hA
        prev = res
        rdmax, rmax = float(np.max(np.abs(rd))), float(np.max(np.abs(r)))
        res = max(rdmax, rmax)
        # converged, out of corrections, or stalled on its rounding floor below what the certificate needs (FIN_EPS)
        if res <= FIN_INNER_TOL or _ == FIN_INNER or (res <= FIN_EPS and res > FIN_STALL * prev):
            break
        dq = np.linalg.solve(L.T, np.linalg.solve(L, -rd - rho * (GA.T @ r)))
        q = q + dq
        y = y + rho * (GA @ dq + r)
    return q, y, rdmax, rmax


def finish_active_set(G, h, g, res: QPResult, rounds_cap: int = 0):
    """Turn the interior-point estimate into the exact minimiser with a KKT certificate: a PRIMAL active-set method
    started at the interior-point iterate x (feasible to the IPM's 1e-11) on the working set A <- {i : z_i > 1e5 s_i}.
    Per round (one factorisation of K_A):
      1. x_A = minimiser on the working set (equality solve above, warm-started at x);
      2. ratio test along d = x_A - x over the rows outside A that d runs into (g_i.d > FIN_GD_MIN):
         alpha = min_i max(slack_i(x), 0) / g_i.d.  If alpha < 1 the step is BLOCKED: x <- x + alpha d, the blocking row
         (lowest canonical index among ties) joins A; next round;
      3. otherwise x <- x_A; the rows of A whose multiplier is < -1e-9 leave A, all of them at once (any subset may leave
         without losing feasibility or descent -- the minimiser on a smaller set is no worse -- and rows that are needed
         after all come back through the ratio test; one at a time the worst problem of the bench batch took 8 rounds, so 5);
         next round;
      4. otherwise x is feasible with non-negative multipliers: the unique optimum of the strictly convex QP -- certified
         when the equality solve met the tolerances (constraint residual 1e-9; stationarity 1e-9 + FIN_DUAL_REL x the largest
         multiplier: the residual 2(q-g) + G_A^T y cannot be evaluated below the rounding of its own terms, which grow
         with y -- IPOPT scales its dual infeasibility by the multiplier size in the same way);
      5. ... after a POLISH round where the equality solve was left above FIN_POLISH_TOL: the multiplier iteration contracts
         the error along a singular direction sigma of G_A by 2 / (2 + rho sigma^2) per correction, so two active rows within
         1e-5 of parallel (sigma ~ 6e-6: rho sigma^2 = 0.4 at rho = 1e10) leave the vertex 2e-6 off after the six
         corrections of a round -- measured against a 40-digit solve of the same working set -- while the residuals
         already read 3e-9.  Such a round (3 % of the N = 16 / 50-obstacle problems, 0.3 % at N = 8 / 10) is followed by one
         at rho = FIN_RHO_POLISH on the same set: 1e-11 from the 40-digit vertex.
    x stays feasible and the objective never increases, so the rounds cannot wander: a blocked step never crosses a row, and a
    blocking row is never (nearly) dependent on A -- g_i.d = 0 for every row in the span of A, d being in A's null space -- so
    the working set stays independent where the add-the-most-violated / drop-the-most-negative exchange of rounds 1-2
    cycled through the dependent rows of degenerate vertices (1.1-1.6 % of the N = 16 / 50-obstacle problems uncertified after
    10 rounds, some after 64; the ratio-test rounds certify all 4096 of them within 10).  The initial set is a deliberate
    UNDER-estimate (rows with a substantial multiplier): a wrongly included, nearly degenerate row makes the
    multipliers of the first equality solve non-unique and their signs unreliable."""
    m = G.shape[0]
    nz = np.any(G != 0.0, axis=1)
    A = (res.z > FIN_IDENT * res.s) & nz
    x, yf = res.q.copy(), np.where(A, res.z, 0.0)
    cap = rounds_cap if rounds_cap > 0 else FIN_ROUNDS
    rho = FIN_RHO
    for rnd in range(1, cap + 1):
        try:
            xw, y, rdmax, rmax = eqp_multiplier_method(G, h, g, A, x, yf, rho)
        except np.linalg.LinAlgError:                      # K_A not numerically positive definite at this penalty
            return x, yf, None, A, rnd, None
        yf = np.zeros(m); yf[A] = y
        slack_w = h - G @ xw
        d = xw - x
        gd = G @ d
        sl = slack_w + gd                                  # slack at x
        cand = ~A & nz & (gd > FIN_GD_MIN)
        if cand.any():
            ratio = np.where(cand, np.maximum(sl, 0.0) / np.where(cand, gd, 1.0), math.inf)
            blk = int(np.argmin(ratio))                    # ties: lowest canonical index
            if ratio[blk] < 1.0:
                x = x + ratio[blk] * d
                A[blk] = True
                continue
        x = xw
        neg = A & (yf < -FIN_EPS)
        if neg.any():                                      # every negative multiplier leaves at once
            A = A & ~neg
            yf[neg] = 0.0
            continue
        if max(rdmax, rmax) > FIN_POLISH_TOL and rho == FIN_RHO and rnd < cap:
            rho = FIN_RHO_POLISH                            # 5. polish: one more round on the same set, stiffer penalty
            continue
        ymax = float(np.max(yf[A])) if A.any() else 0.0
        smin = float(np.min(slack_w[~A & nz])) if (~A & nz).any() else math.inf
        if rmax <= FIN_EPS and rdmax <= FIN_EPS + FIN_DUAL_REL * ymax and smin >= -FIN_EPS and np.all(np.isfinite(x)):
            cert = min(float(np.min(yf[A])) if A.any() else math.inf, smin)
            return x, yf, slack_w, A, rnd, cert
        return x, yf, None, A, rnd, None
    return x, yf, None, A, cap, None


def solve_qp_exact(G, h, g, q0, tol=1e-11, max_iter=60, finish_rounds=0, z0=None, ballast=(0, 0.0)):
    """IPM, then the certified active-set finish.  status 4 = IPM converged but the finish did
    not certify within its round budget (the IPM point is returned).
    ballast = (n_d, s_bar): the interior-point phase runs with n_d extra rows 0.q <= s_bar (presolve_ldcbf); the finish
    sees the real rows only."""
    n_d, s_bar = ballast
    if n_d:
        Gb = np.vstack([G, np.zeros((n_d, G.shape[1]))])
        hb = np.concatenate([h, np.full(n_d, s_bar)])
        res = solve_qp_ipm(Gb, hb, g, q0, tol=tol, max_iter=max_iter, z0=z0)
        res.z, res.s = res.z[:G.shape[0]], res.s[:G.shape[0]]
    else:
        res = solve_qp_ipm(G, h, g, q0, tol=tol, max_iter=max_iter, z0=z0)
    res.z_ipm = res.z.copy()
    res.active = np.zeros(G.shape[0], bool)
    if res.status != STATUS_SOLVED or G.shape[0] == 0:
        return res
    nz = np.any(G != 0.0, axis=1)
    with np.errstate(divide="ignore"):
        lr = np.abs(np.log(res.z[nz] / (FIN_IDENT * res.s[nz])))
    res.margin = float(np.min(lr)) if lr.size else math.inf
    q, y, slack, A, rounds, cert = finish_active_set(G, h, g, res, finish_rounds)
    res.rounds = rounds
    res.cert_margin = cert if cert is not None else 0.0
    if cert is not None:
        res.q, res.z, res.s, res.active = q, y, np.maximum(slack, 0.0), A
    else:
        res.status = STATUS_UNCERTIFIED
        res.active = (res.z > FIN_IDENT * res.s) & nz
    return res


def nnls_lawson_hanson(A, b, tol=1e-13, max_outer=None):
    """min |A x - b| s.t. x >= 0 — Lawson & Hanson (1974), algorithm NNLS, written out here
    because it must be independent of the IPM (scipy's ``nnls`` returned inconsistent
    residuals on these matrices).  Passive-set least squares via numpy ``lstsq``."""
    m, n = A.shape
    passive = np.zeros(n, bool)
    x = np.zeros(n)
    w = A.T @ (b - A @ x)
    max_outer = max_outer or 5 * n
    for _ in range(max_outer):
        cand = np.where(~passive)[0]
        if cand.size == 0 or np.max(w[cand]) <= tol:
            break
        j = cand[np.argmax(w[cand])]
        passive[j] = True
        for _inner in range(5 * n):
            sP = np.linalg.lstsq(A[:, passive], b, rcond=None)[0]
            s = np.zeros(n); s[passive] = sP
            if np.min(sP) > 0.0:
                x = s
                break
            bad = passive & (s <= 0.0)
            alpha = np.min(x[bad] / (x[bad] - s[bad]))
            x = x + alpha * (s - x)
            passive &= ~(np.abs(x) <= 1e-15 * max(1.0, np.max(np.abs(x))))
            passive &= ~((x <= 0.0))
            x[~passive] = 0.0
        w = A.T @ (b - A @ x)
    return x


def solve_qp_ldp_nnls(G, h, g):
    """Independent check: Lawson & Hanson least-distance programming.  With x = q - g the QP is
    min |x| s.t. (-G) x >= (G g - h); LDP solves it through NNLS on E = [(-G)^T; (Gg-h)^T],
    f = e_{n+1}: if the residual r = E u - f is non-zero then x = -r[:n] / r[n]."""
    nz = np.any(G != 0.0, axis=1)
    Gn, hn = -G[nz], (G @ g - h)[nz]
    n = G.shape[1]
    f = np.zeros(n + 1); f[n] = 1.0
    tau, x = 1.0, None
    for _ in range(2):  # second pass rescales by |x| so that r[n] = 1/(1+|x/tau|^2) is O(1)
        E = np.vstack([Gn.T, hn[None, :] / tau])
        u = nnls_lawson_hanson(E, f)
        r = E @ u - f
        if abs(r[n]) < 1e-13:
            return None  # infeasible
        x = -r[:n] / r[n] * tau
        tau = max(1.0, float(np.linalg.norm(x)))
    return g + x


# --------------------------------------------------------------------------------------
# one MPC step and the closed loop
# --------------------------------------------------------------------------------------

def foot_window(step_number, N, start_with_right_foot=True):
    """HumanoidMpc.py:104-108, 399-403."""
    base = 0 if start_with_right_foot else 1
    return [1 if (i % 2) == base else -1 for i in range(step_number, step_number + N + 1)]


def recover_trajectory(q, x0, P: Params):
    """X*(N+1,4), U*(N,2) from positions (Appendix B item 6)."""
    N = P.N
    ch, sh, b, kap = P.ch, P.sh, P.beta, P.kappa
    p = np.vstack([[x0[0], x0[2]], q.reshape(N, 2)])
    v = np.zeros((N + 1, 2)); v[0] = [x0[1], x0[3]]
    U = np.zeros((N, 2))
    for k in range(N):
        v[k + 1] = -v[k] + kap * (p[k + 1] - p[k])
        U[k] = (p[k + 1] - ch * p[k] - (sh / b) * v[k]) / (1.0 - ch)
    X = np.stack([p[:, 0], v[:, 0], p[:, 1], v[:, 1]], axis=1)
    return X, U


def shift_warm_start(q_prev, z_prev, N, n_obs):
    """Warm start of the next step of a closed loop from the previous step's interior-point result (what the
    reference does with the primal part, HumanoidMpc.py:450-455: the next solve is seeded with the shifted previous
    prediction): stage k takes over stage k+1's position and multipliers (canonical row order, k = 0 LDCBF rows
    included as zeros), the last stage extrapolates its position by one more step and keeps its multipliers;
    multipliers are clipped to [WARM_Z_MIN, WARM_Z_MAX] -- a fresh start is z = 30 everywhere."""
    qp = np.asarray(q_prev, float).reshape(N, 2)
    q0 = np.vstack([qp[1:], qp[-1] + (qp[-1] - qp[-2])]).ravel() if N >= 2 else qp.ravel().copy()
    z = np.asarray(z_prev, float)
    zs = z.copy()
    for k in range(N - 1):
        zs[4 * k:4 * k + 4] = z[4 * (k + 1):4 * (k + 1) + 4]                                  # reach
        zs[4 * N + k] = z[4 * N + k + 1]                                                      # manoeuvrability
        zs[5 * N + 4 * k:5 * N + 4 * k + 4] = z[5 * N + 4 * (k + 1):5 * N + 4 * (k + 1) + 4]  # velocity
        zs[9 * N + (k + 1) * n_obs:9 * N + (k + 2) * n_obs] = z[9 * N + (k + 2) * n_obs:9 * N + (k + 3) * n_obs]   # LDCBF stage k+1 <- k+2
    return q0, np.clip(zs, WARM_Z_MIN, WARM_Z_MAX)


def plan_step(state, goal, first_foot, obstacles, delta, P: Params, exact=True, warm=None):
    """One MPC step.  state = (px,vx,py,vy,theta); first_foot = s_0 in {+1,-1};
    obstacles = list of CCW rings (V_j,2).  Returns a dict mirroring the C ABI outputs.
    warm = (q0 [2N], z0 [canonical rows]) from shift_warm_start, or None for the cold start."""
    N = P.N
    x0 = np.asarray(state[:4], float)
    theta, omega = precompute_theta_omega(x0, state[4], goal, P)
    s_v = [first_foot if (i % 2 == 0) else -first_foot for i in range(N + 1)]
    cs, etas, degen = list_c_and_eta(x0, obstacles)
    n_obs = len(obstacles)
    m_tot = n_rows(N, n_obs)
    out = dict(theta=theta, omega=omega, c=cs, eta=etas, status=STATUS_SOLVED, iters=0,
               U=np.full((N, 2), np.nan), X=np.full((N + 1, 4), np.nan), obj=math.nan,
               active=np.zeros(m_tot, bool), working=np.zeros(m_tot, bool), margin=math.inf, tight_margin=math.inf)
    if degen or not np.all(np.isfinite(etas)):
        out["status"] = STATUS_DEGENERATE
        return out
    G, h, g = build_qp_position_form(x0, theta, omega, goal, s_v, cs, etas, delta, P)
    # k = 0 LDCBF rows are constants of the step (zero G row): they can only make it
    # infeasible, and are kept out of the solve; their canonical indices are never "active".
    k0 = slice(9 * N, 9 * N + n_obs)
    if n_obs and np.any(h[k0] < -P.k0_tol):
        out["status"] = STATUS_INFEASIBLE
        return out
    keep = np.ones(m_tot, bool)
    keep[k0] = False
    n_ballast, s_ballast = 0, 0.0
    if exact and P.presolve and not P.warm_start and n_obs:
        red, n_ballast, s_ballast = presolve_ldcbf(x0, cs, etas, delta, P)
        for k in range(1, N + 1):
            keep[9 * N + k * n_obs:9 * N + (k + 1) * n_obs] &= ~red[k - 1]
    Gs, hs = G[keep], h[keep]
    q0 = np.tile([x0[0], x0[2]], N)
    z0 = None
    if warm is not None:
        q0, z0 = np.asarray(warm[0], float), np.asarray(warm[1], float)[keep]
    fin = P.finish_rounds if P.finish_rounds > 0 else (FIN_ROUNDS if N <= 8 else FIN_ROUNDS_LONG)
    res = solve_qp_exact(Gs, hs, g, q0, tol=P.tol, max_iter=P.max_iter, finish_rounds=fin, z0=z0,
                         ballast=(n_ballast, s_ballast)) if exact else \
        solve_qp_ipm(Gs, hs, g, q0, tol=P.tol_interior, max_iter=P.max_iter, z0=z0)
    out["status"], out["iters"], out["rounds"] = res.status, res.iters, res.rounds
    if res.status not in (STATUS_SOLVED, STATUS_UNCERTIFIED):
        return out
    X, U = recover_trajectory(res.q, x0, P)
    out["X"], out["U"] = X, U
    p = X[:, [0, 2]]
    out["obj"] = float(np.sum((p - np.asarray(goal, float)) ** 2))
    # `active`: the rows tight at the returned point (canonical, unique); `working`: the set the finish certified with
    # (interior mode / uncertified: the interior-point estimate z > 1e5 s)
    tight, out["tight_margin"] = tight_set(Gs, hs, res.q)
    out["active"][keep] = tight
    if res.active is not None:
        out["working"][keep] = res.active
    elif Gs.shape[0]:
        out["working"][keep] = res.z > FIN_IDENT * res.s
    out["margin"] = res.margin
    out["cert_margin"] = res.cert_margin
    out["q"] = res.q
    # interior-point state for the next step's warm start: positions and multipliers of the IPM phase, canonical rows
    z_full = np.zeros(m_tot)
    z_full[keep] = getattr(res, "z_ipm", res.z)
    out["q_ipm"], out["z_ipm"] = getattr(res, "q_ipm", res.q), z_full
    return out


def run_closed_loop(goal, obstacles, N_horizon=3, N_mpc_timesteps=100, sampling_time=0.4,
                    init_state=(0, 0, 0, 0, 0), start_with_right_foot=True, delta=0.0,
                    params: Params | None = None, exact=False, warm_start=False):
    """HumanoidMpc.py:345-459 (``run_simulation`` without plotting): returns X_pred (5,K+1),
    U_pred (3,K) with the reference's truncation rule ``X[:, :k+1], U[:, :k]`` (:457-459).

    ``exact=False`` advances with the interior-point iterate (strictly inside every half-space,
    like the reference's IPOPT iterate); the exact vertex solution puts the CoM *on* LDCBF
    boundaries, where the next step's eta = (x-c)/|x-c| (ObstaclesUtils.py:104) is 0/0."""
    P = params or Params()
    P.N = N_horizon
    P.sampling_time = sampling_time
    P.warm_start = bool(warm_start)
    mpc_step = int(P.dt / sampling_time) or 1
    num_inputs = mpc_step * N_mpc_timesteps
    X_pred = np.zeros((5, num_inputs + 1))
    U_pred = np.zeros((3, num_inputs))
    X_pred[:, 0] = np.asarray(init_state, float)
    last_obj = math.inf
    u_keep = np.zeros(2)
    warm = None
    iters_log = []
    k = 0
    for k in range(num_inputs):
        is_mpc = (k % mpc_step) == 0
        if last_obj < 0.05:
            break
        st = X_pred[:, k]
        step_number = k // mpc_step
        s0 = foot_window(step_number, 0, start_with_right_foot)[0]
        if is_mpc:
            r = plan_step(st, goal, s0, obstacles, delta, P, exact=exact, warm=warm if warm_start else None)
            if r["status"] not in (STATUS_SOLVED, STATUS_UNCERTIFIED):
                break
            iters_log.append(r["iters"])
            if warm_start and P.N >= 2:
                warm = shift_warm_start(r["q_ipm"], r["z_ipm"], P.N, len(obstacles))
            last_obj = r["obj"]
            u_keep = r["U"][0]
            theta1, omega0 = r["theta"][1], r["omega"][0]
        else:
            th, om = precompute_theta_omega(st[:4], st[4], goal, P)
            theta1, omega0 = th[1], om[0]
        U_pred[:2, k] = u_keep
        U_pred[2, k] = omega0
        if is_mpc:
            A, B = lip_matrices(P)
            X_pred[:4, k + 1] = A @ st[:4] + B @ u_keep
        else:
            X_pred[:4, k + 1] = st[:4]
        X_pred[4, k + 1] = theta1
    run_closed_loop.last_iters = iters_log          # interior-point iterations per solve of the last call (diagnostics)
    return X_pred[:, :k + 1], U_pred[:, :k]
