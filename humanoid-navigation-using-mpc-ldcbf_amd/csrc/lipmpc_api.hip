// lipmpc_api.hip — C ABI (include/lipmpc.h): handle, parameter checks, kernel dispatch, state advance.
#include "lipmpc_kernel.hpp"

using namespace lipmpc_dev;

namespace {
// state advance (HumanoidMpc.py:432-447)
__global__ void advance_kernel(long B, double ch, double sh_over_beta, double beta_sh, double* __restrict__ state,
                               int8_t* __restrict__ foot, const double* __restrict__ U, const double* __restrict__ theta,
                               const int32_t* __restrict__ status, int N) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int st = status[b];
  if (st != LIPMPC_STATUS_SOLVED && st != LIPMPC_STATUS_UNCERTIFIED) return;
  double* x = state + b * 5;
  const double ux = U[b * N * 2 + 0], uy = U[b * N * 2 + 1];
  const double px = x[0], vx = x[1], py = x[2], vy = x[3];
  // A_l x + B_l u  (HumanoidMpc.py:34-48)
  x[0] = ch * px + sh_over_beta * vx + (1.0 - ch) * ux;
  x[1] = beta_sh * px + ch * vx - beta_sh * ux;
  x[2] = ch * py + sh_over_beta * vy + (1.0 - ch) * uy;
  x[3] = beta_sh * py + ch * vy - beta_sh * uy;
  x[4] = theta[b * (N + 1) + 1];
  foot[b] = (int8_t)(-foot[b]);
}

// one sample of a host-driven fleet loop (include/lipmpc.h: lipmpc_fleet_update_batch)
__global__ void fleet_update_kernel(long B, int k_max, double stop_obj, double ch, double sh_over_beta, double beta_sh, int N,
                                    double* __restrict__ state, int8_t* __restrict__ foot, int8_t* __restrict__ walking,
                                    double* __restrict__ last_obj, int32_t* __restrict__ n_steps,
                                    int32_t* __restrict__ last_status, int32_t* __restrict__ n_overflow,
                                    const int32_t* __restrict__ sample, double* __restrict__ X_pred,
                                    double* __restrict__ U_pred, const double* __restrict__ U,
                                    const double* __restrict__ theta, const double* __restrict__ omega,
                                    const double* __restrict__ obj, const int32_t* __restrict__ status,
                                    const int32_t* __restrict__ overflow) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int k = *sample;
  if (k >= k_max) return;
  bool w = walking[b] != 0 && last_obj[b] >= stop_obj;
  // a scan whose clusters did not fit the obstacle slots is a failed sample: planning against a truncated
  // obstacle list would let the robot walk through something it sensed
  const int st = (overflow && overflow[b]) ? LIPMPC_STATUS_SENSOR_OVERFLOW : status[b];
  if (w && overflow) n_overflow[b] += overflow[b];
  if (w) last_status[b] = st;
  w = w && (st == LIPMPC_STATUS_SOLVED || st == LIPMPC_STATUS_UNCERTIFIED);
  double* x = state + b * 5;
  const double ux = U[b * N * 2 + 0], uy = U[b * N * 2 + 1];
  if (w) {
    last_obj[b] = obj[b];
    const double px = x[0], vx = x[1], py = x[2], vy = x[3];
    x[0] = ch * px + sh_over_beta * vx + (1.0 - ch) * ux;
    x[1] = beta_sh * px + ch * vx - beta_sh * ux;
    x[2] = ch * py + sh_over_beta * vy + (1.0 - ch) * uy;
    x[3] = beta_sh * py + ch * vy - beta_sh * uy;
    x[4] = theta[b * (N + 1) + 1];
    foot[b] = (int8_t)(-foot[b]);
    n_steps[b] += 1;
  }
  walking[b] = w ? 1 : 0;
  double* up = U_pred + (b * (long)k_max + k) * 3;
  up[0] = ux; up[1] = uy; up[2] = omega[b * N];
  double* xp = X_pred + (b * (long)(k_max + 1) + k + 1) * 5;
  for (int i = 0; i < 5; ++i) xp[i] = x[i];
}
__global__ void fleet_next_sample_kernel(int32_t* sample) { *sample += 1; }

// The order of the NEXT step launch on a schedule from the costs this launch left: problems by descending cost (counting
// sort, one workgroup; which of two equally costly problems comes first is immaterial).
__global__ __launch_bounds__(1024) void order_by_cost_kernel(long B, int32_t* __restrict__ sched) {
  __shared__ int cursor_[SCHED_COST_BINS];
  const int32_t* w = sched + SCHED_ORDER + B;
  int32_t* order = sched + SCHED_ORDER;
  for (int k = threadIdx.x; k < SCHED_COST_BINS; k += blockDim.x) cursor_[k] = 0;
  __syncthreads();
  for (long i = threadIdx.x; i < B; i += blockDim.x) atomicAdd(&cursor_[SCHED_COST_BINS - 1 - min(max(w[i], 0), SCHED_COST_BINS - 1)], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < SCHED_COST_BINS; ++k) { const int c = cursor_[k]; cursor_[k] = run; run += c; }
  }
  __syncthreads();
  for (long i = threadIdx.x; i < B; i += blockDim.x)
    order[atomicAdd(&cursor_[SCHED_COST_BINS - 1 - min(max(w[i], 0), SCHED_COST_BINS - 1)], 1)] = (int32_t)i;
  if (threadIdx.x == 0) sched[SCHED_VALID] = (int32_t)B;
}

// Index lists per class from the sort keys (class, cost bucket): a STABLE counting sort in one workgroup -- every thread owns a
// contiguous run of problems and counts its keys (bytes in LDS); per key one wave scans the counts over the threads (eight
// threads' bytes per lane as one 64-bit word, a wave prefix sum over the lanes); then the writes -- so the lists, and with them
// which problems share a wave, are a function of the batch alone.  The list of a class is the concatenation of its buckets,
// dearest first.
constexpr int BIN_THREADS = 512;
constexpr int BIN_KEYS = SPLIT_CLASSES * SPLIT_BUCKETS;
constexpr long BIN_MAX_B = 255L * BIN_THREADS;          // a thread's run holds at most 255 problems (byte counters)
__device__ __forceinline__ int bin_key(int k) { return min(max(k, 0), BIN_KEYS - 1); }
__global__ __launch_bounds__(BIN_THREADS) void split_bin_kernel(long B, int32_t* __restrict__ ws) {
  __shared__ __attribute__((aligned(8))) unsigned char cnt_[BIN_KEYS][BIN_THREADS];
  __shared__ int start_[BIN_KEYS][BIN_THREADS / 8];      // problems with this key in the runs of the threads before lane's eight
  __shared__ int total_[BIN_KEYS], base_[BIN_KEYS];
  const int t = threadIdx.x;
  const long per = (B + BIN_THREADS - 1) / BIN_THREADS, lo = min((long)t * per, B), hi = min(lo + per, B);
  const int32_t* key = ws + SPLIT_HEAD;
  unsigned* z = reinterpret_cast<unsigned*>(&cnt_[0][0]);
  for (int i = t; i < BIN_KEYS * BIN_THREADS / 4; i += BIN_THREADS) z[i] = 0u;
  __syncthreads();
  for (long i = lo; i < hi; ++i) cnt_[bin_key(key[i])][t] += 1;      // (own column: no race)
  __syncthreads();
  const int wave = t >> 6, lane = t & 63;
  for (int k = wave; k < BIN_KEYS; k += BIN_THREADS / 64) {
    const unsigned long long w = *reinterpret_cast<const unsigned long long*>(&cnt_[k][8 * lane]);
    int mine = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) mine += (int)((w >> (8 * u)) & 0xffull);
    int incl = mine;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, 64); if (lane >= m) incl += o; }
    start_[k][lane] = incl - mine;
    if (lane == 63) total_[k] = incl;
  }
  __syncthreads();
  if (t < SPLIT_CLASSES) {
    int run = 0;
    for (int q = 0; q < SPLIT_BUCKETS; ++q) { base_[t * SPLIT_BUCKETS + q] = run; run += total_[t * SPLIT_BUCKETS + q]; }
    ws[t] = run;
  }
  __syncthreads();
  for (long i = lo; i < hi; ++i) {
    const int k = bin_key(key[i]);
    // position: the key's base inside its class list + the runs of the earlier threads + this thread's earlier problems of the key
    int pos = base_[k] + start_[k][t >> 3];
    for (int u = t & ~7; u < t; ++u) pos += cnt_[k][u];
    for (long j = lo; j < i; ++j) pos += (bin_key(key[j]) == k) ? 1 : 0;
    ws[SPLIT_HEAD + B * (1 + k / SPLIT_BUCKETS) + pos] = (int32_t)i;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
struct lipmpc_handle {
  lipmpc_params p;
  KArgs k;
  int device;
  int G;
  int nobs_l;
  int nvar;     // variable slots of the factorisation: G, or 8 (horizons up to 4, register-row instantiations)
  int32_t* sched;       // optional schedule buffer (lipmpc_set_schedule), device memory owned by the caller
  int64_t sched_cap;    // largest batch it holds
  int32_t* ws;          // optional split-launch workspace (lipmpc_set_workspace), device memory owned by the caller
  int64_t ws_cap;
  hipStream_t side[SPLIT_CLASSES - 1];     // the solver bodies of a split launch run side by side (created with the workspace)
  hipEvent_t fork_ev, join_ev[SPLIT_CLASSES - 1];
  bool have_streams;
};

extern "C" {

int lipmpc_default_params(lipmpc_params* p) {
  if (!p) return LIPMPC_E_ARG;
  memset(p, 0, sizeof(*p));
  p->N = 3; p->n_obs_max = 0; p->v_max = 5; p->max_iter = 60; p->flags = 0; p->finish_rounds = 0;
  p->dt = 0.4; p->g = 9.81; p->h_com = 1.0; p->alpha = 3.6;
  p->l_max[0] = 0.10; p->l_max[1] = 0.10; p->l_min[0] = -0.1; p->l_min[1] = -0.1;
  p->v_min[0] = -0.1; p->v_min[1] = 0.1; p->v_max_xy[0] = 0.8; p->v_max_xy[1] = 0.4;
  p->omega_max = 0.156 * M_PI; p->ell = 0.05; p->sampling_time = 0.4;
  p->tol = 1e-11; p->tol_interior = 1e-9; p->k0_tol = 1e-5;
  return LIPMPC_OK;
}

int64_t lipmpc_num_rows(const lipmpc_params* p) { return p ? 9L * p->N + (long)(p->N + 1) * p->n_obs_max : LIPMPC_E_ARG; }
int64_t lipmpc_active_words(const lipmpc_params* p) { return p ? (lipmpc_num_rows(p) + 63) / 64 : LIPMPC_E_ARG; }

int lipmpc_create(const lipmpc_params* p, int device, lipmpc_handle** out) {
  if (!p || !out) return LIPMPC_E_ARG;
  if (p->N < 1 || p->N > 16 || p->n_obs_max < 0 || p->n_obs_max > 50 || p->v_max < 3 || p->v_max > 32 ||
      p->max_iter < 1 || p->finish_rounds < 0 || p->finish_rounds > 64 || !(p->tol > 0.0) || !(p->tol_interior > 0.0) || !(p->dt > 0.0) || !(p->h_com > 0.0) || !(p->g > 0.0))
    return LIPMPC_E_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return LIPMPC_E_HIP;
  lipmpc_handle* h = (lipmpc_handle*)calloc(1, sizeof(lipmpc_handle));
  if (!h) return LIPMPC_E_NOMEM;
  h->p = *p;
  h->device = device;
  h->G = (p->N <= 8) ? 16 : 32;
  const int need = (p->n_obs_max + 1) / 2;
  h->nobs_l = need == 0 ? 0 : need <= 2 ? 2 : need <= 5 ? 5 : need <= 7 ? 7 : need <= 13 ? 13 : 25;
  h->nvar = (h->G == 16 && p->N <= 4 && h->nobs_l <= 7) ? 8 : h->G;
  KArgs& k = h->k;
  k.N = p->N; k.n_obs = p->n_obs_max; k.nvert_max = p->v_max; k.max_iter = p->max_iter; k.flags = p->flags;
  k.fin_rounds = p->finish_rounds > 0 ? p->finish_rounds : (p->N <= 8 ? FIN_ROUNDS : FIN_ROUNDS_LONG);
  k.m_tot = (int)lipmpc_num_rows(p); k.words = (int)lipmpc_active_words(p);
  const double beta = sqrt(p->g / p->h_com), ch = cosh(beta * p->dt), sh = sinh(beta * p->dt);
  k.kappa = beta * sh / (ch - 1.0); k.ch = ch; k.sh_over_beta = sh / beta; k.inv_one_minus_ch = 1.0 / (1.0 - ch);
  k.beta_sh = beta * sh;
  for (int i = 0; i < 2; ++i) { k.l_max[i] = p->l_max[i]; k.l_min[i] = p->l_min[i]; k.v_min[i] = p->v_min[i]; k.v_max[i] = p->v_max_xy[i]; }
  k.alpha_over_pi = p->alpha / M_PI; k.omega_max = p->omega_max; k.ell = p->ell; k.tau = p->sampling_time;
  k.tol = (p->flags & LIPMPC_FLAG_INTERIOR) ? p->tol_interior : p->tol; k.k0_tol = p->k0_tol;
  {   // presolve bound (oracle: reach_step): the largest |p_{k+1} - p_k| the leg-reach rows allow
#pragma clang fp contract(off)
    const double dx = fmax(fabs(p->l_max[0]), fabs(p->l_min[0])), dy = fmax(fabs(p->l_max[1]), fabs(p->l_min[1])) + fabs(p->ell);
    k.reach_step = sqrt(dx * dx + dy * dy);
  }
  *out = h;
  return LIPMPC_OK;
}

static void drop_streams(lipmpc_handle* h) {
  if (!h->have_streams) return;
  (void)hipSetDevice(h->device);
  for (int i = 0; i < SPLIT_CLASSES - 1; ++i) { (void)hipStreamDestroy(h->side[i]); (void)hipEventDestroy(h->join_ev[i]); }
  (void)hipEventDestroy(h->fork_ev);
  h->have_streams = false;
}

void lipmpc_destroy(lipmpc_handle* h) {
  if (!h) return;
  drop_streams(h);
  free(h);
}

// does this handle's step run as a split launch when it has a workspace?  (32 lanes per problem, obstacles, presolve on)
static bool split_capable(const lipmpc_handle* h) {
  return h->G == 32 && h->nobs_l > 0 && !(h->p.flags & (LIPMPC_FLAG_INTERIOR | LIPMPC_FLAG_NO_PRESOLVE | LIPMPC_FLAG_WARM_START));
}

int64_t lipmpc_workspace_bytes(const lipmpc_handle* h, int64_t capacity) {
  if (!h || capacity < 0) return LIPMPC_E_ARG;
  return split_capable(h) ? (int64_t)sizeof(int32_t) * (SPLIT_HEAD + (1 + SPLIT_CLASSES) * capacity) : 0;
}

int lipmpc_set_workspace(lipmpc_handle* h, void* workspace, int64_t capacity) {
  if (!h || capacity < 0 || (workspace && capacity > 0x0fffffff)) return LIPMPC_E_ARG;
  h->ws = (capacity > 0 && split_capable(h)) ? (int32_t*)workspace : nullptr;
  h->ws_cap = h->ws ? capacity : 0;
  if (h->ws && !h->have_streams) {
    if (hipSetDevice(h->device) != hipSuccess) return LIPMPC_E_HIP;
    bool ok = hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) == hipSuccess;
    // (plain streams: side streams of the highest priority, tried so that the dearer bodies' waves are placed first, made the
    // launch 35 % slower -- the high-priority queues delay the next launch's classification and the fork / join events)
    for (int i = 0; ok && i < SPLIT_CLASSES - 1; ++i)
      ok = hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&h->join_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { h->ws = nullptr; h->ws_cap = 0; return LIPMPC_E_HIP; }      // (a partial set leaks a few handles at process scope only)
    h->have_streams = true;
  }
  return LIPMPC_OK;
}

#define LAUNCH(GG, NL, NV)                                                                                     \
  launch_plan_step<GG, NL, NV>(h->k, (long)B, state, goal, first_foot, delta, obs_xy, obs_nv, U, X, theta, omega, obj, \
                               status, iters, (unsigned long long*)active, (unsigned long long*)working, c_eta, diag, bounds, c_eta_in, sched, overflow, stream)

static int plan_step_impl(lipmpc_handle* h, int64_t B, const double* state, const double* goal,
                          const int8_t* first_foot, const double* delta, const double* obs_xy,
                          const int32_t* obs_nv, const double* c_eta_in, double* U, double* X, double* theta, double* omega,
                          double* obj, int32_t* status, int32_t* iters, uint64_t* active, uint64_t* working, double* c_eta,
                          double* diag, const double* bounds, const int32_t* overflow, void* hip_stream) {
  if (!h || B < 0) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state || !goal || !first_foot || !U || !X || !theta || !omega || !obj || !status || !iters || !active)
    return LIPMPC_E_ARG;
  if (h->p.n_obs_max > 0 && !c_eta_in && (!obs_xy || !obs_nv)) return LIPMPC_E_ARG;
  if (hipSetDevice(h->device) != hipSuccess) return LIPMPC_E_HIP;
  hipStream_t stream = (hipStream_t)hip_stream;
  int32_t* sched = (h->sched && B <= h->sched_cap) ? h->sched : nullptr;
  if (h->ws && B <= h->ws_cap && B <= BIN_MAX_B && split_capable(h)) {
    // Split launch: classes -> lists -> one kernel per solver body, side by side.  The rare, long bodies go first on their own
    // streams (a few waves each, they must not queue behind 2048 short ones); the caller's stream takes the 1-slot body and
    // waits for the others.
    constexpr int GPW = WAVE / 32;
    const unsigned blocks = (unsigned)((B + GPW - 1) / GPW);
    int32_t* ws = h->ws;
    int32_t* cost = sched ? sched + SCHED_ORDER + B : nullptr;
    hipLaunchKernelGGL((classify_kernel<32>), dim3(blocks), dim3(WAVE), 0, stream, h->k, (long)B, state, goal, delta, obs_xy, obs_nv,
                       bounds, c_eta_in, ws);
    hipLaunchKernelGGL(split_bin_kernel, dim3(1), dim3(BIN_THREADS), 0, stream, (long)B, ws);
    const int top = split_class_of((h->p.n_obs_max + 1) / 2);
#define LIST(NL, CLS, ST)                                                                                                     \
  launch_solve_list<32, NL, 32>(h->k, (long)B, CLS, ws, state, goal, first_foot, delta, obs_xy, obs_nv, U, X, theta, omega, obj, status, \
                                iters, (unsigned long long*)active, (unsigned long long*)working, c_eta, diag, bounds, c_eta_in, cost,  \
                                overflow, ST)
    static_assert(split_slots(0) == 1 && split_slots(1) == 2 && split_slots(2) == 4 && split_slots(3) == 13 && split_slots(4) == 25,
                  "the LIST() calls below name the bodies of the classes");
    // Which body goes where: the waves of a kernel are placed in launch order and a kernel that starts first takes the free
    // SIMDs first.  The 1-slot body has the most problems and the cheapest ones: it goes LAST, on a side stream (a side
    // stream's kernel starts ~10 us after the caller's stream's: the fork event); the 2-slot body -- a third of the problems,
    // dearer ones -- takes the caller's stream and with it the first pick of the SIMDs; the rare bodies with more slots start
    // next to it.  (Measured at N = 16 / 50 obstacles, B = 4096: 1-slot body on the caller's stream 0.640 ms, this 0.6xx.)
    if (hipEventRecord(h->fork_ev, stream) != hipSuccess) return LIPMPC_E_HIP;
    const int on_main = top >= 1 ? 1 : 0;
    for (int c = top; c >= 0; --c) {
      if (c == on_main) continue;
      const int si = c < on_main ? c : c - 1;                                // four side streams for the four other classes
      hipStream_t st = h->side[si];
      if (hipStreamWaitEvent(st, h->fork_ev, 0) != hipSuccess) return LIPMPC_E_HIP;
      switch (c) {
        case 4: LIST(25, 4, st); break;
        case 3: LIST(13, 3, st); break;
        case 2: LIST(4, 2, st); break;
        case 1: LIST(2, 1, st); break;
        default: LIST(1, 0, st); break;
      }
      if (hipEventRecord(h->join_ev[si], st) != hipSuccess) return LIPMPC_E_HIP;
    }
    if (on_main == 1) LIST(2, 1, stream); else LIST(1, 0, stream);
    for (int c = top; c >= 0; --c) {
      if (c == on_main) continue;
      if (hipStreamWaitEvent(stream, h->join_ev[c < on_main ? c : c - 1], 0) != hipSuccess) return LIPMPC_E_HIP;
    }
#undef LIST
    if (sched) hipLaunchKernelGGL(order_by_cost_kernel, dim3(1), dim3(1024), 0, stream, (long)B, sched);
    return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
  }
  if (h->G == 16 && h->nvar == 8) {
    switch (h->nobs_l) {
      case 0: LAUNCH(16, 0, 8); break;
      case 2: LAUNCH(16, 2, 8); break;
      case 5: LAUNCH(16, 5, 8); break;
      default: LAUNCH(16, 7, 8); break;
    }
  } else if (h->G == 16) {
    switch (h->nobs_l) {
      case 0: LAUNCH(16, 0, 16); break;
      case 2: LAUNCH(16, 2, 16); break;
      case 5: LAUNCH(16, 5, 16); break;
      case 7: LAUNCH(16, 7, 16); break;
      case 13: LAUNCH(16, 13, 16); break;
      default: LAUNCH(16, 25, 16); break;
    }
  } else {
    switch (h->nobs_l) {
      case 0: LAUNCH(32, 0, 32); break;
      case 2: LAUNCH(32, 2, 32); break;
      case 5: LAUNCH(32, 5, 32); break;
      case 7: LAUNCH(32, 7, 32); break;
      case 13: LAUNCH(32, 13, 32); break;
      default: LAUNCH(32, 25, 32); break;
    }
  }
  if (sched) hipLaunchKernelGGL(order_by_cost_kernel, dim3(1), dim3(1024), 0, stream, (long)B, sched);
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}

int lipmpc_set_schedule(lipmpc_handle* h, int32_t* schedule, int64_t capacity) {
  if (!h || capacity < 0 || (schedule && capacity > 0x3fffffff)) return LIPMPC_E_ARG;
  h->sched = capacity > 0 ? schedule : nullptr;
  h->sched_cap = h->sched ? capacity : 0;
  return LIPMPC_OK;
}

int64_t lipmpc_schedule_words(int64_t B) { return B < 0 ? LIPMPC_E_ARG : SCHED_ORDER + 2L * B; }

int lipmpc_plan_step_batch(lipmpc_handle* h, int64_t B, const double* state, const double* goal,
                           const int8_t* first_foot, const double* delta, const double* obs_xy,
                           const int32_t* obs_nv, double* U, double* X, double* theta, double* omega,
                           double* obj, int32_t* status, int32_t* iters, uint64_t* active, uint64_t* working, double* c_eta,
                           double* diag, const double* bounds, void* hip_stream) {
  return plan_step_impl(h, B, state, goal, first_foot, delta, obs_xy, obs_nv, nullptr, U, X, theta, omega, obj, status, iters,
                        active, working, c_eta, diag, bounds, nullptr, hip_stream);
}

int lipmpc_plan_step_batch_c_eta(lipmpc_handle* h, int64_t B, const double* state, const double* goal,
                                 const int8_t* first_foot, const double* delta, const double* c_eta_in,
                                 const int32_t* overflow, double* U, double* X, double* theta, double* omega, double* obj,
                                 int32_t* status, int32_t* iters, uint64_t* active, uint64_t* working, double* diag,
                                 const double* bounds, void* hip_stream) {
  if (h && h->p.n_obs_max > 0 && !c_eta_in) return LIPMPC_E_ARG;
  return plan_step_impl(h, B, state, goal, first_foot, delta, nullptr, nullptr, c_eta_in, U, X, theta, omega, obj, status,
                        iters, active, working, nullptr, diag, bounds, overflow, hip_stream);
}

#define LAUNCH_RO(GG, NL, NV)                                                                                    \
  launch_rollout<GG, NL, NV>(h->k, (long)B, k_max, mpc_step, stop_obj, state0, goal, first_foot, delta, obs_xy, obs_nv, \
                             X_pred, U_pred, n_steps, last_status, total_iters, bounds, stream)

int lipmpc_rollout_batch(lipmpc_handle* h, int64_t B, int32_t k_max, int32_t mpc_step, double stop_obj,
                         const double* state0, const double* goal, const int8_t* first_foot, const double* delta,
                         const double* obs_xy, const int32_t* obs_nv, double* X_pred, double* U_pred,
                         int32_t* n_steps, int32_t* last_status, int32_t* total_iters, const double* bounds,
                         void* hip_stream) {
  if (!h || B < 0 || k_max < 1 || mpc_step < 1) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state0 || !goal || !first_foot || !X_pred || !U_pred || !n_steps || !last_status || !total_iters) return LIPMPC_E_ARG;
  if (h->p.n_obs_max > 0 && (!obs_xy || !obs_nv)) return LIPMPC_E_ARG;
  if (hipSetDevice(h->device) != hipSuccess) return LIPMPC_E_HIP;
  hipStream_t stream = (hipStream_t)hip_stream;
  if (h->G == 16 && h->nvar == 8) {
    switch (h->nobs_l) {
      case 0: LAUNCH_RO(16, 0, 8); break;
      case 2: LAUNCH_RO(16, 2, 8); break;
      case 5: LAUNCH_RO(16, 5, 8); break;
      default: LAUNCH_RO(16, 7, 8); break;
    }
  } else if (h->G == 16) {
    switch (h->nobs_l) {
      case 0: LAUNCH_RO(16, 0, 16); break;
      case 2: LAUNCH_RO(16, 2, 16); break;
      case 5: LAUNCH_RO(16, 5, 16); break;
      case 7: LAUNCH_RO(16, 7, 16); break;
      case 13: LAUNCH_RO(16, 13, 16); break;
      default: LAUNCH_RO(16, 25, 16); break;
    }
  } else {
    switch (h->nobs_l) {
      case 0: LAUNCH_RO(32, 0, 32); break;
      case 2: LAUNCH_RO(32, 2, 32); break;
      case 5: LAUNCH_RO(32, 5, 32); break;
      case 7: LAUNCH_RO(32, 7, 32); break;
      case 13: LAUNCH_RO(32, 13, 32); break;
      default: LAUNCH_RO(32, 25, 32); break;
    }
  }
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}

int lipmpc_sense_plan_step_batch(lipmpc_handle* h, int64_t B, int32_t resolution, int32_t n_env, int32_t v_env,
                                 int32_t env_shared, double lidar_range, double eps, int32_t min_samples,
                                 const double* state, const double* goal, const int8_t* first_foot, const double* delta,
                                 const double* env_xy, const int32_t* env_nv, const double* ray_table, const double* noise,
                                 double* c_eta, int32_t* n_inferred, int32_t* overflow, int32_t* schedule,
                                 double* U, double* X, double* theta, double* omega, double* obj, int32_t* status,
                                 int32_t* iters, uint64_t* active, uint64_t* working, double* diag, const double* bounds,
                                 void* hip_stream) {
  if (!h || B < 0) return LIPMPC_E_ARG;
  if (h->p.n_obs_max < 1) return LIPMPC_E_UNSUPPORTED;        // a handle without obstacle slots has nothing to sense into
  if (!c_eta || !goal || !first_foot || !U || !X || !theta || !omega || !obj || !status || !iters || !active) return LIPMPC_E_ARG;
  const int rc = lipmpc_lidar_c_eta_batch(h->device, B, resolution, n_env, v_env, env_shared, lidar_range, eps, min_samples,
                                          h->p.n_obs_max, h->p.v_max, state, env_xy, env_nv, ray_table, noise, c_eta, n_inferred,
                                          overflow, nullptr, nullptr, nullptr, nullptr, schedule, hip_stream);
  if (rc != LIPMPC_OK) return rc;
  // the scan's overflow flags go to the solve: a robot whose clusters did not fit the obstacle slots gets
  // LIPMPC_STATUS_SENSOR_OVERFLOW and NaN outputs instead of a plan against the truncated list
  return lipmpc_plan_step_batch_c_eta(h, B, state, goal, first_foot, delta, c_eta, overflow, U, X, theta, omega, obj, status, iters,
                                      active, working, diag, bounds, hip_stream);
}

int lipmpc_advance_batch(lipmpc_handle* h, int64_t B, double* state, int8_t* first_foot, const double* U,
                         const double* theta, const int32_t* status, void* hip_stream) {
  if (!h || B < 0) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state || !first_foot || !U || !theta || !status) return LIPMPC_E_ARG;
  if (hipSetDevice(h->device) != hipSuccess) return LIPMPC_E_HIP;
  hipLaunchKernelGGL(advance_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     (long)B, h->k.ch, h->k.sh_over_beta, h->k.beta_sh, state, first_foot, U, theta, status, h->k.N);
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}

int lipmpc_fleet_update_batch(lipmpc_handle* h, int64_t B, int32_t k_max, double stop_obj, double* state,
                              int8_t* first_foot, int8_t* walking, double* last_obj, int32_t* n_steps,
                              int32_t* last_status, int32_t* n_overflow, int32_t* sample, double* X_pred, double* U_pred,
                              const double* U, const double* theta, const double* omega, const double* obj,
                              const int32_t* status, const int32_t* overflow, void* hip_stream) {
  if (!h || B < 0 || k_max < 1) return LIPMPC_E_ARG;
  if (B == 0) return LIPMPC_OK;
  if (!state || !first_foot || !walking || !last_obj || !n_steps || !last_status || !sample || !X_pred || !U_pred || !U ||
      !theta || !omega || !obj || !status || (overflow && !n_overflow))
    return LIPMPC_E_ARG;
  if (hipSetDevice(h->device) != hipSuccess) return LIPMPC_E_HIP;
  hipLaunchKernelGGL(fleet_update_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (long)B,
                     k_max, stop_obj, h->k.ch, h->k.sh_over_beta, h->k.beta_sh, h->k.N, state, first_foot, walking, last_obj,
                     n_steps, last_status, n_overflow, sample, X_pred, U_pred, U, theta, omega, obj, status, overflow);
  hipLaunchKernelGGL(fleet_next_sample_kernel, dim3(1), dim3(1), 0, (hipStream_t)hip_stream, sample);
  return hipGetLastError() == hipSuccess ? LIPMPC_OK : LIPMPC_E_HIP;
}

const char* lipmpc_strerror(int code) {
  switch (code) {
    case LIPMPC_OK: return "ok";
    case LIPMPC_E_ARG: return "invalid argument (null pointer or negative size)";
    case LIPMPC_E_UNSUPPORTED: return "unsupported parameters (N 1..16, n_obs_max 0..50, v_max 3..32)";
    case LIPMPC_E_HIP: return "HIP runtime error (no device, bad device index or launch failure)";
    case LIPMPC_E_NOMEM: return "out of host memory";
    default: return "unknown error";
  }
}

#ifdef LIPMPC_PHASE_TIMING      // an instrumented build writes phase counters where the product writes diag: not loadable as the product
int lipmpc_version(void) { return LIPMPC_ABI_VERSION + LIPMPC_VARIANT_BASE; }
#else
int lipmpc_version(void) { return LIPMPC_ABI_VERSION; }
#endif

}  // extern "C"
