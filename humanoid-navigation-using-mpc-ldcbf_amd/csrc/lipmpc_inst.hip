// lipmpc_inst.hip — one explicit instantiation of the step kernel per object file:
// compiled with -DINST_G=<16|32> -DINST_NL=<0|2|5|7|13|25> -DINST_NV=<variable slots: INST_G, or 8 for horizons up to 4> (see Makefile).
// -DINST_LIST: the solver body of the split launch with INST_NL row slots per lane instead (solve_list_kernel: one kernel per
// body, each with its own register allocation).
#include "lipmpc_kernel.hpp"

namespace lipmpc_dev {

#ifdef INST_LIST
template <int G, int NL, int NVAR>
void launch_solve_list(const KArgs& k, long B, int cls, const int32_t* ws, const double* state, const double* goal,
                       const int8_t* first_foot, const double* delta, const double* obs_xy, const int32_t* obs_nv, double* U,
                       double* X, double* theta, double* omega, double* obj, int32_t* status, int32_t* iters,
                       unsigned long long* active, unsigned long long* working, double* c_eta, double* diag, const double* bounds,
                       const double* c_eta_in, int32_t* cost_out, const int32_t* overflow_in, hipStream_t stream) {
  constexpr int GPW = WAVE / G;
  const unsigned blocks = (unsigned)((B + GPW - 1) / GPW);      // the whole batch's grid: the list's length lives on the device
  hipLaunchKernelGGL((solve_list_kernel<G, NL, NVAR>), dim3(blocks), dim3(WAVE), 0, stream, k, B, cls, ws, state, goal, first_foot, delta,
                     obs_xy, obs_nv, U, X, theta, omega, obj, status, iters, active, working, c_eta, diag, bounds, c_eta_in, cost_out,
                     overflow_in);
}
template void launch_solve_list<INST_G, INST_NL, INST_NV>(const KArgs&, long, int, const int32_t*, const double*, const double*,
                                                          const int8_t*, const double*, const double*, const int32_t*, double*, double*,
                                                          double*, double*, double*, int32_t*, int32_t*, unsigned long long*,
                                                          unsigned long long*, double*, double*, const double*, const double*, int32_t*,
                                                          const int32_t*, hipStream_t);
#else

template <int G, int NOBS_L, int NVAR>
void launch_plan_step(const KArgs& k, long B, const double* state, const double* goal, const int8_t* first_foot,
                      const double* delta, const double* obs_xy, const int32_t* obs_nv, double* U, double* X,
                      double* theta, double* omega, double* obj, int32_t* status, int32_t* iters,
                      unsigned long long* active, unsigned long long* working, double* c_eta, double* diag, const double* bounds,
                      const double* c_eta_in, int32_t* sched, const int32_t* overflow_in, hipStream_t stream) {
  constexpr int GPW = WAVE / G;
  const unsigned blocks = (unsigned)((B + GPW - 1) / GPW);
  // exact mode with the presolve: the kernel with the small solver bodies; otherwise the handle's body alone
  if (k.flags & (LIPMPC_FLAG_INTERIOR | LIPMPC_FLAG_NO_PRESOLVE | LIPMPC_FLAG_WARM_START))
    hipLaunchKernelGGL((plan_step_kernel<G, NOBS_L, NVAR, false>), dim3(blocks), dim3(WAVE), 0, stream, k, B, state, goal, first_foot,
                       delta, obs_xy, obs_nv, U, X, theta, omega, obj, status, iters, active, working, c_eta, diag, bounds, c_eta_in, sched, overflow_in);
  else
    hipLaunchKernelGGL((plan_step_kernel<G, NOBS_L, NVAR, true>), dim3(blocks), dim3(WAVE), 0, stream, k, B, state, goal, first_foot,
                       delta, obs_xy, obs_nv, U, X, theta, omega, obj, status, iters, active, working, c_eta, diag, bounds, c_eta_in, sched, overflow_in);
}

template void launch_plan_step<INST_G, INST_NL, INST_NV>(const KArgs&, long, const double*, const double*, const int8_t*,
                                                const double*, const double*, const int32_t*, double*, double*,
                                                double*, double*, double*, int32_t*, int32_t*, unsigned long long*, unsigned long long*,
                                                double*, double*, const double*, const double*, int32_t*, const int32_t*, hipStream_t);

template <int G, int NOBS_L, int NVAR>
void launch_rollout(const KArgs& k, long B, int k_max, int mpc_step, double stop_obj, const double* state0,
                    const double* goal, const int8_t* first_foot, const double* delta, const double* obs_xy,
                    const int32_t* obs_nv, double* X_pred, double* U_pred, int32_t* n_steps, int32_t* last_status,
                    int32_t* total_iters, const double* bounds, hipStream_t stream) {
  constexpr int GPW = WAVE / G;
  const unsigned blocks = (unsigned)((B + GPW - 1) / GPW);
  hipLaunchKernelGGL((rollout_kernel<G, NOBS_L, NVAR>), dim3(blocks), dim3(WAVE), 0, stream, k, B, k_max, mpc_step, stop_obj,
                     state0, goal, first_foot, delta, obs_xy, obs_nv, X_pred, U_pred, n_steps, last_status, total_iters, bounds);
}

template void launch_rollout<INST_G, INST_NL, INST_NV>(const KArgs&, long, int, int, double, const double*, const double*,
                                              const int8_t*, const double*, const double*, const int32_t*, double*,
                                              double*, int32_t*, int32_t*, int32_t*, const double*, hipStream_t);
#endif  // INST_LIST

}  // namespace lipmpc_dev
