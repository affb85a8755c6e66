// dev tool: relative error of the v_rcp_f64 / v_rsq_f64 seeds and of 1-2 Newton steps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double v = x[i];
  double y0 = __builtin_amdgcn_rcp(v);
  double y1 = fma(fma(-v, y0, 1.0), y0, y0);
  double y2 = fma(fma(-v, y1, 1.0), y1, y1);
  o[3*i] = y0; o[3*i+1] = y1; o[3*i+2] = y2;
}
int main() {
  const int n = 1 << 16; double *hx = new double[n], *ho = new double[3*n], *dx, *dout;
  srand(1); for (int i = 0; i < n; ++i) hx[i] = exp((rand() / (double)RAND_MAX - 0.5) * 60.0) * (1.0 + rand() / (double)RAND_MAX);
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n); hipMemcpy(ho, dout, 3 * n * 8, hipMemcpyDeviceToHost);
  double e[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { double r = fabs(ho[3*i+j] * hx[i] - 1.0); if (r > e[j]) e[j] = r; }
  printf("max rel err: rcp seed %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e[0], e[1], e[2]);
  return 0;
}
