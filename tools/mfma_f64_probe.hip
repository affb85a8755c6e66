// mfma_f64_probe.hip -- what v_mfma_f64_4x4x4_4b_f64 does on gfx950, measured: operand / result lane maps (exact integer
// data, asymmetric operands), the latency of a dependent chain and the issue rate of independent ones.
// Four independent 4x4x4 products per instruction, one per 16-lane group of the wave = one per problem of the step
// kernel's layout (lipmpc_kernel.hpp: one QP per DPP row).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_probe tools/mfma_f64_probe.hip && tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

__global__ void layout_kernel(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

template <int DEP>
__global__ void timing_kernel(double* out, long long* cycles, int reps) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  double c0 = l, c1 = l + 1, c2 = l + 2, c3 = l + 3, c4 = l + 4, c5 = l + 5, c6 = l + 6, c7 = l + 7;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if (DEP) {          // dependent chain: the result is the next C operand
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    } else {            // eight independent accumulators
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + l] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

// dependent chain through the A operand (result -> next A): what a factorisation's panel / trailing updates look like
__global__ void timing_dep_a(double* out, long long* cycles, int reps) {
  const int l = threadIdx.x;
  double a = 1e-3 * l, b = 1.0 - 1e-9 * l, c = 0.0;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + l] = a;
  if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

// v_fmac_f64 dependent chain for scale (same counter)
__global__ void timing_fma(double* out, long long* cycles, int reps) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-9 * l, x = l;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x = fma(x, a, 1e-9);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + l] = x;
  if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
  double ha[64], hb[64], hc[64], hd[64];
  double *a, *b, *c, *d;
  CK(hipMalloc(&a, 512)); CK(hipMalloc(&b, 512)); CK(hipMalloc(&c, 512)); CK(hipMalloc(&d, 512));
  // ---- layout: one-hot probes.  A = e_p (1 at lane p of block 0), B = all lanes k-coded -> which B lanes pair with A lane p,
  // and where the product lands.
  // Step 1: A one-hot at lane p, B[lane] = 1 + lane (distinct), C = 0: D shows a single row of non-zeros: D[lane] = B[some lane].
  printf("A one-hot at lane p (block 0), B[lane] = 100 + lane, C = 0: non-zero D lanes -> value\n");
  for (int p = 0; p < 16; ++p) {
    for (int l = 0; l < 64; ++l) { ha[l] = (l == p) ? 1.0 : 0.0; hb[l] = 100.0 + l; hc[l] = 0.0; }
    CK(hipMemcpy(a, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(c, hc, 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, c, d);
    CK(hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost));
    printf(" p=%2d:", p);
    for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" D[%d]=%.0f", l, hd[l]);
    printf("\n");
  }
  // a full check of the inferred map is printed by the LDL tool; here also the cross-block isolation:
  for (int l = 0; l < 64; ++l) { ha[l] = (l == 16 + 5) ? 1.0 : 0.0; hb[l] = 100.0 + l; hc[l] = 0.0; }
  CK(hipMemcpy(a, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(c, hc, 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, c, d);
  CK(hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost));
  printf("A one-hot at lane 21 (block 1):");
  for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" D[%d]=%.0f", l, hd[l]);
  printf("\n");

  // ---- timing ----
  double* out; long long* cyc;
  CK(hipMalloc(&out, 64 * 8 * 1024)); CK(hipMalloc(&cyc, 8 * 1024));
  const int reps = 2000;
  long long hcyc[4];
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(timing_kernel<1>, dim3(1), dim3(64), 0, 0, out, cyc, reps);
    CK(hipMemcpy(hcyc, cyc, 8, hipMemcpyDeviceToHost));
    if (pass) printf("dependent (C) chain   : %.1f counter ticks per v_mfma_f64_4x4x4\n", (double)hcyc[0] / (8.0 * reps));
    hipLaunchKernelGGL(timing_kernel<0>, dim3(1), dim3(64), 0, 0, out, cyc, reps);
    CK(hipMemcpy(hcyc, cyc, 8, hipMemcpyDeviceToHost));
    if (pass) printf("independent x8        : %.1f counter ticks per v_mfma_f64_4x4x4\n", (double)hcyc[0] / (8.0 * reps));
    hipLaunchKernelGGL(timing_dep_a, dim3(1), dim3(64), 0, 0, out, cyc, reps);
    CK(hipMemcpy(hcyc, cyc, 8, hipMemcpyDeviceToHost));
    if (pass) printf("dependent (A) chain   : %.1f counter ticks per v_mfma_f64_4x4x4\n", (double)hcyc[0] / (8.0 * reps));
    hipLaunchKernelGGL(timing_fma, dim3(1), dim3(64), 0, 0, out, cyc, reps);
    CK(hipMemcpy(hcyc, cyc, 8, hipMemcpyDeviceToHost));
    if (pass) printf("dependent v_fma_f64   : %.1f counter ticks per instruction (4 shader cycles each: the tick/cycle scale)\n", (double)hcyc[0] / (8.0 * reps));
  }
  return 0;
}
