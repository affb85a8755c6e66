#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// dependent chain of v_fmac_f64_dpp with NOPS wait states between them; results must be identical for a safe NOPS
template <int NOPS> __global__ __launch_bounds__(64) void k(double* out, const double* in) {
  double b = in[threadIdx.x], m = in[64 + threadIdx.x];
#pragma unroll
  for (int u = 0; u < 64; ++u) {
    if constexpr (NOPS == 0) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m), "n"(5));
    if constexpr (NOPS == 1) asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m), "n"(5));
    if constexpr (NOPS == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m), "n"(5));
    if constexpr (NOPS == 3) asm volatile("s_nop 3\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m), "n"(5));
    // a plain VALU write of b right before the DPP read (the classic hazard), different lanes' values
    if constexpr (NOPS == 10) asm volatile("v_add_f64 %0, %0, %1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m), "n"(5));
    if constexpr (NOPS == 12) asm volatile("v_add_f64 %0, %0, %1\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m), "n"(5));
    // 32-bit: v_mov_b32 write then v_mov_b32_dpp read
  }
  out[blockIdx.x * 64 + threadIdx.x] = b;
}
int main() {
  double h[128]; for (int i = 0; i < 64; ++i) { h[i] = 1.0 + 0.01 * i; h[64 + i] = 1e-3 * (1 + (i % 7)); }
  double *in, *out; CHECK(hipMalloc(&in, sizeof(h))); CHECK(hipMalloc(&out, 8 * 64 * 1024 * 8));
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  static double r[8][64 * 1024];
  int modes[6] = {0, 1, 2, 3, 10, 12};
  for (int rep = 0; rep < 3; ++rep) {
  hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, out + 0 * 65536, in);
  hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, out + 1 * 65536, in);
  hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, out + 2 * 65536, in);
  hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, out + 3 * 65536, in);
  hipLaunchKernelGGL(k<10>, dim3(1024), dim3(64), 0, 0, out + 4 * 65536, in);
  hipLaunchKernelGGL(k<12>, dim3(1024), dim3(64), 0, 0, out + 5 * 65536, in);
  CHECK(hipDeviceSynchronize());
  for (int m = 0; m < 6; ++m) CHECK(hipMemcpy(r[m], out + m * 65536, 8 * 65536, hipMemcpyDeviceToHost));
  // CPU reference for modes 0..3 (same chain) and 10/12
  double ref[64], ref2[64];
  for (int g = 0; g < 4; ++g) {
    double b[16], b2[16];
    for (int l = 0; l < 16; ++l) { b[l] = h[16 * g + l]; b2[l] = b[l]; }
    for (int u = 0; u < 64; ++u) {
      double src = b[5]; for (int l = 0; l < 16; ++l) b[l] = __builtin_fma(src, h[64 + 16 * g + l], b[l]);
      for (int l = 0; l < 16; ++l) b2[l] = b2[l] + h[64 + 16 * g + l];
      double s2 = b2[5]; for (int l = 0; l < 16; ++l) b2[l] = __builtin_fma(s2, h[64 + 16 * g + l], b2[l]);
    }
    for (int l = 0; l < 16; ++l) { ref[16 * g + l] = b[l]; ref2[16 * g + l] = b2[l]; }
  }
  for (int m = 0; m < 6; ++m) {
    long bad = 0;
    for (int i = 0; i < 65536; ++i) if (r[m][i] != (m < 4 ? ref : ref2)[i % 64]) ++bad;
    printf("rep %d mode nops=%d: %ld of 65536 lanes differ from the exact chain\n", rep, modes[m], bad);
  }
  }
  return 0;
}
