// fp64_peak.hip -- measured FP64 vector FMA peak of the device (replaces the datasheet figure in bench.py's roofline)
// and the latencies the step kernel's substitution chains are made of.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/fp64_peak tools/fp64_peak.hip      run: tools/fp64_peak [json-out]
// 1. peak: every SIMD runs W waves (W = 1, 2, 4) of 8 independent v_fma_f64 chains -> TFLOP/s (2 flop per lane-FMA)
// 2. dependent v_fma_f64 chain: cycles per instruction (s_memtime), one wave per SIMD
// 3. the old substitution step (v_mul_f64, v_mov_b64_dpp row_newbcast, v_cndmask, v_fma_f64) and the fused one
//    (s_nop 1 + v_fmac_f64_dpp row_newbcast) as dependent chains: cycles per step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void peak_kernel(double* out, int iters, double a, double b) {
  double c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c0 = fma(c0, a, b); c1 = fma(c1, a, b); c2 = fma(c2, a, b); c3 = fma(c3, a, b);
      c4 = fma(c4, a, b); c5 = fma(c5, a, b); c6 = fma(c6, a, b); c7 = fma(c7, a, b);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

__global__ __launch_bounds__(64) void dep_fma_kernel(double* out, long long* cyc, int iters, double a, double b) {
  double c = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) c = fma(c, a, b);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = c;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__device__ __forceinline__ double zero_unless(bool c, double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned hi = c ? (unsigned)(u >> 32) : 0u;
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned)u);
}
template <int J> __device__ __forceinline__ double old_step(double b, double ipiv, double kj, int ln) {
  const double wj = zero_unless(ln > J, __builtin_amdgcn_mov_dpp(b * ipiv, 0x150 + J, 0xf, 0xf, false));
  return fma(-kj, wj, b);
}
template <int J> __device__ __forceinline__ double new_step(double b, double xj) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(xj), "n"(J));
  return b;
}
template <int MODE> __global__ __launch_bounds__(64) void chain_kernel(double* out, long long* cyc, int iters, double k, double ipiv) {
  double b = 1.0 + 1e-3 * threadIdx.x;
  const int ln = threadIdx.x & 15;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {
      b = old_step<0>(b, ipiv, k, ln); b = old_step<1>(b, ipiv, k, ln); b = old_step<2>(b, ipiv, k, ln); b = old_step<3>(b, ipiv, k, ln);
      b = old_step<4>(b, ipiv, k, ln); b = old_step<5>(b, ipiv, k, ln); b = old_step<6>(b, ipiv, k, ln); b = old_step<7>(b, ipiv, k, ln);
      b = old_step<8>(b, ipiv, k, ln); b = old_step<9>(b, ipiv, k, ln); b = old_step<10>(b, ipiv, k, ln); b = old_step<11>(b, ipiv, k, ln);
      b = old_step<12>(b, ipiv, k, ln); b = old_step<13>(b, ipiv, k, ln); b = old_step<14>(b, ipiv, k, ln); b = old_step<15>(b, ipiv, k, ln);
    } else {
      b = new_step<0>(b, k); b = new_step<1>(b, k); b = new_step<2>(b, k); b = new_step<3>(b, k);
      b = new_step<4>(b, k); b = new_step<5>(b, k); b = new_step<6>(b, k); b = new_step<7>(b, k);
      b = new_step<8>(b, k); b = new_step<9>(b, k); b = new_step<10>(b, k); b = new_step<11>(b, k);
      b = new_step<12>(b, k); b = new_step<13>(b, k); b = new_step<14>(b, k); b = new_step<15>(b, k);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = b;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* out; long long* cyc;
  const int max_blocks = cus * 4 * 4;
  CHECK(hipMalloc(&out, sizeof(double) * 64 * max_blocks));
  CHECK(hipMalloc(&cyc, sizeof(long long) * max_blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double best = 0.0, tf_w[5] = {0, 0, 0, 0, 0};
  for (int w : {1, 2, 4}) {
    const int blocks = cus * 4 * w, iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(peak_kernel, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999999, 1e-7);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double tf = (double)blocks * 64 * iters * 16 * 8 * 2 / (ms * 1e-3) / 1e12;
      tf_w[w] = std::max(tf_w[w], tf);
    }
    best = std::max(best, tf_w[w]);
  }
  std::vector<long long> h(cus * 4);
  auto run_cycles = [&](auto launch, int iters, int per_iter) {
    launch(iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * cus * 4, hipMemcpyDeviceToHost));
    return median(h) / ((double)iters * per_iter);
  };
  const int blocks = cus * 4;
  const double dep = run_cycles([&](int it) { hipLaunchKernelGGL(dep_fma_kernel, dim3(blocks), dim3(64), 0, 0, out, cyc, it, 0.999999, 1e-7); }, 2000, 64);
  const double oldc = run_cycles([&](int it) { hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, it, 1e-3, 0.5); }, 2000, 16);
  const double newc = run_cycles([&](int it) { hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, it, -1e-3, 0.5); }, 2000, 16);
  char buf[1024];
  snprintf(buf, sizeof(buf),
           "{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"fp64_fma_tflops\": {\"1_wave_per_simd\": %.2f, \"2\": %.2f, \"4\": %.2f}, "
           "\"fp64_peak_tflops_measured\": %.2f, \"dependent_v_fma_f64_cycles\": %.2f, "
           "\"substitution_step_cycles\": {\"mul+mov_dpp+cndmask+fma\": %.2f, \"s_nop1+v_fmac_f64_dpp\": %.2f}}",
           prop.name, cus, prop.clockRate / 1000, tf_w[1], tf_w[2], tf_w[4], best, dep, oldc, newc);
  printf("%s\n", buf);
  if (argc > 1) { FILE* f = fopen(argv[1], "w"); if (f) { fprintf(f, "%s\n", buf); fclose(f); } }
  return 0;
}
