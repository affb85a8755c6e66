#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// each MODE: body executed 16x per loop iteration
template <int MODE> __global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters, double kk) {
  double b = 1.0 + 1e-3 * threadIdx.x, c = b + 1, d = b + 2, e = b + 3, f = b + 4, g = b + 5, h = b + 6, i2 = b + 7;
  double m = kk;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (MODE == 0) {  // dependent fmac_dpp, no nop (timing only)
        asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m));
      } else if constexpr (MODE == 1) {  // dependent with s_nop 0
        asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m));
      } else if constexpr (MODE == 2) {  // dependent with s_nop 1
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(m));
      } else if constexpr (MODE == 3) {  // 8 independent fmac_dpp (throughput)
        asm volatile("v_fmac_f64_dpp %0, %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %5, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %6, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %7, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2) : "v"(m));
      } else if constexpr (MODE == 4) {  // 8 independent plain fmac
        asm volatile("v_fmac_f64 %0, %0, %8\n\tv_fmac_f64 %1, %1, %8\n\tv_fmac_f64 %2, %2, %8\n\tv_fmac_f64 %3, %3, %8\n\t"
                     "v_fmac_f64 %4, %4, %8\n\tv_fmac_f64 %5, %5, %8\n\tv_fmac_f64 %6, %6, %8\n\tv_fmac_f64 %7, %7, %8"
                     : "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2) : "v"(m));
      } else if constexpr (MODE == 5) {  // 8 independent v_mov_b64_dpp
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %5, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %7, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2));
      } else if constexpr (MODE == 6) {  // dependent: v_mov_b64_dpp then plain fma (compiler inserts hazards)
        double t = __builtin_amdgcn_mov_dpp(b, 0x153, 0xf, 0xf, false);
        b = fma(t, m, b);
      } else if constexpr (MODE == 7) {  // dependent 32-bit dpp pair (quad_perm) + add: exchange pattern
        double t = __builtin_amdgcn_mov_dpp(b, 0xB1, 0xf, 0xf, true);
        b = b + t * m;
      } else if constexpr (MODE == 8) {  // 8 independent v_cndmask_b32 ... (32-bit op issue)
        asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\t"
                     "v_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %0"
                     : "+v"(((int*)&b)[0]), "+v"(((int*)&c)[0]), "+v"(((int*)&d)[0]), "+v"(((int*)&e)[0]), "+v"(((int*)&f)[0]), "+v"(((int*)&g)[0]), "+v"(((int*)&h)[0]), "+v"(((int*)&i2)[0]));
      } else if constexpr (MODE == 9) {  // accvgpr write+read pairs, 4 each
        asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\t"
                     "v_accvgpr_read_b32 %4, a0\n\tv_accvgpr_read_b32 %5, a1\n\tv_accvgpr_read_b32 %6, a2\n\tv_accvgpr_read_b32 %7, a3"
                     : "+v"(((int*)&b)[0]), "+v"(((int*)&c)[0]), "+v"(((int*)&d)[0]), "+v"(((int*)&e)[0]), "+v"(((int*)&f)[0]), "+v"(((int*)&g)[0]), "+v"(((int*)&h)[0]), "+v"(((int*)&i2)[0]) :: "a0","a1","a2","a3");
      } else if constexpr (MODE == 10) { // dependent rcp f64
        b = __builtin_amdgcn_rcp(b) + m;
      } else if constexpr (MODE == 11) { // 8 independent v_max_f64
        asm volatile("v_max_f64 %0, %0, %8\n\tv_max_f64 %1, %1, %8\n\tv_max_f64 %2, %2, %8\n\tv_max_f64 %3, %3, %8\n\t"
                     "v_max_f64 %4, %4, %8\n\tv_max_f64 %5, %5, %8\n\tv_max_f64 %6, %6, %8\n\tv_max_f64 %7, %7, %8"
                     : "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2) : "v"(m));
      } else if constexpr (MODE == 13) { // 8 independent v_rcp_f64
        asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t"
                     "v_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7"
                     : "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2));
      } else if constexpr (MODE == 14) { // dependent plain fma
        b = fma(b, m, m);
      } else if constexpr (MODE == 15) { // 8 independent v_permlane16_swap_b32
        asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                     "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                     : "+v"(((int*)&b)[0]), "+v"(((int*)&c)[0]), "+v"(((int*)&d)[0]), "+v"(((int*)&e)[0]), "+v"(((int*)&f)[0]), "+v"(((int*)&g)[0]), "+v"(((int*)&h)[0]), "+v"(((int*)&i2)[0]));
      } else if constexpr (MODE == 16) { // dependent IEEE division
        b = m / b + 1.0;
      } else if constexpr (MODE == 17) { // 8 independent 64-bit selects (2 v_cndmask_b32 each)
        b = (threadIdx.x & 1) ? b : c; c = (threadIdx.x & 2) ? c : d; d = (threadIdx.x & 4) ? d : e; e = (threadIdx.x & 8) ? e : f;
        f = (threadIdx.x & 16) ? f : g; g = (threadIdx.x & 32) ? g : h; h = (threadIdx.x & 1) ? h : i2; i2 = (threadIdx.x & 2) ? i2 : b;
      } else if constexpr (MODE == 12) { // ds_read_b64 lane-private, 8 independent then wait
        __shared__ double sm[64 * 8];
        sm[threadIdx.x] = b;
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                     "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h), "=v"(i2) : "v"((unsigned)(threadIdx.x * 8)) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = b + c + d + e + f + g + h + i2;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* cyc; const int blocks = 1024;
  CHECK(hipMalloc(&out, 8 * 64 * blocks)); CHECK(hipMalloc(&cyc, 8 * blocks));
  std::vector<long long> h(blocks);
  auto rep = [&](const char* name, auto launch, int per) {
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-45s %.2f cycles per op\n", name, (double)h[blocks / 2] / (2000.0 * 16 * per));
  };
#define R(M, name, per) rep(name, [&] { hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(64), 0, 0, out, cyc, 2000, 1e-3); }, per)
  R(0, "dep fmac_dpp no nop", 1); R(1, "dep s_nop0+fmac_dpp", 1); R(2, "dep s_nop1+fmac_dpp", 1);
  R(3, "indep fmac_dpp x8", 8); R(4, "indep fmac x8", 8); R(5, "indep mov_b64_dpp x8", 8);
  R(6, "dep mov_b64_dpp + fma (per pair)", 1); R(7, "dep quad_perm pair + mul/add (per group)", 1);
  R(8, "indep v_mov_b32 x8", 8); R(9, "accvgpr write x4 + read x4 (per op)", 8); R(10, "dep rcp+add (per pair)", 1);
  R(11, "indep v_max_f64 x8", 8); R(12, "ds_read_b64 x8 + wait (per read)", 8);
  R(13, "indep v_rcp_f64 x8", 8); R(14, "dep v_fma_f64", 1); R(15, "indep v_permlane16_swap_b32 x8", 8); R(16, "dep IEEE f64 division + add (per pair)", 1);
  R(17, "64-bit selects x8 (per select)", 8);
  return 0;
}
