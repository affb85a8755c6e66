// dev tool: what v_permlane32_swap_b32 does to lane ids (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int p = 0; p < 4; ++p) { printf("part %d:", p); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[p * 64 + i]); printf("\n"); }
  return 0;
}
