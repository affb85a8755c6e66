// dev tool: what v_permlane16_swap_b32 does to lane ids (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned *d, h[128]; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) printf("r0 row%d: first lane value %u   r1 row%d: first lane value %u\n", r, h[16 * r], r, h[64 + 16 * r]);
  return 0;
}
