// dev tool: the G=32 exchange helpers against __shfl on random data (all lanes active, and with one group masked off)
#include "../humanoid-navigation-using-mpc-ldcbf_amd/csrc/lipmpc_kernel.hpp"
#include <stdio.h>
using namespace lipmpc_dev;
template <int D> __device__ int chk_updown(double x, int lane, int* err) {
  const double a = gup<32, D>(x, lane), b = gdown<32, D>(x, lane);
  double ra = __shfl_up(x, D, 32); if (lane < D) ra = 0.0;
  double rb = __shfl_down(x, D, 32); if (lane + D >= 32) rb = 0.0;
  return (a != ra) + 2 * (b != rb);
}
__global__ void k(const double* in, int* err, int mask_group) {
  const int lane = threadIdx.x & 31;
  const double x = in[threadIdx.x];
  int e = 0;
  if (mask_group < 0 || (threadIdx.x >> 5) != mask_group) {
    e |= chk_updown<2>(x, lane, err) ? 1 : 0;
    e |= chk_updown<4>(x, lane, err) ? 2 : 0;
    e |= chk_updown<8>(x, lane, err) ? 4 : 0;
    e |= chk_updown<16>(x, lane, err) ? 8 : 0;
    e |= (gxor<32, 16>(x) != __shfl_xor(x, 16, 32)) ? 16 : 0;
    e |= (gxor<32, 4>(x) != __shfl_xor(x, 4, 32)) ? 32 : 0;
    e |= (gxor<32, 8>(x) != __shfl_xor(x, 8, 32)) ? 64 : 0;
    e |= (gbcast<32, 5>(x) != __shfl(x, 5, 32)) ? 128 : 0;
    e |= (gbcast<32, 21>(x) != __shfl(x, 21, 32)) ? 256 : 0;
    int i = (int)(x * 1000); double v = x; int i2 = i; double v2 = x;
    gargmin<32>(v, i);
    for (int m = 1; m < 32; m <<= 1) { double ov = __shfl_xor(v2, m, 32); int oi = __shfl_xor(i2, m, 32); bool t = (ov < v2) || (ov == v2 && oi < i2); v2 = t ? ov : v2; i2 = t ? oi : i2; }
    e |= (v != v2 || i != i2) ? 512 : 0;
    const int ti = (int)threadIdx.x * 7 + 3;       // the 32-bit path of the cross-row swap (argmin indices)
    e |= (gxor<32, 16>(ti) != __shfl_xor(ti, 16, 32)) ? 1024 : 0;
  }
  err[threadIdx.x] = e;
}
int main() {
  double h[64]; for (int i = 0; i < 64; ++i) h[i] = (double)((i * 37) % 64) + 0.25;
  double* d; int* e; int he[64];
  hipMalloc(&d, 512); hipMalloc(&e, 256); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  for (int mg = -1; mg < 2; ++mg) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, mg); hipMemcpy(he, e, 256, hipMemcpyDeviceToHost);
    int all = 0; for (int i = 0; i < 64; ++i) all |= he[i];
    printf("mask_group %d: error bits 0x%x\n", mg, all);
  }
  return 0;
}
