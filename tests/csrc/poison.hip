// poison.hip -- fills the register files and the LDS of every CU with a bit pattern (registers and LDS keep whatever
// the previous kernel left there: a kernel that reads a register / LDS word / spilled SGPR lane it never wrote sees
// THAT, which differs from box to box).  Used by tests to run the solver kernels after an all-NaN and after an
// all-zero poisoning: results must not change.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tests/csrc/libpoison.so tests/csrc/poison.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void poison_kernel(uint32_t pat, uint32_t* sink, int what) {
  extern __shared__ uint32_t lds[];
  if (what & 1) for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 64) lds[i] = pat;          // 40 KB per wave, 4 waves per CU
  __syncthreads();
  // all 512 vector registers of the lane (256 VGPR + 256 AGPR) and the scalar file
  if (what & 2) asm volatile(
      ".altmacro\n"
      ".macro fillv n\n v_mov_b32 v\\n, %0\n.endm\n"
      ".set i, 8\n .rept 248\n fillv %%i\n .set i, i+1\n .endr\n"
      :: "v"(pat) : "memory", "v254", "v255");
  if (what & 4) asm volatile(
      ".altmacro\n"
      ".macro filla n\n v_accvgpr_write_b32 a\\n, %0\n.endm\n"
      ".set i, 0\n .rept 256\n filla %%i\n .set i, i+1\n .endr\n"
      :: "v"(pat) : "memory", "v254", "v255", "a255");      // the clobbers make the kernel own all 512 registers
  if (what & 8) asm volatile(
      ".altmacro\n"
      ".macro fills n\n s_mov_b32 s\\n, %0\n.endm\n"
      ".set i, 20\n .rept 80\n fills %%i\n .set i, i+1\n .endr\n"
      :: "s"(pat) : "memory", "s99");
  if (sink && pat == 0x12345678u && threadIdx.x == 63) sink[blockIdx.x] = lds[threadIdx.x];
}

extern "C" int lipmpc_poison(uint32_t pattern, int what) {
  (void)hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  // 8 waves per SIMD cannot be resident with 512 registers each: one wave per SIMD, several rounds
  (void)hipGetLastError();
  for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(poison_kernel, dim3(256 * 4 * 2), dim3(64), 40 * 1024, 0, pattern, (uint32_t*)nullptr, what);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
