// ldl_mfma.hip -- the KKT factorisation + two solves of one interior-point iteration (16 x 16, four problems per
// wavefront), timed in the two forms north_star allows for it:
//
//  (a) DPP form (what the step kernel runs, csrc/lipmpc_kernel.hpp: factor() / solve(), FUSED path): one row per lane, one
//      problem per 16-lane DPP row, 120 v_fmac_f64_dpp row_newbcast elimination steps + 2 x 30 substitution links; the four
//      problems of a wave advance in lockstep (every instruction works on all four).
//  (b) MFMA form.  Measured first (tools/mfma_f64_probe.hip): v_mfma_f64_4x4x4_4b_f64 takes A at lane 16k + 4blk + i,
//      B at lane 16k + 4blk + j and returns D[i][j] at lane 16i + 4blk + j -- the contraction index k is the DPP ROW, a block
//      is the quad column {4blk..4blk+3} of all four rows.  In the step kernel's layout (problem = DPP row) one instruction
//      therefore SUMS OVER THE FOUR PROBLEMS; per-problem use needs the problem spread over the four DPP rows, i.e. K_g as the
//      accumulator of v_mfma_f64_16x16x4_f64 (col = lane & 15, row = (lane >> 4) + 4 reg): then rows 4J..4J+3 of K_g sit in
//      register J on DPP rows 0..3 -- exactly the B operand (and, by symmetry, the A operand) of the rank-4 trailing update
//      K -= P^T (K_JJ^-1 P), P = that register: a blocked right-looking LDL^T with NO data movement for the update itself.
//      What it costs per problem and block column J = 0..3: eliminate the 4 x 4 pivot block (4 dependent rank-1 steps,
//      v_mfma_f64_4x4x4 with masked operands), W = K_JJ^-1 P (one 4x4x4), trailing update (one 16x16x4): 6 MFMAs, 24 per
//      problem, 96 per wave -- the four problems do NOT share instructions, they only interleave (independent chains).
//      A solve is 4 block steps x (pivot-block solve + off-diagonal update) per direction: 16 dependent MFMAs per solve and
//      problem, 128 per wave for the two solves.
//      The kernel below issues exactly that instruction stream with the true dependences (each product consumes the
//      previous result as an operand) and the four problems interleaved; the glue a real implementation needs on top
//      (pivot reciprocals, operand masks, moving the pivot block into the A layout) is LEFT OUT: a lower bound for (b).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldl_mfma tools/ldl_mfma.hip && tools/ldl_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

template <int I, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < E) { f(std::integral_constant<int, I>{}); static_for<I + 1, E>(f); }
}
__device__ __forceinline__ int fresh(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double zero_unless(bool c, double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned hi = c ? (unsigned)(u >> 32) : 0u;
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned)u);
}
__device__ __forceinline__ double fast_rcp(double x) { double y = __builtin_amdgcn_rcp(x); return fma(fma(-x, y, 1.0), y, y); }
template <int J> __device__ __forceinline__ double bcast16(double x) { return __builtin_amdgcn_mov_dpp(x, 0x150 + J, 0xf, 0xf, false); }
__device__ __forceinline__ void dpp_fence() { asm volatile("s_nop 4"); }
#include "../humanoid-navigation-using-mpc-ldcbf_amd/csrc/lipmpc_fused_steps.inc"

// (a) the step kernel's factorisation and solve, verbatim in structure (lipmpc_kernel.hpp, FUSED branch)
__global__ __launch_bounds__(64) void dpp_form(const double* __restrict__ Kin, const double* __restrict__ bin, double* __restrict__ xout,
                                               long long* cycles, int reps) {
  constexpr int NV = 16;
  const int tid = threadIdx.x, lane = tid & 15;
  double K0[NV], b0 = bin[tid];
  for (int c = 0; c < NV; ++c) K0[c] = Kin[(tid >> 4) * 256 + lane * 16 + c];
  double x1 = 0.0, x2 = 0.0;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    double Krow[NV], Xl[NV], Yu[NV], ipiv = 0.5;
#pragma unroll
    for (int c = 0; c < NV; ++c) Krow[c] = K0[c] + 1e-12 * x2;            // (depends on the previous repetition)
    const int ln = fresh(lane);
    dpp_fence();
    static_for<0, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double pj = bcast16<j>(Krow[j]);
      const double ip = fast_rcp(pj);
      const double nf = zero_unless(ln > j, Krow[j] * -ip);
      ipiv = (ln == j) ? ip : ipiv;
      Xl[j] = nf;
      FactorStep<NV, j>::run(Krow, nf);
    });
    const double nip = -ipiv;
    static_for<1, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      Yu[j] = zero_unless(fresh(lane) < j, Krow[j] * nip);
    });
    dpp_fence();
    double b = solve_forward_chain(b0, Xl);
    x1 = solve_backward_chain(b * ipiv, Yu);
    dpp_fence();
    b = solve_forward_chain(b0 + x1, Xl);                                   // the corrector's right-hand side needs the predictor's result
    x2 = solve_backward_chain(b * ipiv, Yu);
  }
  const long long t1 = __builtin_readcyclecounter();
  xout[blockIdx.x * 64 + tid] = x1;
  xout[gridDim.x * 64 + blockIdx.x * 64 + tid] = x2;
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

typedef double d4 __attribute__((ext_vector_type(4)));
// (b) the MFMA instruction stream of the blocked form, true dependences, four problems interleaved, no glue
__global__ __launch_bounds__(64) void mfma_form(double* __restrict__ out, long long* cycles, int reps, int with_solves) {
  const int tid = threadIdx.x;
  d4 K[4];                       // K_g as the 16x16x4 accumulator, one per problem
  for (int g = 0; g < 4; ++g) for (int m = 0; m < 4; ++m) K[g][m] = ((tid & 15) == ((tid >> 4) + 4 * m) ? 4.0 : 0.01) + 1e-3 * g;
  double v[4] = {1.0, 2.0, 3.0, 4.0};
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      double p[4], w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) p[g] = K[g][J];                          // register J = rows 4J..4J+3: the panel
      // pivot block: 4 dependent rank-1 eliminations (4x4x4), interleaved over the problems
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) p[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(p[g], p[g], p[g], 0, 0, 0);
      // W = K_JJ^-1 P
#pragma unroll
      for (int g = 0; g < 4; ++g) w[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(p[g], K[g][J], 0.0, 0, 0, 0);
      // trailing update K -= P^T W (rank 4, the whole 16 x 16 accumulator)
#pragma unroll
      for (int g = 0; g < 4; ++g) K[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(K[g][J], w[g], K[g], 0, 0, 0);
    }
    if (with_solves) {
      // two solves, each: forward 4 block steps (pivot-block solve 4x4x4, then off-diagonal update 16x16x4 with the vector in
      // one k slot) and the same backwards; the second solve's right-hand side depends on the first's result
#pragma unroll
      for (int sv = 0; sv < 2; ++sv)
#pragma unroll
        for (int dir = 0; dir < 2; ++dir)
#pragma unroll
          for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              v[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(K[g][J], v[g], 0.0, 0, 0, 0);
              d4 acc = {v[g], 0.0, 0.0, 0.0};
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(K[g][(J + 1) & 3], v[g], acc, 0, 0, 0);
              v[g] = acc[0];
            }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) K[g][0] = K[g][0] * 1e-30 + 4.0 + 1e-3 * v[g] * 1e-30;      // keep the values bounded, keep the dependence
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int g = 0; g < 4; ++g) s += K[g][0] + K[g][1] + K[g][2] + K[g][3] + v[g];
  out[blockIdx.x * 64 + tid] = s;
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
  // four SPD matrices (2I + G^T D G-like), right-hand sides; correctness of (a) against a host LDL^T
  double hK[4 * 256], hb[64], hx[2 * 64 * 1024];
  srand(1);
  for (int g = 0; g < 4; ++g) {
    double M[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) M[i][j] = (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = (i == j) ? 2.0 : 0.0;
      for (int k = 0; k < 16; ++k) s += M[k][i] * M[k][j] * (1.0 + 100.0 * (k & 3));
      hK[g * 256 + i * 16 + j] = s;
    }
  }
  for (int i = 0; i < 64; ++i) hb[i] = rand() / (double)RAND_MAX - 0.5;
  double *K, *b, *x; long long* cyc;
  const int blocks = 1024;
  CK(hipMalloc(&K, sizeof(hK))); CK(hipMalloc(&b, sizeof(hb))); CK(hipMalloc(&x, 2 * 64 * blocks * 8)); CK(hipMalloc(&cyc, blocks * 8));
  CK(hipMemcpy(K, hK, sizeof(hK), hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dpp_form, dim3(1), dim3(64), 0, 0, K, b, x, cyc, 1);
  CK(hipMemcpy(hx, x, 2 * 64 * 8, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int g = 0; g < 4; ++g) for (int i = 0; i < 16; ++i) {      // residual K x1 - b
    double s = -hb[g * 16 + i];
    for (int j = 0; j < 16; ++j) s += hK[g * 256 + i * 16 + j] * hx[g * 16 + j];
    worst = fmax(worst, fabs(s));
  }
  printf("(a) DPP form: max |K x - b| over the four problems = %.2e\n", worst);
  const int reps = 500;
  long long* hc = (long long*)malloc(blocks * 8);
  for (int grid : {1, blocks}) {
    for (int pass = 0; pass < 2; ++pass) {
      hipLaunchKernelGGL(dpp_form, dim3(grid), dim3(64), 0, 0, K, b, x, cyc, reps);
      CK(hipMemcpy(hc, cyc, grid * 8, hipMemcpyDeviceToHost));
      double m = 0; for (int i = 0; i < grid; ++i) m += hc[i]; m /= grid;
      if (pass) printf("(a) DPP form,  %4d wave(s): %.0f cycles per (factorisation + two solves) of a wave = 4 problems\n", grid, m / reps);
      for (int ws = 0; ws < 2; ++ws) {
        hipLaunchKernelGGL(mfma_form, dim3(grid), dim3(64), 0, 0, x, cyc, reps, ws);
        CK(hipMemcpy(hc, cyc, grid * 8, hipMemcpyDeviceToHost));
        m = 0; for (int i = 0; i < grid; ++i) m += hc[i]; m /= grid;
        if (pass) printf("(b) MFMA form, %4d wave(s): %.0f cycles per %s of a wave = 4 problems (MFMA stream only, no glue: a lower bound)\n",
                         grid, m / reps, ws ? "(factorisation + two solves)" : "factorisation alone");
      }
    }
  }
  // (a) without the solves, for the split
  return 0;
}
