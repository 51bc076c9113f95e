// Probe: issue cost of the split's VALU instructions on gfx950, alone and in the shadow of MFMAs.
// One workgroup of 256 threads per CU x OCC; each wave runs ITER iterations of
//   [ 8 x ( MFMA 32x32x16 f16 ; NV x op ) ]   (MFMAs on 4 independent accumulators, ops on 8 independent registers)
// and reports shader cycles per iteration (s_memtime).  NV = 0 gives the matrix-pipe floor (8 x 32).
// Reading the numbers: hipcc puts an `s_nop 0` (4 cycles) after every single-instruction asm statement,
// so "alone" shows issue cost + 4 (7.9 = a 4-cycle op); beside MFMAs NV = 4 ops (+4 nops) fill the
// 32-cycle gap exactly, NV = 6 and 8 overflow it by their full cost: the per-gap budget of
// MI355X_MICROARCH.md (MFMA holds vector issue for 8 of its 32 cycles; ~24 cycles of fillers hide).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { OP_MUL, OP_SUB, OP_CVTPK, OP_CVTBACK, OP_CVTBACK_SDWA, OP_PKFMA, OP_AND, OP_MOV, OP_LSHLADD64, OP_FMA, OP_CNDMASK, OP_RSQ, OP_RCP, OP_PKMUL, N_OPS };
static const char* kNames[] = {"v_mul_f32", "v_sub_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa",
                               "v_pk_fma_f32", "v_and_b32", "v_mov_b32", "v_lshl_add_u64", "v_fma_f32", "v_cndmask_b32",
                               "v_rsq_f32", "v_rcp_f32", "v_pk_mul_f32"};

template <int OP>
__device__ __forceinline__ void op1(float& d, float s, double& d64) {
  if constexpr (OP == OP_MUL) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_SUB) asm volatile("v_sub_f32 %0, %1, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_CVTBACK) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_CVTBACK_SDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(d64) : "v"(d64));
  else if constexpr (OP == OP_AND) asm volatile("v_and_b32 %0, %1, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(d64) : "v"(d64));
  else if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %1, vcc" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_RSQ) asm volatile("v_rsq_f32 %0, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %1" : "=v"(d) : "v"(s));
  else if constexpr (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(d64) : "v"(d64));
}

template <int OP, int NV, bool MFMA>
__global__ __launch_bounds__(256) void probe(long long* out, int iters, float seed) {
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed; c2[r] = seed; c3[r] = seed; }
  f16x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (_Float16)seed; b[r] = (_Float16)(seed + r); }
  float d[8]; double d64[8];
  for (int j = 0; j < 8; ++j) { d[j] = seed + j; d64[j] = seed + j; }
  float s = seed + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if constexpr (MFMA) {
        f32x16& c = (g & 3) == 0 ? c0 : (g & 3) == 1 ? c1 : (g & 3) == 2 ? c2 : c3;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) op1<OP>(d[j & 7], s, d64[j & 7]);
    }
  }
  long long t1 = clock64();
  float acc = 0;
  for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
  for (int j = 0; j < 8; ++j) acc += d[j] + (float)d64[j];
  if (acc == 123.456f) out[1000] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int OP, int NV, bool MFMA>
double run(int occ) {
  static long long* d = nullptr;
  if (!d) (void)hipMalloc(&d, 16384);
  const int iters = 1000;
  probe<OP, NV, MFMA><<<256 * occ, 256>>>(d, iters, 1.0f);
  probe<OP, NV, MFMA><<<256 * occ, 256>>>(d, iters, 1.0f);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  return (double)h / iters;
}

template <int OP>
void row() {
  // alone: 8 ops per group x 8 groups = 64 ops per iteration
  const double a1 = run<OP, 8, false>(1) / 64, a2 = run<OP, 8, false>(2) / 64;
  const double m4_1 = run<OP, 4, true>(1), m6_1 = run<OP, 6, true>(1), m8_1 = run<OP, 8, true>(1);
  const double m4_2 = run<OP, 4, true>(2), m6_2 = run<OP, 6, true>(2), m8_2 = run<OP, 8, true>(2);
  printf("%-20s alone: %4.1f cyc/op (1 wave/SIMD) %4.1f (2)   | beside 8 MFMA (floor 256 / 512): 1 wave/SIMD NV=4: %5.0f  6: %5.0f  8: %5.0f   2 waves/SIMD NV=4: %5.0f  6: %5.0f  8: %5.0f\n",
         kNames[OP], a1, a2, m4_1, m6_1, m8_1, m4_2, m6_2, m8_2);
}
int main() {
  printf("floor: MFMA only 1 wave/SIMD %.0f, 2 waves/SIMD %.0f cycles per 8 MFMAs\n", run<OP_MUL, 0, true>(1), run<OP_MUL, 0, true>(2));
  row<OP_MUL>(); row<OP_SUB>(); row<OP_FMA>(); row<OP_CVTPK>(); row<OP_CVTBACK>(); row<OP_CVTBACK_SDWA>(); row<OP_PKFMA>();
  row<OP_AND>(); row<OP_MOV>(); row<OP_LSHLADD64>(); row<OP_CNDMASK>();
  row<OP_RSQ>(); row<OP_RCP>(); row<OP_PKMUL>();      // the replay loop's quarter-rate and packed instructions
  return 0;
}
