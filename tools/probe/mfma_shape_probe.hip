// Probe: under this chip's power management, which f16 MFMA shape delivers more FLOP/s on RANDOM operands —
// v_mfma_f32_32x32x16_f16 (what the planes GEMMs issue) or v_mfma_f32_16x16x32_f16?  Same flops per "unit" (one
// 32x32x16 = two 16x16x32 ... no: 32x32x16 = 32768 flop, 16x16x32 = 16384 flop), operands in registers, one wave per SIMD
// and two, 3 products per accumulator (hi*hi, hi*lo, lo*hi: the split's mix).  MI355X_MICROARCH.md (DVFS item 7) reports
// 1.12-1.15x for bf16; this measures f16 on the box at hand.  Prints TF/s and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool SMALL>
__global__ __launch_bounds__(256) void probe(const f16x8* __restrict__ src, float* out, long long* clk, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int ts = t & 65535;                       // (src holds 4 x 65536 vectors)
  f16x8 a0 = src[ts], a1 = src[ts + 65536], b0 = src[ts + 131072], b1 = src[ts + 196608];
  const long long t0 = clock64();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SMALL) {
    f32x4 c[8];
    for (int q = 0; q < 8; ++q) c[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[q], 0, 0, 0);
        c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[q], 0, 0, 0);
        c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[q], 0, 0, 0);
      }
      // (operands change a little every iteration so that the products are not constant)
      a0[it & 7] = (_Float16)((float)a0[it & 7] * 1.0009765625f);
    }
    for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  } else {
    f32x16 c[2];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c[q], 0, 0, 0);
          c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c[q], 0, 0, 0);
          c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[q], 0, 0, 0);
        }
      a0[it & 7] = (_Float16)((float)a0[it & 7] * 1.0009765625f);
    }
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  }
  const long long t1 = clock64();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  out[t] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <bool SMALL>
void run(const char* name, int waves_per_simd, const f16x8* src) {
  float* out; long long* clk;
  hipMalloc(&out, 4 * 256 * 256 * 2 * sizeof(float)); hipMalloc(&clk, 64);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;       // 256 threads = 4 waves = one per SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) probe<SMALL><<<blocks, 256>>>(src, out, clk, iters);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) probe<SMALL><<<blocks, 256>>>(src, out, clk, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // flops per wave and iteration: SMALL 24 x 16x16x32 (16384) ; else 12 x 32x32x16 (32768) — the same
  const double fl = 24.0 * 16384.0 * iters * blocks * 4;
  printf("%-24s waves/SIMD=%d  %8.1f us  %7.1f TF/s f16 executed  cycles/iter %.1f (ideal 384)  clock %.2f GHz\n", name, waves_per_simd, ms * 1e3,
         fl / (ms * 1e-3) / 1e12, (double)h[0] / iters, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
  hipFree(out); hipFree(clk);
}

int main() {
  const size_t n = 4 * 65536;
  f16x8* src; hipMalloc(&src, n * sizeof(f16x8));
  _Float16* hsrc = (_Float16*)malloc(n * 16);
  srand(1);
  for (size_t i = 0; i < n * 8; ++i) hsrc[i] = (_Float16)(((rand() % 2001) - 1000) * (1.0f / 64.0f) * ((rand() & 1) ? 1.f : 0.f));   // half zeros, like relu'd rows
  hipMemcpy(src, hsrc, n * 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int occ = 1; occ <= 2; ++occ) {
      run<false>("v_mfma_32x32x16_f16", occ, src);
      run<true>("v_mfma_16x16x32_f16", occ, src);
    }
  return 0;
}
