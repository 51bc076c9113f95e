// Probe: do VALU instructions issue in the shadow of MFMA on gfx950, and does it depend on the
// accumulator register class (ArchVGPR vs AccVGPR)?  Prints cycles per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VALU1(x, y) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(y))
#define VALU4(x, y) VALU1(x, y); VALU1(x, y); VALU1(x, y); VALU1(x, y)

template <int NV, bool AGPR>
__global__ __launch_bounds__(256) void probe(long long* out, int iters, float seed) {
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed; c2[r] = seed; c3[r] = seed; }
  f16x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (_Float16)seed; b[r] = (_Float16)(seed + r); }
  uint32_t sink = 0; float y = seed + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define MF(c)                                                                                          \
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#define GAP                                                        \
    if constexpr (NV >= 4) { VALU4(sink, y); }                     \
    if constexpr (NV >= 8) { VALU4(sink, y); }                     \
    if constexpr (NV >= 12) { VALU4(sink, y); }                    \
    if constexpr (NV >= 16) { VALU4(sink, y); }
    MF(c0) GAP MF(c1) GAP MF(c2) GAP MF(c3) GAP MF(c0) GAP MF(c1) GAP MF(c2) GAP MF(c3) GAP
  }
  long long t1 = clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 123.456f || sink == 77) out[1000] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int NV, bool AGPR>
void run(const char* name, int blocks_per_cu) {
  long long* d; hipMalloc(&d, 8192);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NV, AGPR><<<256 * blocks_per_cu, 256>>>(d, iters, 1.0f);
  hipEventRecord(e0);
  probe<NV, AGPR><<<256 * blocks_per_cu, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double tf = 2.0 * 32 * 32 * 16 * 8.0 * iters * 4 * 256 * blocks_per_cu / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD=%d  valu/mfma=%2d  cycles/iter=%7.1f (8 MFMA = 256 pipe cycles/wave)  %.0f us  %.0f TF  clk~%.2f GHz\n",
         name, blocks_per_cu, NV, (double)h / iters, ms * 1e3, tf, (double)h / (ms * 1e-3) / 1e9);
  hipFree(d);
}
int main() {
  for (int occ = 1; occ <= 2; ++occ) {
    run<0, false>("acc=VGPR", occ);  run<4, false>("acc=VGPR", occ);  run<8, false>("acc=VGPR", occ);  run<16, false>("acc=VGPR", occ);
    run<0, true>("acc=AGPR", occ);   run<4, true>("acc=AGPR", occ);   run<8, true>("acc=AGPR", occ);   run<16, true>("acc=AGPR", occ);
  }
  return 0;
}
