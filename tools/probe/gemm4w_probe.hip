// Probe: the k loop of the planes GEMM (3 MFMA products per k-step, LDS-DMA staging) as FOUR waves with 512 registers
// each — one wave per SIMD, accumulators 256 registers, fragments double-buffered, ONE barrier per k-step, every
// ds_read / LDS-DMA issue between the wave's own MFMAs — against the product kernel's eight waves in two ping-pong
// groups (two barriers per k-step, 60-64 % matrix duty in the layer-1 kernels: DESIGN 3.2).
// Shape: the layer-1 forward of config 3 (M = 65536 examples, N = 512, K = 1664), tile 512 columns x 128 examples.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/gemm4w_probe.hip -o tools/probe/gemm4w_probe && tools/probe/gemm4w_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256, ROWB = 64, BK = 16;
constexpr int XT = 8, YT = 2;                         // per wave: 8 x 32 columns, 2 x 32 examples
constexpr int BN = 2 * XT * 32, BM = 2 * YT * 32;     // 2 x 2 waves: 512 x 128
constexpr int ROWS = BN + BM, STAGE = ROWS * ROWB, NBUF = 3;
constexpr int LPS = ROWS * 4 / THREADS;               // LDS-DMA pieces per thread and stage (10)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Frags { f16x8 ah[XT], al[XT], bh[YT], bl[YT]; };

template <bool STORE, int ORDER>
__global__ __launch_bounds__(THREADS, 1) void gemm4w_k(const char* __restrict__ A, int64_t bsa, const char* __restrict__ B, int64_t bsb,
                                                       int M, int N, int K, float* __restrict__ C, float* __restrict__ sink, long long* clk) {
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * STAGE];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wn = wv & 1, wm = wv >> 1;
  const int m0 = blockIdx.x * BM;
  const int nk = K / BK;

  const char* src[LPS];
#pragma unroll
  for (int j = 0; j < LPS; ++j) {
    const int p = j * THREADS + t;
    const int row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
    if (row < BN) src[j] = A + static_cast<int64_t>(min(row, N - 1)) * ROWB + c * 16;
    else src[j] = B + static_cast<int64_t>(min(m0 + row - BN, M - 1)) * ROWB + c * 16;
  }
  auto dma_piece = [&](int j, int kt, int buf) {
    const bool isA = (j * THREADS) / 4 < BN;           // (pieces 0..7: weights, 8..9: examples — whole pieces: BN * 4 % THREADS == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * (isA ? bsa : bsb)),
                                     (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (j * THREADS + wv * 64) * 16), 16, 0, 0);
  };
  const int sw = (i >> 2) & 3;
  const int ch = (h ^ sw) * 16, cl = ((2 | h) ^ sw) * 16;
  const int offA = (wn * 32 * XT + i) * ROWB;
  const int offB = (BN + wm * 32 * YT + i) * ROWB;

  // ORDER 2: the same loop — same loads, same number of flops and accumulator registers — issued as v_mfma_f32_16x16x32_f16
  // (two per 32x32x16, on the same operand registers: the RESULT IS NOT A GEMM, only the energy is meant)
  f32x4 acc4[ORDER == 2 ? 4 * XT * YT : 1];
#pragma unroll
  for (int q = 0; q < (ORDER == 2 ? 4 * XT * YT : 1); ++q) acc4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc[XT][YT];
#pragma unroll
  for (int x = 0; x < XT; ++x)
#pragma unroll
    for (int y = 0; y < YT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  // one fragment read (piece q of the 2 XT + 2 YT of a tile)
  auto read_piece = [&](Frags& f, const char* buf, int q) {
    if (q < XT) f.ah[q] = *reinterpret_cast<const f16x8*>(buf + offA + q * 32 * ROWB + ch);
    else if (q < 2 * XT) f.al[q - XT] = *reinterpret_cast<const f16x8*>(buf + offA + (q - XT) * 32 * ROWB + cl);
    else if (q < 2 * XT + YT) f.bh[q - 2 * XT] = *reinterpret_cast<const f16x8*>(buf + offB + (q - 2 * XT) * 32 * ROWB + ch);
    else f.bl[q - 2 * XT - YT] = *reinterpret_cast<const f16x8*>(buf + offB + (q - 2 * XT - YT) * 32 * ROWB + cl);
  };
  constexpr int NR = 2 * XT + 2 * YT;                 // 20 fragment reads per tile
  constexpr int NM = 3 * XT * YT;                     // 48 MFMAs per tile

  // iteration t: the MFMAs of tile t (fragments `cur`), and between them the fragment reads of tile t + 1 (into `nxt`)
  // and this wave's LDS-DMA share of tile t + 3 (into tile t's buffer, whose reads ended before the last barrier)
  auto step = [&](const Frags& cur, Frags& nxt, int tt) {
    int rb = tt + 1; rb -= (rb / NBUF) * NBUF;
    int fb = tt; fb -= (fb / NBUF) * NBUF;
    const char* rbuf = smem + rb * STAGE;
    const int kt = min(tt + NBUF, nk - 1);
#pragma unroll
    for (int idx = 0; idx < NM; ++idx) {
      // ORDER 0: product-major (an accumulator's three products are XT YT MFMAs apart); ORDER 1: column-tile-major (a
      // column tile's 3 YT MFMAs back to back: the weight operand changes twice per six MFMAs instead of every second one)
      const int pr = ORDER == 0 ? idx / (XT * YT) : (idx % (3 * YT)) / YT;
      const int x = ORDER == 0 ? (idx % (XT * YT)) / YT : idx / (3 * YT);
      const int y = idx % YT;
      if constexpr (ORDER == 2) {
        const int q = 4 * (x * YT + y) + 2 * (pr & 1);
        acc4[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 0 ? cur.al[x] : cur.ah[x], pr == 1 ? cur.bl[y] : cur.bh[y], acc4[q], 0, 0, 0);
        acc4[q + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 1 ? cur.bl[y] : cur.bh[y], pr == 0 ? cur.al[x] : cur.ah[x], acc4[q + 1], 0, 0, 0);
      } else
      acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? cur.al[x] : cur.ah[x], pr == 1 ? cur.bl[y] : cur.bh[y], acc[x][y], 0, 0, 0);
      // 30 side instructions over 48 MFMAs: the 20 reads first (their data is needed at the top of the next
      // iteration), the 10 DMA pieces after
      __builtin_amdgcn_sched_barrier(0);
      if (idx < NR) read_piece(nxt, rbuf, idx);
      else if (idx - NR < LPS) dma_piece(idx - NR, kt, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<LPS>();                                 // tile t + 2 has landed (this wave's share); tile t + 3 may be in flight
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const long long c0 = clock64();
  const long long w0 = wall_clock64();
#pragma unroll
  for (int s = 0; s < NBUF; ++s)
#pragma unroll
    for (int j = 0; j < LPS; ++j) dma_piece(j, min(s, nk - 1), s);
  wait_vmcnt<2 * LPS>();
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
#pragma unroll
  for (int q = 0; q < NR; ++q) read_piece(f0, smem, q);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vmcnt<LPS>();
  __builtin_amdgcn_s_barrier();
  const long long c1 = clock64();
#pragma unroll 1
  for (int tt = 0; tt < nk; tt += 2) {
    step(f0, f1, tt);
    step(f1, f0, tt + 1);
  }
  const long long c2 = clock64();
  const long long w2 = wall_clock64();
  wait_vmcnt<0>();

  float s = 0.f;
  if constexpr (STORE) {
#pragma unroll
    for (int y = 0; y < YT; ++y) {
      const int m = m0 + wm * 32 * YT + y * 32 + i;
#pragma unroll
      for (int x = 0; x < XT; ++x)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = wn * 32 * XT + x * 32 + 8 * g + 4 * h;
          if (m < M && n < N)
            *reinterpret_cast<float4*>(C + static_cast<int64_t>(m) * N + n) =
                make_float4(acc[x][y][4 * g], acc[x][y][4 * g + 1], acc[x][y][4 * g + 2], acc[x][y][4 * g + 3]);
        }
    }
  } else {
#pragma unroll
    for (int x = 0; x < XT; ++x)
#pragma unroll
      for (int y = 0; y < YT; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[x][y][r];
    if constexpr (ORDER == 2)
#pragma unroll
      for (int q = 0; q < 4 * XT * YT; ++q) s += acc4[q][0] + acc4[q][1] + acc4[q][2] + acc4[q][3];
    sink[blockIdx.x * THREADS + t] = s;
  }
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    const int o = blockIdx.x == 0 ? 0 : 4;
    clk[o] = c1 - c0; clk[o + 1] = c2 - c1; clk[o + 2] = w2 - w0; clk[o + 3] = w0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const int M = 65536, N = 512, K = 1664;
  const int nk = K / BK;
  static_assert((1664 / 16) % 2 == 0, "the loop is unrolled by two");
  const size_t abytes = static_cast<size_t>(nk) * N * ROWB, bbytes = static_cast<size_t>(nk) * M * ROWB;
  std::vector<_Float16> ha(abytes / 2), hb(bbytes / 2);
  uint32_t seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 9) & 0xffff) / 65536.0f - 0.5f; };
  // planes: [k block][row][16 hi, 16 lo]; lo ~ 2^-11 of hi
  for (size_t q = 0; q < ha.size(); ++q) ha[q] = static_cast<_Float16>(((q / 16) & 1) ? rnd() * 4.8e-4f : rnd());
  for (size_t q = 0; q < hb.size(); ++q) hb[q] = static_cast<_Float16>(((q / 16) & 1) ? rnd() * 4.8e-4f : rnd());
  char *dA, *dB; float *dC, *sink; long long* clk;
  CK(hipMalloc(&dA, abytes)); CK(hipMalloc(&dB, bbytes));
  CK(hipMalloc(&dC, static_cast<size_t>(M) * N * 4)); CK(hipMalloc(&sink, static_cast<size_t>(M / BM) * THREADS * 4));
  CK(hipMalloc(&clk, 64));
  CK(hipMemcpy(dA, ha.data(), abytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hb.data(), bbytes, hipMemcpyHostToDevice));
  const dim3 grid(M / BM), blk(THREADS);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> hcheck(static_cast<size_t>(M) * N);
  for (int pass = 0; pass < 5; ++pass) {
    const int store = pass == 1, order = pass == 3 ? 1 : pass == 4 ? 2 : 0;   // pass 3: column-tile-major MFMA order, zero operands again? no: live operands restored
    // pass 2: the same loop on ALL-ZERO operands (after the check below has its outputs): same instructions, same
    // bytes, no toggling in the multipliers — what the clock does then says whether the limit is power
    if (pass == 3) { CK(hipMemcpy(dA, ha.data(), abytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), bbytes, hipMemcpyHostToDevice)); }
    if (pass == 2) { CK(hipMemcpy(hcheck.data(), dC, hcheck.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemset(dA, 0, abytes)); CK(hipMemset(dB, 0, bbytes)); }

    auto launch = [&]() {
      if (order == 2) gemm4w_k<false, 2><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk);
      else if (order) { if (store) gemm4w_k<true, 1><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk);
        else gemm4w_k<false, 1><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk); }
      else if (store) gemm4w_k<true, 0><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk);
      else gemm4w_k<false, 0><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk);
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hc[8]; CK(hipMemcpy(hc, clk, 64, hipMemcpyDeviceToHost));
    const double us = ms / reps * 1e3;
    const double flop = 3.0 * 2.0 * M * N * K;
    printf("%s: %.1f us per launch, %.0f TF/s executed (%.2f of 2500), prologue %lld cycles, k loop %lld cycles = %.0f per k-step (48 MFMAs = 1536)\n",
           pass == 4 ? "16x16x32 shape  " : pass == 3 ? "tile-major order" : pass == 2 ? "zero operands   " : store ? "with fp32 store " : "loop only       ", us, flop / us * 1e-6, flop / us * 1e-6 / 2500.0, hc[0], hc[1], (double)hc[1] / nk);
    // wall_clock64: 100 MHz.  First and last workgroup of the grid: their prologue + loop in wall time, the shader clock that implies, when the last one started
    printf("   workgroup 0: %.1f us wall for %lld cycles = %.2f GHz; last workgroup: %.1f us wall for %lld cycles = %.2f GHz, started %.1f us after the first\n",
           hc[2] / 100.0, hc[0] + hc[1], (hc[0] + hc[1]) / (hc[2] / 100.0) * 1e-3, hc[6] / 100.0, hc[4] + hc[5], (hc[4] + hc[5]) / (hc[6] / 100.0) * 1e-3,
           (hc[7] - hc[3]) / 100.0);
  }
  // check a few elements against the planes' own arithmetic (hi*hi + hi*lo + lo*hi)
  std::vector<float>& hc = hcheck;
  double worst = 0;
  for (int s = 0; s < 64; ++s) {
    const int m = (s * 9973 + 17) % M, n = (s * 131 + 5) % N;
    double ref = 0;
    for (int kb = 0; kb < nk; ++kb)
      for (int k = 0; k < 16; ++k) {
        const double ah = (double)ha[(static_cast<size_t>(kb) * N + n) * 32 + k], al = (double)ha[(static_cast<size_t>(kb) * N + n) * 32 + 16 + k];
        const double bh = (double)hb[(static_cast<size_t>(kb) * M + m) * 32 + k], bl = (double)hb[(static_cast<size_t>(kb) * M + m) * 32 + 16 + k];
        ref += ah * bh + ah * bl + al * bh;
      }
    worst = fmax(worst, fabs(hc[static_cast<size_t>(m) * N + n] - ref) / (fabs(ref) + 1e-3));
  }
  printf("max relative error of 64 sampled outputs: %.2e %s\n", worst, worst < 1e-4 ? "(ok)" : "(WRONG)");
  return worst < 1e-4 ? 0 : 1;
}
