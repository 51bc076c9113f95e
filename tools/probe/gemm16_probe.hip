// Probe (round 4): the k loop of the layer-1 forward GEMM on v_mfma_f32_16x16x32_f16 with the planes' 64-byte rows
// [16 hi | 16 lo] taken as ONE k = 32 operand — a REAL loop (LDS-DMA staging, ds_read_b128 fragments, the three
// products of the fp16 high/low split, a correct GEMM: sampled outputs are checked) beside tools/probe/gemm4w_probe.hip,
// the same loop on v_mfma_f32_32x32x16_f16.  The question (profiles/r03_gemm_power_limit.md): the
// layer-1 GEMMs sit at the chip's power limit — does the 16x16x32 shape (half the accumulator bytes read and written
// per flop) buy time there?
//
// Per pair of 16-k blocks (s0, s1) and 16 x 16 output tile, three MFMAs of k = 32 — the same flops as six 32x32x16:
//     P1(s0):  [a_hi(s0) | a_lo(s0)] x [b_hi(s0) | b_hi(s0)]   = a_hi b_hi + a_lo b_hi over s0   (A operand = a whole plane row)
//     P1(s1):  the same over s1
//     P3:      [a_hi(s0) | a_hi(s1)] x [b_lo(s0) | b_lo(s1)]   = a_hi b_lo over both blocks       (no wasted half)
// Four waves, one per SIMD (512 registers: 256 accumulators, two fragment sets of 80), tile 512 columns x 128 examples,
// four 40-KB stage buffers; per pair three phases of 64 MFMAs, each followed by one barrier; the next phase's 20
// fragment reads and the LDS-DMA pieces of stages s + 4, s + 5 sit between the phase's MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/gemm16_probe.hip -o tools/probe/gemm16_probe && tools/probe/gemm16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256, ROWB = 64, BK = 16;
constexpr int RG = 16, CG = 4;                        // per wave: 16 groups of 16 weight rows, 4 groups of 16 examples
constexpr int BN = 2 * RG * 16, BM = 2 * CG * 16;     // 2 x 2 waves: 512 x 128
constexpr int ROWS = BN + BM, STAGE = ROWS * ROWB, NBUF = 4;
constexpr int LPS = ROWS * 4 / THREADS;               // LDS-DMA pieces per thread and stage (10)
constexpr int NR = RG + CG;                           // fragment reads per phase (20)
constexpr int NM = RG * CG;                           // MFMAs per phase (64)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Frags { f16x8 a[RG], b[CG]; };

template <bool STORE>
__global__ __launch_bounds__(THREADS, 1) void gemm16_k(const char* __restrict__ A, int64_t bsa, const char* __restrict__ B, int64_t bsb,
                                                       int M, int N, int K, float* __restrict__ C, float* __restrict__ sink, long long* clk) {
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * STAGE];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
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
  // piece j of stage `stage` into buffer `buf` (= stage % NBUF: a compile-time number at every call)
  auto dma_piece = [&](int j, int stage, int buf) {
    const int kt = min(stage, nk - 1);
    const bool isA = (j * THREADS) / 4 < BN;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * (isA ? bsa : bsb)),
                                     (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (j * THREADS + wv * 64) * 16), 16, 0, 0);
  };
  // Fragment addresses: a lane's offset inside a stage + a compile-time multiple of 1 KiB (16 rows) + the buffer.  The 16-byte
  // chunks of a row are stored XOR-swizzled by (row >> 2) & 3 — the same for all 16-row groups of a lane (16 q rows further).
  // P1: A = chunk kq of the row (hi octets 0, 1, lo octets 2, 3), B = chunk kq & 1 (hi octets, twice)
  // P3: both from the buffer of s0 (kq < 2) or s1 = the NEXT buffer (kq >= 2): A = hi chunk kq & 1, B = lo chunk 2 + (kq & 1)
  const int rowA0 = wn * 16 * RG + r16, rowB0 = BN + wm * 16 * CG + r16;
  const int swA = (rowA0 >> 2) & 3, swB = (rowB0 >> 2) & 3;
  const int p3buf = kq >= 2 ? STAGE : 0;
  const int oA1 = rowA0 * ROWB + ((kq ^ swA) << 4), oB1 = rowB0 * ROWB + (((kq & 1) ^ swB) << 4);
  const int oA3 = rowA0 * ROWB + (((kq & 1) ^ swA) << 4) + p3buf, oB3 = rowB0 * ROWB + (((2 + (kq & 1)) ^ swB) << 4) + p3buf;

  f32x4 acc[RG][CG];
#pragma unroll
  for (int x = 0; x < RG; ++x)
#pragma unroll
    for (int y = 0; y < CG; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read q (0 .. NR - 1) of a phase: kind 0 = P1 of the stage in buffer b0, kind 1 = P3 over (b0, b0 + 1)
  auto read_piece = [&](Frags& f, int kind, int b0, int q) {
    const char* buf = smem + b0 * STAGE;
    if (q < RG) f.a[q] = *reinterpret_cast<const f16x8*>(buf + (kind == 0 ? oA1 : oA3) + q * 16 * ROWB);
    else f.b[q - RG] = *reinterpret_cast<const f16x8*>(buf + (kind == 0 ? oB1 : oB3) + (q - RG) * 16 * ROWB);
  };
  // one phase: the 64 MFMAs of `cur`; between them the 20 fragment reads of the next phase into `nxt` and n_dma LDS-DMA
  // pieces (dma_j0 .. of stage dma_stage into buffer dma_buf, then — phase C — the first pieces of the stage / buffer after it)
  auto phase = [&](const Frags& cur, Frags& nxt, int nkind, int nb0, int dma_stage, int dma_buf, int dma_j0, int n_dma) {
#pragma unroll
    for (int idx = 0; idx < NM; ++idx) {
      const int x = idx / CG, y = idx % CG;
      // (inline asm with the accumulator pinned to the AGPR file: through the builtin hipcc shuffles the 64 four-register
      // accumulators between the two register files — 1,500 v_accvgpr moves and 300 bytes of scratch in this loop)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[x][y]) : "v"(cur.a[x]), "v"(cur.b[y]));
      __builtin_amdgcn_sched_barrier(0);
      if (idx < NR) read_piece(nxt, nkind, nb0, idx);
      else if (idx - NR < n_dma) {
        const int q = dma_j0 + idx - NR;               // pieces run on into the next stage (phase C: 5 of s + 4, 5 of s + 5)
        dma_piece(q % LPS, dma_stage + q / LPS, (dma_buf + q / LPS) & (NBUF - 1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const long long c0 = clock64();
  const long long w0 = wall_clock64();
#pragma unroll
  for (int s = 0; s < NBUF; ++s)
#pragma unroll
    for (int j = 0; j < LPS; ++j) dma_piece(j, s, s);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
#pragma unroll
  for (int q = 0; q < NR; ++q) read_piece(f0, 0, 0, q);               // P1(s0 = stage 0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long c1 = clock64();
  // One pair (s0 = 2 i in buffer b0, s1 = 2 i + 1 in b1; the other two buffers b2, b3 hold s0 + 2, s1 + 2):
  //   phase A: MFMA P1(s0)  | read P3 <- b0, b1          | DMA: the last 5 pieces of stage s1 + 2 (into b3, free since the last pair's phase B)
  //   barrier                                              (b0 is free)
  //   phase B: MFMA P3      | read P1(s1) <- b1          | DMA: pieces 0-4 of stage s0 + 4 (into b0)
  //   barrier, vmcnt: stage s0 + 2 has landed              (b1 is free)
  //   phase C: MFMA P1(s1)  | read next P1(s0') <- b2    | DMA: pieces 5-9 of s0 + 4, pieces 0-4 of s1 + 4 (into b1)
  //   barrier, vmcnt: stage s1 + 2 has landed
  // (in-order completion: at the end of B 15 younger pieces may be outstanding, at the end of C 15 as well)
  // (b0 — 0 for an even pair, 2 for an odd one — is a compile-time number at both call sites: every LDS address in the
  // loop is a lane offset + an immediate)
  auto pair = [&](Frags& fa, Frags& fb, int i, int b0) {    // enters with fa = P1(s0) fragments; leaves with fb = next P1(s0')
    const int s0 = 2 * i, b1 = b0 + 1, b2 = (b0 + 2) & 3, b3 = (b0 + 3) & 3;
    phase(fa, fb, 1, b0, s0 + 3, b3, 5, i > 0 ? 5 : 0);
    sync();
    phase(fb, fa, 0, b1, s0 + 4, b0, 0, 5);
    wait_vmcnt<15>();
    sync();
    phase(fa, fb, 0, b2, s0 + 4, b0, 5, 10);
    wait_vmcnt<15>();
    sync();
  };
#pragma unroll 1
  for (int i = 0; i < nk / 2; i += 2) {
    pair(f0, f1, i, 0);
    pair(f1, f0, i + 1, 2);
  }
  const long long c2 = clock64();
  const long long w2 = wall_clock64();
  wait_vmcnt<0>();

  if constexpr (STORE) {
#pragma unroll
    for (int y = 0; y < CG; ++y) {
      const int m = m0 + wm * 16 * CG + y * 16 + r16;
#pragma unroll
      for (int x = 0; x < RG; ++x) {
        const int n = wn * 16 * RG + x * 16 + 4 * kq;
        if (m < M && n < N)
          *reinterpret_cast<float4*>(C + static_cast<int64_t>(m) * N + n) = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
      }
    }
  } else {
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < RG; ++x)
#pragma unroll
      for (int y = 0; y < CG; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
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
  static_assert((1664 / 16) % 4 == 0, "the loop walks two pairs of k-blocks per trip");
  const size_t abytes = static_cast<size_t>(nk) * N * ROWB, bbytes = static_cast<size_t>(nk) * M * ROWB;
  std::vector<_Float16> ha(abytes / 2), hb(bbytes / 2);
  uint32_t seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 9) & 0xffff) / 65536.0f - 0.5f; };
  // planes: [k block][row][16 hi, 16 lo]; lo ~ 2^-11 of hi (the same operands as gemm4w_probe)
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
  for (int pass = 0; pass < 4; ++pass) {
    const bool store = pass == 1;
    // pass 2: ALL-ZERO operands (same instructions, same bytes, nothing toggles in the multipliers); pass 3: live again
    if (pass == 2) { CK(hipMemcpy(hcheck.data(), dC, hcheck.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemset(dA, 0, abytes)); CK(hipMemset(dB, 0, bbytes)); }
    if (pass == 3) { CK(hipMemcpy(dA, ha.data(), abytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), bbytes, hipMemcpyHostToDevice)); }
    auto launch = [&]() {
      if (store) gemm16_k<true><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk);
      else gemm16_k<false><<<grid, blk>>>(dA, (int64_t)N * ROWB, dB, (int64_t)M * ROWB, M, N, K, dC, sink, clk);
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
    printf("%s: %.1f us per launch, %.0f TF/s executed (%.2f of 2500), prologue %lld cycles, k loop %lld cycles = %.0f per 16 k (96 MFMAs 16x16x32 = 1536)\n",
           pass == 3 ? "live operands again" : pass == 2 ? "zero operands      " : store ? "with fp32 store    " : "loop only          ", us, flop / us * 1e-6,
           flop / us * 1e-6 / 2500.0, hc[0], hc[1], (double)hc[1] / nk);
    printf("   workgroup 0: %.1f us wall for %lld cycles = %.2f GHz; last workgroup: %.1f us wall for %lld cycles = %.2f GHz, started %.1f us after the first\n",
           hc[2] / 100.0, hc[0] + hc[1], (hc[0] + hc[1]) / (hc[2] / 100.0) * 1e-3, hc[6] / 100.0, hc[4] + hc[5], (hc[4] + hc[5]) / (hc[6] / 100.0) * 1e-3,
           (hc[7] - hc[3]) / 100.0);
  }
  // sampled outputs against the planes' own arithmetic (hi*hi + hi*lo + lo*hi)
  double worst = 0;
  for (int s = 0; s < 256; ++s) {
    const int m = (s * 9973 + 17) % M, n = (s * 131 + 5) % N;
    double ref = 0;
    for (int kb = 0; kb < nk; ++kb)
      for (int k = 0; k < 16; ++k) {
        const double ah = (double)ha[(static_cast<size_t>(kb) * N + n) * 32 + k], al = (double)ha[(static_cast<size_t>(kb) * N + n) * 32 + 16 + k];
        const double bh = (double)hb[(static_cast<size_t>(kb) * M + m) * 32 + k], bl = (double)hb[(static_cast<size_t>(kb) * M + m) * 32 + 16 + k];
        ref += ah * bh + ah * bl + al * bh;
      }
    worst = fmax(worst, fabs(hcheck[static_cast<size_t>(m) * N + n] - ref) / (fabs(ref) + 1e-3));
  }
  printf("max relative error of 256 sampled outputs: %.2e %s\n", worst, worst < 1e-4 ? "(ok)" : "(WRONG)");
  return worst < 1e-4 ? 0 : 1;
}
