// fp32-accurate GEMMs of the MLP on the f16 matrix pipe, operands held as PRE-SPLIT 16-bit planes.
//
// Replaces tf.layers.dense / tf.layers.dropout (trainers/deep_fm.py:98-108) and their data gradients
// for every layer whose shape allows it (gemm.hip keeps the any-shape kernels; the weight gradient from planes is
// wgrad_pl.hip).
//
// Operand format ("planes", mi_planes_t): a matrix [rows][K] is stored as fp16 high + fp16 low parts
// of x * 2^s_r with ONE power-of-two exponent s_r PER ROW (row max -> [2^14, 2^15)):
//     x * 2^s_r = hi + lo,   hi = fp16(x * 2^s_r)  (RNE),  lo = fp16(x * 2^s_r - hi)
// so every row keeps ~22 significant bits relative to ITS OWN largest element (elements more than
// 2^-17 below their row's maximum lose low bits gradually; the round-1 matrix-wide scale lost them for
// whole rows, e.g. the dY rows of well-fit examples).  In memory the matrix is K-BLOCK MAJOR: for each block
// of 16 k, all rows back to back, a row's piece being 16 x hi then 16 x lo (64 B); blocks are blk_stride
// bytes apart.  The 16-k tile of any range of rows is therefore ONE contiguous run of full cache lines —
// what a workgroup's LDS-DMA stage reads (measured against row-major pieces of 64 B, 6.6 KB apart: layer-1
// forward 422 -> 339 us).
//
// Both operands of these GEMMs are "k-contiguous" (NT form):
//     forward        Y[m][n]  = sum_k X[m][k]  * Wt[n][k]      (Wt = planes of W transposed, rows = n)
//     data gradient  dX[m][n] = sum_k dY[m][k] * W [n][k]      (W  = planes of W as stored,  rows = n)
// The row exponents are undone exactly in the epilogue (per output column for the weights, per output
// row for the activations).  Three MFMA products per k-step: lo*hi, hi*lo, hi*hi (the dropped lo*lo is
// <= 2^-22 |ab|); a product of two fp16 values is exact in fp32 and the matrix pipe accumulates in fp32.
//
// Kernel: one 512-thread workgroup (8 waves as 4 x 2) per CU computes a tile of (128 TN) output columns x
// (64 TM) output rows; the WEIGHT rows are the MFMA's A operand and the EXAMPLE rows its B operand, so a
// lane owns one example (column of the 32x32 accumulator tile) and its registers run over output columns:
// per-example quantities (row exponent, row abs-max of the result) are per-lane scalars, and a lane pair
// assembles 64 contiguous output bytes (16 hi + 16 lo) with v_permlane32_swap — no LDS transpose.
// With TN = 4 a workgroup holds ALL 512 columns of a row: the epilogue knows the row's abs-max and writes
// the next layer's operand directly as planes.
// Staging is LDS-DMA only (global_load_lds_dwordx4, no register staging, no VALU in the loop): 3 stage
// buffers of 16 k, counted vmcnt + raw s_barrier, one barrier per k-tile, loads two tiles ahead; the LDS
// image is dense 64-B rows with the 16-B chunk index XOR-swizzled by (row >> 2) & 3 on the SOURCE address
// (ds_read_b128 fragment reads are then conflict-free).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float fl32x2 __attribute__((ext_vector_type(2)));

constexpr int PL_THREADS = 512;
constexpr int PL_BK = 16;
constexpr int PL_ROWB = 64;       // bytes per tile row and stage: 16 k x (hi, lo) x 2 B

enum { PL_FWD = 0, PL_DGRAD = 1, PL_TOP = 2 };

struct PlArgs {
  const char* A; int64_t bsa; const int32_t* a_exp;     // weight side (rows = output columns); bsa: k-block stride in bytes
  const char* B; int64_t bsb; const int32_t* b_exp;     // example side (rows = output rows)
  int M, N, K;
  int tiles_n;
  float* C; int64_t ldc;                                // fp32 result [M][N] or NULL
  char* Cp; int64_t bsc; int32_t* c_exp;                // planes of the result or NULL (one column tile only)
  const float* bias; int relu; float keep_prob, keep_div, keep_rcp; uint64_t seed;   // keep_rcp = RN(1 / keep_div), from the host
  const char* mask; int64_t bsm;                        // dgrad: planes of the stored activation (hi > 0 <=> active & kept)
  // The relu/dropout mask as ONE BIT per element, [M][mbld] 32-bit words, bit c & 31 of word c >> 5 of a row = "output c is
  // active and kept" (value > 0): written by the forward epilogue (FWD: an output), read by the data gradient instead of the
  // stored activation's high plane (DGRAD: an input; 2 MB instead of 134 MB at 65536 x 512) and by the logits layer's.
  uint32_t* mbits; int64_t mbld;
  float* amax_c;                                        // abs-max vector of the result (the weight gradient's matrix-wide scales) or NULL
  const mi_step_state_t* st;                            // device-resident step state of a captured step, or NULL
  // PL_TOP (mi_hidden_logits_head_fused): the one-unit logits layer and the head behind this layer, in its epilogue
  const float* top_w; const float* top_b;               // logits layer: weights [N], bias [1] (or NULL)
  const float* top_lin; const float* top_lin_bias; const float* top_fm; const uint8_t* top_labels; float top_scale;
  float* top_dnn; float* top_logits; float* top_dlogit; // per example (top_dnn may be NULL)
  float* top_part;                                      // [workgroups][N + 2]: the logits layer's dW partial, sum d, sum loss
};

__device__ __forceinline__ float pl_pow2(int s) { return __uint_as_float(static_cast<uint32_t>(127 + s) << 23); }
// exponent s with amax * 2^s in [2^14, 2^15), clamped so that 2^s and 2^-s are normal numbers
__device__ __forceinline__ int pl_exp_for(float amax) {
  const int e = static_cast<int>((__float_as_uint(amax) >> 23) & 0xffu);
  return max(-100, min(100, 141 - e));
}

template <int N> __device__ __forceinline__ void pl_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// -DMI_PL_STAMPS: clock64() stamps at the phase boundaries of every k-step, from lane 0 of waves 0 (group 0) and 4
// (group 1) of the first 32 workgroups (tools/gemm_pl_stamps.py reads them back).  Not part of the product build.
#ifdef MI_PL_STAMPS
__device__ long long g_pl_stamps[32 * 2 * 128 * 8];
#define PL_STAMP(slot) do { if (stamp_on && t < 128) g_pl_stamps[((stamp_wg * 2 + stamp_grp) * 128 + t) * 8 + (slot)] = clock64(); } while (0)
// whole-workgroup marks in the unused slot 7 of rows 0..3: kernel entry, first k-step, end of the k loop, end of the epilogue
#define PL_MARK(row) do { if (stamp_on) g_pl_stamps[((stamp_wg * 2 + stamp_grp) * 128 + (row)) * 8 + 7] = clock64(); } while (0)
#else
#define PL_STAMP(slot) do {} while (0)
#define PL_MARK(row) do {} while (0)
#endif

// HOT: the call of a steady-state training step, its launch-uniform options fixed at compile time — whole tiles (M a multiple
// of the row tile, N the column tile), no fp32 copy of the result, planes out; FWD: dropout on and the mask bits written; DGRAD:
// the mask read as bits.  The generic kernel tests these per launch, and hipcc keeps every option's code and a scalar branch
// or a select per group of columns for each: ~4,000-5,900 instructions per wave of epilogue, which two waves per SIMD issue
// in 28-31 k cycles per tile (in-kernel marks, tools/gemm_pl_timeline.py: as long as 12 of the 16-32 k-steps of a small layer).
template <int TN, int TM, int EPI, bool HOT = false>
// (TN == 1: two workgroups share a CU — 4 waves per SIMD, at most 128 registers: said, not hoped)
__global__ __launch_bounds__(PL_THREADS, 2) void gemm_pl_k(const PlArgs a) {
  // stage buffers: 4 where one workgroup per CU runs anyway (256 registers); 3 for the 128-column tile, whose
  // 128 registers and 72 KB let two workgroups share a CU (one stores its result while the other computes)
  constexpr int PL_NBUF = TN == 1 ? 3 : 4;
  constexpr int BNt = 128 * TN, BMt = 64 * TM, ROWS = BNt + BMt;
  constexpr int LPS = ROWS * 4 / PL_THREADS;             // LDS-DMA instructions per thread and stage
  static_assert(ROWS * 4 % PL_THREADS == 0, "tile rows");
  constexpr int STAGE = ROWS * PL_ROWB;
  static_assert(PL_NBUF * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[PL_NBUF * STAGE];

  // XCD-aware bijective remap (blocks round-robin over the 8 XCDs): the column tiles of one row panel,
  // which read the same activation rows, run on one XCD
  const int nb = gridDim.x, bid = blockIdx.x;
  const int qq = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < rr) ? xcd * (qq + 1) + idx : rr * (qq + 1) + (xcd - rr) * qq + idx;
  const int n0 = (lid % a.tiles_n) * BNt, m0 = (lid / a.tiles_n) * BMt;
  const int nk = a.K / PL_BK;

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wn = wv & 3, wm = wv >> 2;

  // ---- LDS-DMA source addresses: chunk p = j * 512 + t of a stage lands at LDS byte 16 p ----
  const char* src[LPS];
#pragma unroll
  for (int j = 0; j < LPS; ++j) {
    const int p = j * PL_THREADS + t;
    const int row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
    if (j < TN) {                                          // rows [0, BNt): weights
      const int n = min(n0 + row, a.N - 1);
      src[j] = a.A + static_cast<int64_t>(n) * PL_ROWB + c * 16;
    } else {                                               // rows [BNt, ROWS): examples
      const int m = min(m0 + row - BNt, a.M - 1);
      src[j] = a.B + static_cast<int64_t>(m) * PL_ROWB + c * 16;
    }
  }

  // ---- fragment addresses (bytes inside a stage) ----
  const int sw = (i >> 2) & 3;
  const int ch = (h ^ sw) * 16, cl = ((2 | h) ^ sw) * 16;
  const int offA = (wn * 32 * TN + i) * PL_ROWB;
  const int offB = (BNt + wm * 32 * TM + i) * PL_ROWB;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int x = 0; x < TN; ++x)
#pragma unroll
    for (int y = 0; y < TM; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  // Software pipeline: the two waves of a SIMD ping-pong.  The workgroup's 8 waves form two groups (waves
  // 0-3 and 4-7; wave w and w + 4 share a SIMD).  Every wave runs the same loop
  //     L(t): read the fragments of tile t from LDS, wait for its own LDS-DMA share of stage t + 1
  //     barrier
  //     C(t): the 3 TN TM MFMAs of tile t, from registers only, with its LDS-DMA share of stage t - 1 + NBUF
  //           issued between them
  //     barrier
  // but group 1 enters it one barrier late (and group 0 leaves it one barrier late), so between any two
  // consecutive barriers one group is in C and the other in L: each SIMD's matrix pipe always has exactly
  // one wave feeding it, and every ds_read and wait of a wave sits under its partner's MFMAs.  With the barriers
  // numbered globally, group 0 runs L(t) in slot [2t, 2t+1] and C(t) in [2t+1, 2t+2]; group 1 runs L(t) in
  // [2t+1, 2t+2] and C(t) in [2t+2, 2t+3].  Tile t lives in stage buffer t % NBUF.
  //   * a wave waits for its own share of stage t + 1 at the end of L(t): all shares are complete before
  //     barrier 2t+2, the first read of tile t + 1 comes after it;
  //   * C(t) refills the buffer tile t - 1 has left: its last reader was group 1's L(t - 1), which drained
  //     its LDS reads (lgkmcnt(0)) before barrier 2t, and no C(t) starts before barrier 2t+1;
  //   * at the end of L(t) a wave has issued stages up to t - 2 + NBUF (and, at t = 0, one duplicate) and needs
  //     stage t + 1: vmcnt((NBUF - 3) LPS); a share is in flight for 2 NBUF - 4 slots.
  auto issue_share = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < LPS; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * (j < TN ? a.bsa : a.bsb)),
                                       (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (j * PL_THREADS + wv * 64) * 16),
                                       16, 0, 0);
  };
  f16x8 ah[TN], al[TN], bh[TM], bl[TM];
  auto read_frags = [&](int bufi) {
    const char* buf = smem + bufi * STAGE;
#pragma unroll
    for (int x = 0; x < TN; ++x) {
      ah[x] = *reinterpret_cast<const f16x8*>(buf + offA + x * 32 * PL_ROWB + ch);
      al[x] = *reinterpret_cast<const f16x8*>(buf + offA + x * 32 * PL_ROWB + cl);
    }
#pragma unroll
    for (int y = 0; y < TM; ++y) {
      bh[y] = *reinterpret_cast<const f16x8*>(buf + offB + y * 32 * PL_ROWB + ch);
      bl[y] = *reinterpret_cast<const f16x8*>(buf + offB + y * 32 * PL_ROWB + cl);
    }
  };
  // C(t): the 3 TN TM MFMAs of tile t from registers, and — spread between them — this wave's LDS-DMA share of stage
  // t - 1 + NBUF into the buffer tile t - 1 has left.  (Stamps, tools/gemm_pl_stamps.py: issued at the head of L(t)
  // next to the fragment reads, the 5 DMA pieces made L 1000 cycles long against C's 870 — a piece costs 100-185
  // cycles of issue in a phase that also carries ds_reads, ~60 among bare MFMAs — and L, not the matrix pipe, set
  // the pace.)  At t = 0 the share re-fetches stage NBUF - 1 into its own buffer: the same bytes, uniform counts.
  // (The 128-column tile — 3 buffers, two workgroups per CU — keeps its share at the head of L(t): with one stage
  // fewer in flight it would wait for every DMA in every k-step; measured 399 vs 390 us on the layer-1 data gradient.)
  // (Tried: two of the five pieces back at the head of L(t), which has slack after the move — a piece there costs
  // ~480 cycles, not 40: L 1200, C 1205 cycles, 344 vs 313 us.  All pieces stay between the MFMAs.)
  constexpr bool DMA_IN_C = TN != 1;
  auto phase_c = [&](int t) {
    int fb = t - 1 + PL_NBUF;
    fb -= (fb / PL_NBUF) * PL_NBUF;
    const int kt = min(t - 1 + PL_NBUF, nk - 1);          // (the tail re-issues the last tile: uniform counts)
    // product-major order (an accumulator's three products are TN TM MFMAs apart, smallest partial product first);
    // after every GAP MFMAs one DMA piece, pinned there (hipcc otherwise bunches the pieces and chains the
    // dependent MFMAs back to back)
    constexpr int NT = TN * TM, NM = 3 * NT, GAP = NM / (LPS + 1);
#pragma unroll
    for (int idx = 0; idx < NM; ++idx) {
      const int pr = idx / NT, x = (idx % NT) / TM, y = idx % TM;
      acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[x] : ah[x], pr == 1 ? bl[y] : bh[y], acc[x][y], 0, 0, 0);
      if (DMA_IN_C && (idx + 1) % GAP == 0 && (idx + 1) / GAP <= LPS) {
        const int j = (idx + 1) / GAP - 1;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * (j < TN ? a.bsa : a.bsb)),
                                         (__attribute__((address_space(3))) void*)(smem + fb * STAGE + (j * PL_THREADS + wv * 64) * 16),
                                         16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // L(t): the fragments of tile t
  auto phase_l = [&](int t) {
    if (!DMA_IN_C && t >= 1) {
      int fb = t - 1;
      fb -= (fb / PL_NBUF) * PL_NBUF;
      issue_share(min(t - 1 + PL_NBUF, nk - 1), fb);
    }
    int rb = t;
    rb -= (rb / PL_NBUF) * PL_NBUF;
    read_frags(rb);
  };
  const bool g1 = __builtin_amdgcn_readfirstlane(wm) != 0;

#ifdef MI_PL_STAMPS
  // 32 workgroups spread over the grid (every generation of workgroups is sampled)
  const int stamp_every = max(1, nb / 32);
  const bool stamp_on = lane == 0 && (wv == 0 || wv == 4) && lid % stamp_every == 0 && lid / stamp_every < 32;
  const int stamp_wg = lid / stamp_every, stamp_grp = wv >> 2;
#endif
  PL_MARK(0);
#pragma unroll
  for (int s = 0; s < PL_NBUF; ++s) issue_share(min(s, nk - 1), s);
  pl_wait_vmcnt<LPS*(PL_NBUF - 1)>();                      // own share of stage 0
  __builtin_amdgcn_s_barrier();                            // barrier 0
  if (g1) __builtin_amdgcn_s_barrier();                    // the stagger
  // (sched_barrier(0): hipcc otherwise moves register-only MFMAs across s_barrier, which would put both
  // groups' MFMAs into the same slot)
  PL_MARK(1);
#pragma unroll 1
  for (int t = 0; t < nk; ++t) {
    PL_STAMP(0);
    phase_l(t);
    PL_STAMP(1);                                           // DMA + fragment reads issued
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PL_STAMP(2);                                           // fragments in registers
    pl_wait_vmcnt<LPS*(DMA_IN_C ? PL_NBUF - 3 : PL_NBUF - 2)>();   // own share of stage t + 1 (issued in C(t + 2 - NBUF) / L(t + 2 - NBUF))
    PL_STAMP(3);                                           // own DMA share of the next stage has landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PL_STAMP(4);                                           // past the barrier: MFMAs start
    __builtin_amdgcn_s_setprio(1);
    phase_c(t);
    __builtin_amdgcn_s_setprio(0);
    PL_STAMP(5);                                           // MFMAs issued
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PL_STAMP(6);                                           // past the second barrier
  }
  if (!g1) __builtin_amdgcn_s_barrier();
  PL_MARK(2);

  // ---------------------------------------------------------------- PL_TOP: the last hidden layer + logits layer + head
  // (mi_hidden_logits_head_fused; TN = 1: the workgroup's 128 columns are the whole layer.)  The layer's output never leaves the
  // chip: it stays in the accumulators through
  //   h     = dropout(relu(acc 2^-s_x 2^-s_w + bias))                     (the FWD epilogue's arithmetic)
  //   dnn   = h . w + b ; logits = lin + lin_bias + fm + dnn ; loss, d = (sigmoid(logits) - label) scale   (tail.hip's)
  //   dW   += h d ; db += d                                               (per-workgroup partials, folded by tail_fold_k)
  //   dX    = d w, kept where h > 0, divided by keep_prob  -> planes       (what the layer below's data gradient reads)
  // A lane owns an example (i) and 16 of its 128 columns (wave column group wn, half h, registers r): sums over a row's
  // columns go lane pair -> LDS -> the four column groups in order; sums over examples go through a 32-lane butterfly.
  if constexpr (EPI == PL_TOP) {
    static_assert(TN == 1, "PL_TOP: one 128-column tile");
    pl_wait_vmcnt<0>();
    __syncthreads();
    float* e_fw = reinterpret_cast<float*>(smem);     // [128] 2^-s of the weight row
    float* e_bias = e_fw + 128;                       // [128]
    float* e_w = e_bias + 128;                        // [128] logits-layer weights
    float* e_part = e_w + 128;                        // [4][BMt] per column group: dot-product partials, then row abs-max
    float* e_dw = e_part + 4 * BMt;                   // [2][128] dW partial of each row half of the tile
    float* e_sc = e_dw + 256;                         // [2][2] per row half: sum loss, sum d
    float* e_wmax = e_sc + 4;                         // [8]
    static_assert((128 * 3 + 4 * BMt + 256 + 4 + 8) * 4 <= 8192 && 8192 + 8 * 4096 <= PL_NBUF * STAGE, "PL_TOP LDS");
    char* wreg = smem + 8192 + wv * 4096;
    for (int n = t; n < 128; n += PL_THREADS) {
      const bool ok = n < a.N;
      e_fw[n] = ok ? pl_pow2(-a.a_exp[ok ? n : 0]) : 0.f;
      e_bias[n] = (a.bias && ok) ? a.bias[n] : 0.f;
      e_w[n] = ok ? a.top_w[n] : 0.f;
    }
    __syncthreads();
    const bool drop = a.keep_prob < 1.f;
    const uint32_t thresh16 = drop ? mi_drop_thresh16(a.keep_prob) : 0x10000u;
    const uint64_t seed = a.seed + (a.st ? a.st->seed_term : 0ull);
    const int nw = wn * 32;
    const uint32_t pair_base = static_cast<uint32_t>((nw + 4 * h) >> 1) * MI_DROP_PAIR_MUL;
    const float kd = a.keep_div, kr = a.keep_rcp;       // (1 and 1 without dropout)
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    // this lane's 16 logits-layer weights: register r <-> column nw + 8 (r >> 2) + 4 h + (r & 3)
    float wr[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 w4 = *reinterpret_cast<const float4*>(e_w + nw + 8 * g + 4 * h);
      wr[4 * g] = w4.x; wr[4 * g + 1] = w4.y; wr[4 * g + 2] = w4.z; wr[4 * g + 3] = w4.w;
    }
    // ---- the layer's output into the accumulators, and this lane's share of h . w
#pragma unroll
    for (int y = 0; y < TM; ++y) {
      const int rl = wm * 32 * TM + y * 32 + i;
      const int m = m0 + rl;
      const int mc = m < a.M ? m : a.M - 1;
      const float fx = pl_pow2(-a.b_exp[mc]);
      const uint32_t rowkey = drop ? mi_drop_rowkey(seed, static_cast<uint32_t>(m)) : 0u;
      float p = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 fw4 = *reinterpret_cast<const float4*>(e_fw + nw + 8 * g + 4 * h);
        const float4 b4 = *reinterpret_cast<const float4*>(e_bias + nw + 8 * g + 4 * h);
        uint32_t w2[2] = {0u, 0u};
        if (drop) {
          const uint32_t pt = pair_base + static_cast<uint32_t>((8 * g) >> 1) * MI_DROP_PAIR_MUL;
          w2[0] = mi_drop_pairhash(rowkey, pt);
          w2[1] = mi_drop_pairhash(rowkey, pt + MI_DROP_PAIR_MUL);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const float fwj = j == 0 ? fw4.x : j == 1 ? fw4.y : j == 2 ? fw4.z : fw4.w;
          float v = (acc[0][y][r] * fx) * fwj;
          v += j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
          v = fmaxf(v, relu_floor);
          const uint32_t bits16 = (j & 1) ? (w2[j >> 1] >> 16) : (w2[j >> 1] & 0xffffu);
          v = bits16 < thresh16 ? mi_div_const(v, kd, kr) : 0.f;
          acc[0][y][r] = v;
          p = fmaf(v, wr[r], p);
        }
      }
      p += __shfl_xor(p, 32);                          // the lane pair that shares this example
      if (h == 0) e_part[wn * BMt + rl] = p;
    }
    __syncthreads();
    // ---- the head, per example (every column group computes its rows' values; group 0 writes them)
    const float b0 = a.top_b ? a.top_b[0] : 0.f;
    const float lb = (a.top_lin && a.top_lin_bias) ? a.top_lin_bias[0] : 0.f;
    float gy[TM];
    float acc_l = 0.f, acc_d = 0.f;
#pragma unroll
    for (int y = 0; y < TM; ++y) {
      const int rl = wm * 32 * TM + y * 32 + i;
      const int m = m0 + rl;
      const bool mok = m < a.M;
      const int mc = mok ? m : a.M - 1;
      const float dnn = ((e_part[rl] + e_part[BMt + rl]) + (e_part[2 * BMt + rl] + e_part[3 * BMt + rl])) + b0;
      float z = 0.f;
      if (a.top_lin) z += a.top_lin[mc] + lb;
      if (a.top_fm) z += a.top_fm[mc];
      z += dnn;
      const float yl = a.top_labels[mc] ? 1.f : 0.f;
      const float loss = (fmaxf(z, 0.f) - z * yl + log1pf(expf(-fabsf(z)))) * a.top_scale;
      const float e = expf(-fabsf(z));
      const float sg = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      const float g = mok ? (sg - yl) * a.top_scale : 0.f;
      gy[y] = g;
      if (mok && wn == 0 && h == 0) {
        if (a.top_dnn) a.top_dnn[m] = dnn;
        a.top_logits[m] = z;
        a.top_dlogit[m] = g;
        acc_l += loss; acc_d += g;
      }
    }
    __syncthreads();                                    // e_part is about to hold the row maxima
    // ---- the logits layer's backward: dW partial sums (registers, over the tile's rows), dX into the accumulators
    float dw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[r] = 0.f;
    float rmx[TM];
#pragma unroll
    for (int y = 0; y < TM; ++y) {
      const int rl = wm * 32 * TM + y * 32 + i;
      const float g = gy[y];
      float mx = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float hv = acc[0][y][r];
        dw[r] = fmaf(hv, g, dw[r]);
        const float v = hv > 0.f ? mi_div_const(g * wr[r], kd, kr) : 0.f;
        acc[0][y][r] = v;
        mx = fmaxf(mx, fabsf(v));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      rmx[y] = mx;
      if (h == 0) e_part[wn * BMt + rl] = mx;
    }
    // sums over the 32 examples of this wave (both y blocks already added): a butterfly inside each half-wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sv = dw[r];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) sv += __shfl_xor(sv, o);
      dw[r] = sv;
    }
    if (i == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(e_dw + wm * 128 + nw + 8 * g + 4 * h) = make_float4(dw[4 * g], dw[4 * g + 1], dw[4 * g + 2], dw[4 * g + 3]);
    }
    if (wn == 0) {                                      // (lanes with h == 1 hold zeros)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { acc_l += __shfl_xor(acc_l, o); acc_d += __shfl_xor(acc_d, o); }
      if (lane == 0) { e_sc[wm * 2] = acc_l; e_sc[wm * 2 + 1] = acc_d; }
    }
    if (a.amax_c) {
      float wmx = 0.f;
#pragma unroll
      for (int y = 0; y < TM; ++y) wmx = fmaxf(wmx, rmx[y]);          // (rows past M hold zeros: d = 0)
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o));
      if (lane == 0) e_wmax[wv] = wmx;
    }
    __syncthreads();
    if (t < 128) a.top_part[static_cast<int64_t>(blockIdx.x) * (a.N + 2) + t] = e_dw[t] + e_dw[128 + t];
    else if (t == 128) a.top_part[static_cast<int64_t>(blockIdx.x) * (a.N + 2) + a.N] = e_sc[1] + e_sc[3];
    else if (t == 129) a.top_part[static_cast<int64_t>(blockIdx.x) * (a.N + 2) + a.N + 1] = e_sc[0] + e_sc[2];
    if (a.amax_c && t == 0) {
      float mx = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) mx = fmaxf(mx, e_wmax[w]);
      unsigned int* slot = reinterpret_cast<unsigned int*>(a.amax_c) + (blockIdx.x & (MI_AMAX_SLOTS - 1));
      const unsigned int bits = __float_as_uint(mx);
      if (bits > *reinterpret_cast<volatile unsigned int*>(slot)) atomicMax(slot, bits);
    }
    // ---- dX as planes: row exponent from the row's abs-max over the 128 columns (tail.hip's conversion: a positive value keeps
    // a positive high part)
#pragma unroll
    for (int y = 0; y < TM; ++y) {
      const int ml = wm * 32 * TM + y * 32 + i;
      const int m = m0 + ml;
      const float mx = fmaxf(fmaxf(e_part[ml], e_part[BMt + ml]), fmaxf(e_part[2 * BMt + ml], e_part[3 * BMt + ml]));
      const int sx = pl_exp_for(mx);
      const float sc = pl_pow2(sx);
      if (m < a.M && h == 0 && wn == 0) a.c_exp[m] = sx;
      uint32_t ph[8], pq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float u0 = acc[0][y][2 * q] * sc, u1 = acc[0][y][2 * q + 1] * sc;
        const fl32x2 uu = {u0, u1};
        h16x2 hh = __builtin_convertvector(uu, h16x2);
        uint32_t hb = __builtin_bit_cast(uint32_t, hh);
        if (u0 > 0.f && (hb & 0xffffu) == 0u) hb |= 1u;
        if (u1 > 0.f && (hb >> 16) == 0u) hb |= 0x10000u;
        hh = __builtin_bit_cast(h16x2, hb);
        const fl32x2 rr2 = {u0 - static_cast<float>(hh[0]), u1 - static_cast<float>(hh[1])};
        const h16x2 ll = __builtin_convertvector(rr2, h16x2);
        ph[q] = hb;
        pq[q] = __builtin_bit_cast(uint32_t, ll);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        auto r1 = __builtin_amdgcn_permlane32_swap(ph[q], ph[q + 4], false, false);
        ph[q] = r1[0]; ph[q + 4] = r1[1];
        auto r2 = __builtin_amdgcn_permlane32_swap(pq[q], pq[q + 4], false, false);
        pq[q] = r2[0]; pq[q + 4] = r2[1];
      }
      char* d = wreg + (h * 32 + i) * 64;
      const int rot = (i >> 1) & 3;
      *reinterpret_cast<uint4*>(d + ((0 ^ rot) << 4)) = make_uint4(ph[0], ph[1], ph[4], ph[5]);
      *reinterpret_cast<uint4*>(d + ((1 ^ rot) << 4)) = make_uint4(ph[2], ph[3], ph[6], ph[7]);
      *reinterpret_cast<uint4*>(d + ((2 ^ rot) << 4)) = make_uint4(pq[0], pq[1], pq[4], pq[5]);
      *reinterpret_cast<uint4*>(d + ((3 ^ rot) << 4)) = make_uint4(pq[2], pq[3], pq[6], pq[7]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int off = q * 1024 + lane * 16;
        const int blk = off >> 11, row = (off >> 6) & 31, slot = (off >> 4) & 3;
        const uint4 v4 = *reinterpret_cast<const uint4*>(wreg + off);
        const int mm = m0 + wm * 32 * TM + y * 32 + row, ncol = nw + blk * 16;
        if (mm < a.M && ncol < a.N)
          *reinterpret_cast<uint4*>(a.Cp + (ncol >> 4) * a.bsc + static_cast<int64_t>(mm) * PL_ROWB + ((slot ^ ((row >> 1) & 3)) << 4)) = v4;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }

  // ---------------------------------------------------------------- epilogue
  pl_wait_vmcnt<0>();                 // the tail's re-issued loads
  __syncthreads();                    // every wave is done with the stage buffers: LDS is free
  float* e_fw = reinterpret_cast<float*>(smem);     // [BNt] 2^-s of the weight row (0 outside the matrix)
  float* e_bias = e_fw + BNt;                       // [BNt]
  float* e_rmax = e_bias + BNt;                     // [4][BMt] row abs-max per column group of waves
  float* e_wmax = e_rmax + 4 * BMt;                 // [8] per-wave abs-max (amax_c)
  // one private LDS region per wave for the result's way out (row-contiguous stores, below)
  constexpr int FS = TN * 128 + 16;                 // bytes per fp32 row of the wave's 32 x 32 TN sub-tile (+ pad: banks)
  constexpr int WREG = 32 * FS > TN * 4096 ? 32 * FS : TN * 4096;
  constexpr int WPR = BNt / 32;                      // mask words per tile row
  static_assert(8192 + 8 * WREG + BMt * WPR * 4 <= PL_NBUF * STAGE, "epilogue LDS");
  char* wreg = smem + 8192 + wv * WREG;
  uint32_t* e_bits = reinterpret_cast<uint32_t*>(smem + 8192 + 8 * WREG);     // [BMt][WPR] the tile's mask words (FWD, mask bits asked for)
  for (int n = t; n < BNt; n += PL_THREADS) {
    const int gn = n0 + n;
    const bool ok = HOT || gn < a.N;
    e_fw[n] = ok ? pl_pow2(-a.a_exp[ok ? gn : 0]) : 0.f;
    e_bias[n] = (EPI == PL_FWD && a.bias && ok) ? a.bias[gn] : 0.f;
  }
  __syncthreads();

  // The element loop.  Round 2's had the launch's (uniform) options as run-time tests per ELEMENT: hipcc emitted a scalar
  // branch or two per element, which also fenced every element's LDS reads of its column factors behind an s_waitcnt of
  // their own, a quarter-rate 32-bit multiply hash and an IEEE division behind a divergent branch — ~65 instructions per
  // output (in-kernel marks, tools/gemm_pl_timeline.py: the epilogue was 18 % of the layer-1 forward's time and 65 % of
  // the layer-2 data gradient's).  Now the options only select VALUES (no dropout: threshold 2^16 and divisor 1; no mask:
  // an all-positive mask word), the per-element code is straight-line, and the uniform branches that remain are per group
  // of four columns (hash / mask load / fp32 copy).
  const bool drop = EPI == PL_FWD && (HOT || a.keep_prob < 1.f);
  float* const Cf = HOT ? nullptr : a.C;
  const uint32_t thresh16 = drop ? mi_drop_thresh16(a.keep_prob) : 0x10000u;
  const uint64_t seed = a.seed + (a.st ? a.st->seed_term : 0ull);
  const int nw = wn * 32 * TN;                      // this wave's first column inside the tile
  // (the hash of a column pair: (col >> 1) * MUL = pair_base + a compile-time multiple of MUL)
  const uint32_t pair_base = static_cast<uint32_t>((n0 + nw + 4 * h) >> 1) * MI_DROP_PAIR_MUL;
  const bool bitmask = EPI == PL_DGRAD && (HOT || a.mbits != nullptr);
  const bool masked = !HOT && EPI == PL_DGRAD && a.mask != nullptr && !bitmask;
  const bool divide = drop || masked || bitmask;
  const float kd = divide ? a.keep_div : 1.f, kr = divide ? a.keep_rcp : 1.f;
  const float relu_floor = a.relu ? 0.f : -__builtin_inff();
  float rmx[TM];
#pragma unroll
  for (int y = 0; y < TM; ++y) {
    const int m = m0 + wm * 32 * TM + y * 32 + i;
    const bool mok = HOT || m < a.M;
    const int mc = mok ? m : a.M - 1;
    const float fx = pl_pow2(-a.b_exp[mc]);
    float mx = 0.f;
    const uint32_t rowkey = drop ? mi_drop_rowkey(seed, static_cast<uint32_t>(m)) : 0u;
    // the 4 columns' factors of a group: one 16-byte LDS read each, for group gi + 1 while group gi is computed
    auto col_of = [&](int gi) { return nw + (gi >> 2) * 32 + 8 * (gi & 3) + 4 * h; };
    float4 fw_n = *reinterpret_cast<const float4*>(e_fw + col_of(0));
    float4 b_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI == PL_FWD) b_n = *reinterpret_cast<const float4*>(e_bias + col_of(0));
    // DGRAD: the mask words (high plane of the stored activation, 4 x fp16 per group) come from GLOBAL memory: fetched
    // MK_AHEAD groups ahead — a group that loaded its own word and then waited for it paid a memory latency per group
    // (in-kernel marks: the layer-2 data gradient's epilogue was 73 k cycles against 43 k for its k loop).
    constexpr int MK_AHEAD = TN == 1 ? 2 : 4;           // (TN == 1 must stay within 128 registers: two workgroups per CU)
    uint2 mkq[MK_AHEAD];
    auto mask_word = [&](int gi) {
      const int gn = min(n0 + col_of(gi), a.N - 4);
      return *reinterpret_cast<const uint2*>(a.mask + (gn >> 4) * a.bsm + static_cast<int64_t>(mc) * PL_ROWB + (gn & 15) * 2);
    };
    if constexpr (EPI == PL_DGRAD) {
#pragma unroll
      for (int q = 0; q < MK_AHEAD; ++q) mkq[q] = (masked && q < TN * 4) ? mask_word(q) : make_uint2(0x00010001u, 0x00010001u);
    }
    // mask bits of this lane's example: the TN words of the wave's 32 TN columns (DGRAD: loaded here, one 4 TN-byte read;
    // FWD: collected below).  No bit mask: all ones.
    uint32_t mw[TN];
#pragma unroll
    for (int x = 0; x < TN; ++x) {
      mw[x] = EPI == PL_DGRAD ? 0xffffffffu : 0u;
      if constexpr (EPI == PL_DGRAD) {
        const int wi = ((n0 + nw) >> 5) + x;
        if (bitmask) mw[x] = (HOT || wi < a.mbld) ? a.mbits[static_cast<int64_t>(mc) * a.mbld + wi] : 0u;
      }
    }
#pragma unroll
    for (int gi = 0; gi < TN * 4; ++gi) {
      const int x = gi >> 2, g = gi & 3;
      const float4 fw4 = fw_n, b4 = b_n;
      if (gi + 1 < TN * 4) {
        fw_n = *reinterpret_cast<const float4*>(e_fw + col_of(gi + 1));
        if constexpr (EPI == PL_FWD) b_n = *reinterpret_cast<const float4*>(e_bias + col_of(gi + 1));
      }
      // FWD: the two pair hashes of the group's 4 columns (0: every 16-bit half is below 2^16 — kept);
      // DGRAD: the group's mask word (positive <=> active and kept; no mask: all positive)
      uint32_t w2[2] = {0u, 0u};
      if constexpr (EPI == PL_DGRAD) {
        // the group's four mask bits (columns 8 g + 4 h + j of tile x) as four "positive fp16" halves, ANDed into the
        // planes mask's word (no planes mask: all positive; no bit mask: all ones)
        const uint32_t nib = mw[x] >> (8 * g + 4 * h);
        const uint32_t b0 = (nib & 1u) | ((nib & 2u) << 15), b1 = ((nib >> 2) & 1u) | ((nib & 8u) << 13);
        w2[0] = bitmask ? b0 : mkq[gi % MK_AHEAD].x; w2[1] = bitmask ? b1 : mkq[gi % MK_AHEAD].y;
        if (masked && gi + MK_AHEAD < TN * 4) mkq[gi % MK_AHEAD] = mask_word(gi + MK_AHEAD);
      } else if (drop) {
        const uint32_t pt = pair_base + static_cast<uint32_t>((x * 32 + 8 * g) >> 1) * MI_DROP_PAIR_MUL;
        w2[0] = mi_drop_pairhash(rowkey, pt);
        w2[1] = mi_drop_pairhash(rowkey, pt + MI_DROP_PAIR_MUL);
      }
      // (columns past N: their weight-row factor e_fw is 0 and their bias 0, so the value is exactly 0)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        const float fwj = j == 0 ? fw4.x : j == 1 ? fw4.y : j == 2 ? fw4.z : fw4.w;
        float v = (acc[x][y][r] * fx) * fwj;
        const uint32_t bits16 = (j & 1) ? (w2[j >> 1] >> 16) : (w2[j >> 1] & 0xffffu);
        if constexpr (EPI == PL_FWD) {
          v += j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
          v = fmaxf(v, relu_floor);
          v = bits16 < thresh16 ? mi_div_const(v, kd, kr) : 0.f;
        } else {
          // positive fp16 (sign clear, not zero): the unit was active and kept
          v = (bits16 - 1u) < 0x7fffu ? mi_div_const(v, kd, kr) : 0.f;
        }
        acc[x][y][r] = v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(acc[x][y][4 * g + j]));
      if (Cf)
        *reinterpret_cast<float4*>(wreg + i * FS + (x * 32 + 8 * g + 4 * h) * 4) =
            make_float4(acc[x][y][4 * g], acc[x][y][4 * g + 1], acc[x][y][4 * g + 2], acc[x][y][4 * g + 3]);
    }
    if (Cf) {
      // the wave's 32 rows x 32 TN columns went through its own LDS region: read them back row by row, so that
      // a store instruction writes whole 128 TN-byte row segments instead of 64 scattered 16-byte pieces
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int LPR_ = TN * 8, RPI = 64 / LPR_;                  // lanes per row, rows per instruction
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + lane / LPR_, c16 = lane % LPR_;
        const float4 v4 = *reinterpret_cast<const float4*>(wreg + row * FS + c16 * 16);
        const int mm = m0 + wm * 32 * TM + y * 32 + row, nn = n0 + nw + c16 * 4;
        if (mm < a.M && nn < a.N) *reinterpret_cast<float4*>(Cf + static_cast<int64_t>(mm) * a.ldc + nn) = v4;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (EPI == PL_FWD) {
      if (HOT || a.mbits) {
        // one bit per output, from the finished values (+2.5 us on the 512 -> 256 forward, +1 us on 256 -> 128; the data
        // gradients that read the bits instead of the activation's planes gain 25-30 and 2 us: tools/mlp_tail_bench.py)
#pragma unroll
        for (int x = 0; x < TN; ++x) {
          uint32_t lo8 = 0u, hi8 = 0u;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            lo8 |= (acc[x][y][r] > 0.f ? 1u : 0u) << (8 * (r >> 2) + (r & 3));
            hi8 |= (acc[x][y][8 + r] > 0.f ? 1u : 0u) << (8 * (r >> 2) + (r & 3));
          }
          mw[x] = lo8 | (hi8 << 16);
        }
        // lane h holds the bits of columns 8 g + 4 h + j at positions 8 g + j: shifted by 4 h and ORed over the lane pair they
        // are the tile's 32-bit word.  Into LDS as [row][word]; the workgroup stores the tile's words as whole lines below.
#pragma unroll
        for (int x = 0; x < TN; ++x) {
          uint32_t wbits = mw[x] << (4 * h);
          wbits |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(wbits), 32));
          if (h == 0) e_bits[(wm * 32 * TM + y * 32 + i) * WPR + wn * TN + x] = wbits;
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));               // the lane pair that shares this example
    rmx[y] = mx;
    if (h == 0) e_rmax[wn * BMt + wm * 32 * TM + y * 32 + i] = mx;
  }
  if (a.amax_c) {
    float wmx = 0.f;
#pragma unroll
    for (int y = 0; y < TM; ++y) wmx = fmaxf(wmx, (m0 + wm * 32 * TM + y * 32 + i < a.M) ? rmx[y] : 0.f);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o));
    if (lane == 0) e_wmax[wv] = wmx;
  }
  PL_MARK(4);                          // element loops done (this wave)
  __syncthreads();
  if constexpr (EPI == PL_FWD) {
    if (HOT || a.mbits) {
      for (int idx = t; idx < BMt * WPR; idx += PL_THREADS) {
        const int row = idx / WPR, wi = (n0 >> 5) + idx % WPR;
        if (HOT || (m0 + row < a.M && wi < a.mbld)) a.mbits[static_cast<int64_t>(m0 + row) * a.mbld + wi] = e_bits[idx];
      }
    }
  }
  if (a.amax_c && t == 0) {
    float mx = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mx = fmaxf(mx, e_wmax[w]);
    unsigned int* slot = reinterpret_cast<unsigned int*>(a.amax_c) + (blockIdx.x & (MI_AMAX_SLOTS - 1));
    const unsigned int bits = __float_as_uint(mx);
    if (bits > *reinterpret_cast<volatile unsigned int*>(slot)) atomicMax(slot, bits);
  }
  if (!HOT && !a.Cp) {
#ifdef MI_PL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PL_MARK(3);
    return;
  }

  // ---- the result as planes: row exponent from the row's abs-max over ALL columns (tiles_n == 1) ----
#pragma unroll
  for (int y = 0; y < TM; ++y) {
    const int ml = wm * 32 * TM + y * 32 + i;
    const int m = m0 + ml;
    const bool mok = HOT || m < a.M;
    const float mx = fmaxf(fmaxf(e_rmax[ml], e_rmax[BMt + ml]), fmaxf(e_rmax[2 * BMt + ml], e_rmax[3 * BMt + ml]));
    const int s = pl_exp_for(mx);
    const float sc = pl_pow2(s);
    if (mok && h == 0 && wn == 0) a.c_exp[m] = s;
#pragma unroll
    for (int x = 0; x < TN; ++x) {
      uint32_t ph[8], pq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float u0 = acc[x][y][2 * q] * sc, u1 = acc[x][y][2 * q + 1] * sc;
        const fl32x2 uu = {u0, u1};
        h16x2 hh = __builtin_convertvector(uu, h16x2);                 // v_cvt_pk_f16_f32 (RNE)
        uint32_t hb = __builtin_bit_cast(uint32_t, hh);
        if constexpr (EPI == PL_FWD) {
          // a positive value must stay positive in the high plane: the data gradient's relu/dropout mask
          // reads "hi > 0" (only values below 2^-39 of the row maximum round to zero at all): high half =
          // max(high half, u > 0) as ONE packed unsigned maximum — positive fp16 order like their bit patterns, and a
          // negative half (sign bit set) is above 1 as an unsigned number, so it stays what it is.
          const uint32_t nz = (u0 > 0.f ? 1u : 0u) | (u1 > 0.f ? 0x10000u : 0u);
          typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
          const u16x2 mxv = __builtin_elementwise_max(__builtin_bit_cast(u16x2, hb), __builtin_bit_cast(u16x2, nz));
          hb = __builtin_bit_cast(uint32_t, mxv);
          hh = __builtin_bit_cast(h16x2, hb);
        }
        const fl32x2 rr2 = {u0 - static_cast<float>(hh[0]), u1 - static_cast<float>(hh[1])};
        const h16x2 ll = __builtin_convertvector(rr2, h16x2);
        ph[q] = hb;
        pq[q] = __builtin_bit_cast(uint32_t, ll);
      }
      // lane h = 0 collects columns 0..15 of the 32-column tile, lane h = 1 columns 16..31
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        auto r1 = __builtin_amdgcn_permlane32_swap(ph[q], ph[q + 4], false, false);
        ph[q] = r1[0]; ph[q + 4] = r1[1];
        auto r2 = __builtin_amdgcn_permlane32_swap(pq[q], pq[q + 4], false, false);
        pq[q] = r2[0]; pq[q + 4] = r2[1];
      }
      // this lane's 64 bytes (row i of 16-column block 2 x + h) into the wave's LDS region as [block][row][64 B],
      // the four 16-byte pieces rotated by the row so that the 8 lanes of a write group spread over the banks
      char* d = wreg + ((2 * x + h) * 32 + i) * 64;
      const int rot = (i >> 1) & 3;
      *reinterpret_cast<uint4*>(d + ((0 ^ rot) << 4)) = make_uint4(ph[0], ph[1], ph[4], ph[5]);
      *reinterpret_cast<uint4*>(d + ((1 ^ rot) << 4)) = make_uint4(ph[2], ph[3], ph[6], ph[7]);
      *reinterpret_cast<uint4*>(d + ((2 ^ rot) << 4)) = make_uint4(pq[0], pq[1], pq[4], pq[5]);
      *reinterpret_cast<uint4*>(d + ((3 ^ rot) << 4)) = make_uint4(pq[2], pq[3], pq[6], pq[7]);
    }
    // read the region back linearly: one instruction = 16 rows x 64 B of one block = 1 KiB contiguous in the
    // k-block-major result
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < TN * 4; ++q) {
      const int off = q * 1024 + lane * 16;
      const int blk = off >> 11, row = (off >> 6) & 31, slot = (off >> 4) & 3;
      const uint4 v4 = *reinterpret_cast<const uint4*>(wreg + off);
      const int mm = m0 + wm * 32 * TM + y * 32 + row, ncol = n0 + nw + blk * 16;
      if constexpr (HOT) {
        // a uniform block base + ONE 32-bit lane offset + a compile-time constant (the planes of a hot launch are below 2 GB):
        // a ds_read and a store per 1 KiB; the general form's 64-bit address of every store was ten instructions
        const uint32_t lane_off = static_cast<uint32_t>(m0 + wm * 32 * TM + (lane >> 2)) * PL_ROWB + (((lane & 3) ^ ((lane >> 3) & 3)) << 4);
        char* const blk_base = a.Cp + static_cast<int64_t>(((n0 + nw) >> 4) + (q >> 1)) * a.bsc;
        *reinterpret_cast<uint4*>(blk_base + (lane_off + static_cast<uint32_t>(y * 32 + (q & 1) * 16) * PL_ROWB)) = v4;
      } else if (mm < a.M && ncol < a.N) {
        *reinterpret_cast<uint4*>(a.Cp + (ncol >> 4) * a.bsc + static_cast<int64_t>(mm) * PL_ROWB + ((slot ^ ((row >> 1) & 3)) << 4)) = v4;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  PL_MARK(5);                          // planes converted, stores issued
#ifdef MI_PL_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  PL_MARK(3);
}

// ---- fp32 rows -> planes (weights once per step; any buffer no kernel here produced) ---------------
// One wave per output row.  transpose: the source is [K][rows] row-major and output row r is its column r
// (the weights of the forward pass: W[k][n] -> rows n).
__global__ __launch_bounds__(256) void split_rows_k(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                    int transpose, char* __restrict__ out, int64_t ldo_b,
                                                    int32_t* __restrict__ row_exp, float* __restrict__ amax_out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  float mx = 0.f;
  if (r < rows) {
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(transpose ? X[static_cast<int64_t>(k) * ldx + r] : X[r * ldx + k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (r < rows) {
    const int s = pl_exp_for(mx);
    const float sc = pl_pow2(s);
    if (lane == 0) row_exp[r] = s;
    const int K16 = (K + 15) & ~15;
    char* d = out + r * PL_ROWB;
    for (int k = lane; k < K16; k += 64) {
      const float x = k < K ? (transpose ? X[static_cast<int64_t>(k) * ldx + r] : X[r * ldx + k]) : 0.f;
      const float u = x * sc;
      _Float16 hi = static_cast<_Float16>(u);
      if (u > 0.f && hi == static_cast<_Float16>(0.f)) hi = __builtin_bit_cast(_Float16, static_cast<unsigned short>(1));
      const _Float16 lo = static_cast<_Float16>(u - static_cast<float>(hi));
      _Float16* e = reinterpret_cast<_Float16*>(d + (k >> 4) * ldo_b);
      e[k & 15] = hi;
      e[16 + (k & 15)] = lo;
    }
  }
  if (amax_out) {                       // (every thread of the block reaches this)
    __shared__ float part[4];
    if (lane == 0) part[threadIdx.x >> 6] = r < rows ? mx : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m4 = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
      unsigned int* slot = reinterpret_cast<unsigned int*>(amax_out) + (blockIdx.x & (MI_AMAX_SLOTS - 1));
      const unsigned int bits = __float_as_uint(m4);
      if (bits > *reinterpret_cast<volatile unsigned int*>(slot)) atomicMax(slot, bits);
    }
  }
}

// The same split for K % 16 == 0 and 16-byte aligned rows: a lane owns whole 16-k blocks (64 B in, 64 B out as
// four 16-byte stores) and LPR lanes share a row, so that a wave works on 64 / LPR rows — the one-wave-per-row
// kernel above leaves most lanes idle at K = 128 and writes 2-byte pieces.
template <int LPR>
__global__ __launch_bounds__(256) void split_rows_blk_k(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                        char* __restrict__ out, int64_t ldo_b,
                                                        int32_t* __restrict__ row_exp, float* __restrict__ amax_out) {
  const int t = threadIdx.x, l = t & (LPR - 1);
  const int64_t r = (static_cast<int64_t>(blockIdx.x) * 256 + t) / LPR;
  const int nkb = K >> 4;
  const bool on = r < rows;
  const float* src = X + (on ? r : 0) * ldx;
  // (the first block of a lane stays in registers: K <= 16 LPR, the usual case, reads the row once)
  float4 v0[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                  make_float4(0.f, 0.f, 0.f, 0.f)};
  float mx = 0.f;
  if (on)
    for (int kb = l; kb < nkb; kb += LPR) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(src + kb * 16 + q * 4);
        if (kb == l) v0[q] = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
#pragma unroll
  for (int o = LPR >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (on) {
    const int s = pl_exp_for(mx);
    const float sc = pl_pow2(s);
    if (l == 0) row_exp[r] = s;
    for (int kb = l; kb < nkb; kb += LPR) {
      uint32_t ph[8], pq[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = kb == l ? v0[q] : *reinterpret_cast<const float4*>(src + kb * 16 + q * 4);
        const float u[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const fl32x2 uu = {u[2 * e], u[2 * e + 1]};
          h16x2 hh = __builtin_convertvector(uu, h16x2);
          uint32_t hb = __builtin_bit_cast(uint32_t, hh);
          if (uu[0] > 0.f && (hb & 0xffffu) == 0u) hb |= 1u;          // positive stays positive (mask reads hi > 0)
          if (uu[1] > 0.f && (hb >> 16) == 0u) hb |= 0x10000u;
          hh = __builtin_bit_cast(h16x2, hb);
          const fl32x2 rr2 = {uu[0] - static_cast<float>(hh[0]), uu[1] - static_cast<float>(hh[1])};
          ph[2 * q + e] = hb;
          pq[2 * q + e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr2, h16x2));
        }
      }
      uint4* d = reinterpret_cast<uint4*>(out + kb * ldo_b + r * PL_ROWB);
      d[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      d[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
      d[2] = make_uint4(pq[0], pq[1], pq[2], pq[3]);
      d[3] = make_uint4(pq[4], pq[5], pq[6], pq[7]);
    }
  }
  if (amax_out) {                       // (every thread of the block reaches this)
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) part[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
      const float m4 = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
      unsigned int* slot = reinterpret_cast<unsigned int*>(amax_out) + (blockIdx.x & (MI_AMAX_SLOTS - 1));
      const unsigned int bits = __float_as_uint(m4);
      if (bits > *reinterpret_cast<volatile unsigned int*>(slot)) atomicMax(slot, bits);
    }
  }
}

// ---- data gradient of the N = 1 logits layer (deep_fm.py:108 backward) written straight as planes ----------------
// dX[m][k] = dY[m] * W[k], masked by the stored activation (ReLU: kept where Xact > 0, divided by keep_prob) — the
// arithmetic of gemm.hip's gemv_dgrad_k, hence the same bits — then the split of split_rows_blk_k on the values
// still in registers: the fp32 matrix is neither written nor read back (it has no other reader once the weight
// gradient takes planes).
template <int LPR>
__global__ __launch_bounds__(256) void vec_dgrad_planes_k(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W,
                                                          const float* __restrict__ Xact, int64_t ldxa, float keep_div, float keep_rcp,
                                                          int64_t rows, int K, int rows_per_block, float* __restrict__ dX,
                                                          int64_t lddx, char* __restrict__ out, int64_t ldo_b,
                                                          int32_t* __restrict__ row_exp, float* __restrict__ amax_out,
                                                          const uint32_t* __restrict__ mbits, int64_t mbld) {
  // LPR lanes share a row; a lane owns the float4 groups l, l + LPR, ... (adjacent lanes read adjacent 16 bytes).
  // The plane pieces of the block's rows go through LDS as [16-k block][row][64 B] and leave as ONE contiguous run
  // per 16-k block (consecutive lanes, 16 bytes each): written piece by piece from the lanes that computed them,
  // a store instruction would scatter 32-byte fragments over K / 16 regions 64 M bytes apart (measured 3x slower).
  extern __shared__ __attribute__((aligned(16))) char stage[];           // [K / 16][rows_per_block][64]
  const int t = threadIdx.x, l = t & (LPR - 1);
  const int nq = K >> 2, nkb = K >> 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int nrows = static_cast<int>(min(static_cast<int64_t>(rows_per_block), rows - r0));
  float bmx = 0.f;
  for (int rl = t / LPR; rl < rows_per_block; rl += 256 / LPR) {            // (uniform trip count: shuffles below)
    const bool on = rl < nrows;
    const int64_t r = r0 + (on ? rl : 0);
    const float g = dY[r * lddy];
    auto value4 = [&](int k) {
      const float4 w = *reinterpret_cast<const float4*>(W + k);
      float4 v = make_float4(g * w.x, g * w.y, g * w.z, g * w.w);
      if (mbits) {          // the forward pass's one-bit mask instead of the stored activation (1/32 of the bytes): same decisions
        const uint32_t nib = mbits[r * mbld + (k >> 5)] >> (k & 31);
        v.x = (nib & 1u) ? mi_div_const(v.x, keep_div, keep_rcp) : 0.f; v.y = (nib & 2u) ? mi_div_const(v.y, keep_div, keep_rcp) : 0.f;
        v.z = (nib & 4u) ? mi_div_const(v.z, keep_div, keep_rcp) : 0.f; v.w = (nib & 8u) ? mi_div_const(v.w, keep_div, keep_rcp) : 0.f;
      } else if (Xact) {
        const float4 x = *reinterpret_cast<const float4*>(Xact + r * ldxa + k);
        // (mi_div_const: the bits of '/', 3 instructions; the same masked division as gemm.hip's gemv_dgrad_k)
        v.x = x.x > 0.f ? mi_div_const(v.x, keep_div, keep_rcp) : 0.f; v.y = x.y > 0.f ? mi_div_const(v.y, keep_div, keep_rcp) : 0.f;
        v.z = x.z > 0.f ? mi_div_const(v.z, keep_div, keep_rcp) : 0.f; v.w = x.w > 0.f ? mi_div_const(v.w, keep_div, keep_rcp) : 0.f;
      }
      return v;
    };
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float mx = 0.f;
    if (on)
      for (int q = l; q < nq; q += LPR) {
        const float4 v = value4(4 * q);
        if (q == l) v0 = v;
        if (dX) *reinterpret_cast<float4*>(dX + r * lddx + 4 * q) = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
#pragma unroll
    for (int o = LPR >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    bmx = fmaxf(bmx, mx);
    if (on) {
      const int s = pl_exp_for(mx);
      const float sc = pl_pow2(s);
      if (l == 0) row_exp[r] = s;
      for (int q = l; q < nq; q += LPR) {
        const float4 v = q == l ? v0 : value4(4 * q);
        const float u[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
        uint32_t ph[2], pq[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const fl32x2 uu = {u[2 * e], u[2 * e + 1]};
          h16x2 hh = __builtin_convertvector(uu, h16x2);
          uint32_t hb = __builtin_bit_cast(uint32_t, hh);
          if (uu[0] > 0.f && (hb & 0xffffu) == 0u) hb |= 1u;          // positive stays positive (mask reads hi > 0)
          if (uu[1] > 0.f && (hb >> 16) == 0u) hb |= 0x10000u;
          hh = __builtin_bit_cast(h16x2, hb);
          const fl32x2 rr2 = {uu[0] - static_cast<float>(hh[0]), uu[1] - static_cast<float>(hh[1])};
          ph[e] = hb;
          pq[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr2, h16x2));
        }
        char* d = stage + ((q >> 2) * rows_per_block + rl) * PL_ROWB + (q & 3) * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(ph[0], ph[1]);
        *reinterpret_cast<uint2*>(d + 32) = make_uint2(pq[0], pq[1]);
      }
    }
  }
  __syncthreads();
  const int run16 = nrows * 4;                           // 16-byte pieces of one block's run
  for (int kb = 0; kb < nkb; ++kb) {
    const uint4* src = reinterpret_cast<const uint4*>(stage + kb * rows_per_block * PL_ROWB);
    uint4* dst = reinterpret_cast<uint4*>(out + kb * ldo_b + r0 * PL_ROWB);
    for (int p = t; p < run16; p += 256) dst[p] = src[p];
  }
  if (amax_out) {                       // (every thread of the block reaches this)
    __shared__ float part[4];
    float mx = bmx;
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) part[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
      const float m4 = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
      unsigned int* slot = reinterpret_cast<unsigned int*>(amax_out) + (blockIdx.x & (MI_AMAX_SLOTS - 1));
      const unsigned int bits = __float_as_uint(m4);
      if (bits > *reinterpret_cast<volatile unsigned int*>(slot)) atomicMax(slot, bits);
    }
  }
}

// ---- the transposed split (weights of the forward pass: X is [K][rows], output row r = column r of X) in
// one launch: a block owns 32 output rows; pass 1 their abs-max over all k (coalesced 128-B row pieces),
// pass 2 64-k tiles transposed through LDS and written as planes (the tiles come back from L2).
__global__ __launch_bounds__(256) void split_t_k(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                 char* __restrict__ out, int64_t ldo_b, int32_t* __restrict__ row_exp,
                                                 float* __restrict__ amax_out) {
  __shared__ float tile[64][33];
  __shared__ float cmax[8][32];
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 32;
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const bool cok = r0 + c < rows;
  float mx = 0.f;
  if (cok)
    for (int k = q; k < K; k += 8) mx = fmaxf(mx, fabsf(X[static_cast<int64_t>(k) * ldx + r0 + c]));
  cmax[q][c] = mx;
  __syncthreads();
  const int rl = threadIdx.x >> 2, kb = (threadIdx.x & 3) * 16;     // (threads < 128) output row in the block, 16-k block
  float rmx = 0.f;
  if (threadIdx.x < 128) {
#pragma unroll
    for (int j = 0; j < 8; ++j) rmx = fmaxf(rmx, cmax[j][rl]);
  }
  const int sx = pl_exp_for(rmx);
  const float sc = pl_pow2(sx);
  const int64_t r = r0 + rl;
  if (threadIdx.x < 128 && (threadIdx.x & 3) == 0 && r < rows) {
    row_exp[r] = sx;
    if (amax_out && rmx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(amax_out) + (r & (MI_AMAX_SLOTS - 1)), __float_as_uint(rmx));
  }
  const int K16 = (K + 15) & ~15;
  for (int k0 = 0; k0 < K16; k0 += 64) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + q * 8 + j;
      tile[q * 8 + j][c] = (k < K && cok) ? X[static_cast<int64_t>(k) * ldx + r0 + c] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 128 && r < rows && k0 + kb < K16) {
      uint32_t ph[8], pq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u0 = tile[kb + 2 * j][rl] * sc, u1 = tile[kb + 2 * j + 1][rl] * sc;
        const fl32x2 uu = {u0, u1};
        h16x2 hh = __builtin_convertvector(uu, h16x2);
        uint32_t hb = __builtin_bit_cast(uint32_t, hh);
        if (u0 > 0.f && (hb & 0xffffu) == 0u) hb |= 1u;
        if (u1 > 0.f && (hb >> 16) == 0u) hb |= 0x10000u;
        hh = __builtin_bit_cast(h16x2, hb);
        const fl32x2 dd = {u0 - static_cast<float>(hh[0]), u1 - static_cast<float>(hh[1])};
        ph[j] = hb;
        pq[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(dd, h16x2));
      }
      char* d = out + ((k0 + kb) >> 4) * ldo_b + r * PL_ROWB;
      *reinterpret_cast<uint4*>(d) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      *reinterpret_cast<uint4*>(d + 16) = make_uint4(ph[4], ph[5], ph[6], ph[7]);
      *reinterpret_cast<uint4*>(d + 32) = make_uint4(pq[0], pq[1], pq[2], pq[3]);
      *reinterpret_cast<uint4*>(d + 48) = make_uint4(pq[4], pq[5], pq[6], pq[7]);
    }
  }
}

// ---- all weight planes of a training step in ONE launch ---------------------------------------------
// Every hidden layer's kernel W[K][N] is needed twice: as stored (rows = K: the data gradient's operand) and
// transposed (rows = N: the forward pass's).  One exponent for the whole parameter block, from its abs-max
// (mi_absmax): no per-row reduction, so a 64 x 64 tile of W goes through LDS once and leaves as both.
// (Weights more than 2^-17 below the block's largest lose low bits: their products are that far below the
// largest terms of any dot product they enter.  The per-ROW exponents that matter are the examples'.)
struct WJob { int64_t off; int K, N; char* pw; int64_t bsw; int32_t* ew; char* pt; int64_t bst; int32_t* et; int tile0; };
struct WJobs { WJob j[MI_MAX_WEIGHT_JOBS]; int n; };

__global__ __launch_bounds__(256) void split_weights_k(const float* __restrict__ dense, const WJobs jobs,
                                                       const float* __restrict__ amax) {
  __shared__ float tile[64][65];
  int ji = 0;
#pragma unroll
  for (int q = 1; q < MI_MAX_WEIGHT_JOBS; ++q)
    if (q < jobs.n && static_cast<int>(blockIdx.x) >= jobs.j[q].tile0) ji = q;
  const WJob& jb = jobs.j[ji];
  const int tn = (jb.N + 63) >> 6;
  const int tl = blockIdx.x - jb.tile0;
  const int k0 = (tl / tn) * 64, n0 = (tl % tn) * 64;
  float m = 0.f;
  if (threadIdx.x < MI_AMAX_SLOTS) m = amax[threadIdx.x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  m = __shfl(m, 0);                                  // (wave 0 holds it; the other waves get it through LDS)
  __shared__ float s_m;
  if (threadIdx.x == 0) s_m = m;
  const float* W = dense + jb.off;
  const int c = threadIdx.x & 63, q4 = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + q4 * 16 + j, n = n0 + c;
    tile[q4 * 16 + j][c] = (k < jb.K && n < jb.N) ? W[static_cast<int64_t>(k) * jb.N + n] : 0.f;
  }
  __syncthreads();
  const int sx = pl_exp_for(s_m);
  const float sc = pl_pow2(sx);
  const int rl = threadIdx.x >> 2, cb = (threadIdx.x & 3) * 16;
  auto emit = [&](bool transposed, char* d) {
    uint32_t ph[8], pq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u0 = (transposed ? tile[cb + 2 * j][rl] : tile[rl][cb + 2 * j]) * sc;
      const float u1 = (transposed ? tile[cb + 2 * j + 1][rl] : tile[rl][cb + 2 * j + 1]) * sc;
      const fl32x2 uu = {u0, u1};
      const h16x2 hh = __builtin_convertvector(uu, h16x2);
      const fl32x2 dd = {u0 - static_cast<float>(hh[0]), u1 - static_cast<float>(hh[1])};
      ph[j] = __builtin_bit_cast(uint32_t, hh);
      pq[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(dd, h16x2));
    }
    *reinterpret_cast<uint4*>(d) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    *reinterpret_cast<uint4*>(d + 16) = make_uint4(ph[4], ph[5], ph[6], ph[7]);
    *reinterpret_cast<uint4*>(d + 32) = make_uint4(pq[0], pq[1], pq[2], pq[3]);
    *reinterpret_cast<uint4*>(d + 48) = make_uint4(pq[4], pq[5], pq[6], pq[7]);
  };
  if (jb.pw && k0 + rl < jb.K && n0 + cb < ((jb.N + 15) & ~15)) {          // rows = k, columns = n
    emit(false, jb.pw + ((n0 + cb) >> 4) * jb.bsw + static_cast<int64_t>(k0 + rl) * PL_ROWB);
    if (n0 + cb == 0) jb.ew[k0 + rl] = sx;
  }
  if (jb.pt && n0 + rl < jb.N && k0 + cb < ((jb.K + 15) & ~15)) {          // rows = n, columns = k
    emit(true, jb.pt + ((k0 + cb) >> 4) * jb.bst + static_cast<int64_t>(n0 + rl) * PL_ROWB);
    if (k0 + cb == 0) jb.et[n0 + rl] = sx;
  }
}

// planes -> fp32 (tests, summaries)
__global__ __launch_bounds__(256) void merge_rows_k(const char* __restrict__ in, int64_t ldi_b, const int32_t* __restrict__ row_exp,
                                                    int64_t rows, int K, float* __restrict__ X, int64_t ldx) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t r = idx / K;
  if (r >= rows) return;
  const int k = static_cast<int>(idx - r * K);
  const _Float16* d = reinterpret_cast<const _Float16*>(in + (k >> 4) * ldi_b + r * PL_ROWB);
  const float v = static_cast<float>(d[k & 15]) + static_cast<float>(d[16 + (k & 15)]);
  X[r * ldx + k] = v * pl_pow2(-row_exp[r]);
}

// The proof obligation of mi_div_const (common.h): the bits of x / d for every fp32 x.  One thread per bit pattern.
// out[0] += mismatches with 2^-100 <= |x| <= 2^100 (must be 0), out[1] += all mismatches (tiny x: the quotient's residual
// is no longer exactly representable; huge x: x r overflows where x / d would not; infinities: inf - inf).
__global__ __launch_bounds__(256) void selftest_div_k(float d, float r, uint32_t first_bits, int64_t count, unsigned long long* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  bool bad = false, in_range = false;
  if (i < count) {
    const float x = __uint_as_float(first_bits + static_cast<uint32_t>(i));
    const float want = x / d, got = mi_div_const(x, d, r);
    bad = __float_as_uint(want) != __float_as_uint(got) && !(want != want && got != got);   // (NaN payloads aside)
    const uint32_t ax = __float_as_uint(x) & 0x7fffffffu;
    in_range = ax >= 0x0D800000u && ax <= 0x71800000u;                                         // 2^-100 .. 2^100
  }
  const unsigned long long b_in = __ballot(bad && in_range), b_all = __ballot(bad);
  if ((threadIdx.x & 63) == 0) {
    if (b_in) atomicAdd(out, static_cast<unsigned long long>(__popcll(b_in)));
    if (b_all) atomicAdd(out + 1, static_cast<unsigned long long>(__popcll(b_all)));
  }
}

// out[k] = the sum of the workgroups' partials of gemm_pl_k<.., PL_TOP> (tail.hip's tail_fold_k: 16 columns x 16 slices per
// workgroup, the slices folded in order): the logits layer's dW [N], the sum of d (db and / or d_sum), the loss
__global__ __launch_bounds__(256) void top_fold_k(const float* __restrict__ part, int nparts, int N, float* __restrict__ dW,
                                                  float* __restrict__ db, float* __restrict__ d_sum, float* __restrict__ loss) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + c;
  float acc = 0.f;
  if (k < N + 2) {
#pragma unroll 8
    for (int q = sl; q < nparts; q += 16) acc += part[static_cast<int64_t>(q) * (N + 2) + k];
  }
  red[sl][c] = acc;
  __syncthreads();
  if (sl == 0 && k < N + 2) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][c];
    if (k < N) dW[k] = v;
    else if (k == N) { if (db) db[0] = v; if (d_sum) d_sum[0] = v; }
    else if (loss) loss[0] = v;
  }
}

bool planes_ok(const mi_planes_t* p, int64_t rows, int K) {
  return p && p->data && p->row_exp && mi::aligned16(p->data) && p->blk_stride >= rows * PL_ROWB && (p->blk_stride & 63) == 0 && rows >= 0;
}

template <int EPI>
int32_t launch_pl(PlArgs& a, hipStream_t st, const char* what) {
  // column tile: the narrowest that holds N (planes out needs all of N in one tile); row tile: 256 rows for
  // the narrower column tiles (same accumulator budget), 128 for 512 columns
  int tn = a.N <= 128 ? 1 : (a.N <= 256 ? 2 : 4);
  if (a.N > 512) tn = 2;                              // fp32 result only (layer-1 data gradient, 1664 columns): 256-column tiles.
                                                      // Round 2 took 128 columns (two workgroups per CU: 426 us against 467 and 520
                                                      // for 256 and 512); with round 3's epilogue 256 columns win although the 7th
                                                      // column tile is half empty — isolated 340-350 us against 370-380, in the step
                                                      // 0.482 against 0.509 ms for the three data gradients (in-kernel marks: the
                                                      // 128-column kernel's k loop runs its matrix pipes at 35 %, this one's at 60 %)
  if (const int v = mi::env_int("MI_PL_TILE", 0)) {   // tuning experiments (tools/gemm_pl_bench.py, the tools' build only)
    if ((v == 1 || v == 2 || v == 4) && (!a.Cp || a.N <= 128 * v)) tn = v;
  }
  // 256-row tiles for the narrower column tiles — unless that leaves CUs without a workgroup (a chunk of a multi-GPU
  // step, a small batch: M = 32768 gives 128 tiles of 256 rows for 256 CUs): then 128 rows, like the 512-column tile
  int tm = tn == 4 ? 2 : 4;
  if (tm == 4 && static_cast<int64_t>((a.N + 128 * tn - 1) / (128 * tn)) * ((a.M + 255) / 256) < 256) tm = 2;
  const int bn = 128 * tn, bm = 64 * tm;
  a.tiles_n = (a.N + bn - 1) / bn;
  const int64_t blocks = static_cast<int64_t>(a.tiles_n) * ((a.M + bm - 1) / bm);
  if (blocks <= 0 || blocks > INT32_MAX) {
    mi::set_error("%s: bad grid", what);
    return MI_ERR_INVALID;
  }
  if (a.Cp && a.tiles_n != 1) {
    mi::set_error("%s: a planes result needs N <= 512 (one column tile holds the whole row)", what);
    return MI_ERR_UNSUPPORTED;
  }
  const dim3 g(static_cast<unsigned>(blocks)), b(PL_THREADS);
  // the steady-state training call (see gemm_pl_k: HOT)
  const bool hot = a.Cp && !a.C && a.tiles_n == 1 && a.N == bn && a.M % bm == 0 && a.amax_c && a.bsc < (int64_t(1) << 31) &&
                   (EPI == PL_FWD ? (a.keep_prob < 1.f && a.mbits && a.mbld * 32 == a.N && a.bias)
                                  : (a.mbits && a.mbld * 32 == a.N));
  if (hot && tn == 1 && tm == 4) gemm_pl_k<1, 4, EPI, true><<<g, b, 0, st>>>(a);
  else if (hot && tn == 2 && tm == 4) gemm_pl_k<2, 4, EPI, true><<<g, b, 0, st>>>(a);
  // (not the 512-column forward — layer 1, at the chip's power limit either way: its fixed-option epilogue needs more than
  // 256 registers)
  else if (hot && tn == 4 && EPI == PL_DGRAD) gemm_pl_k<4, 2, PL_DGRAD, true><<<g, b, 0, st>>>(a);
  else if (tn == 1 && tm == 4) gemm_pl_k<1, 4, EPI><<<g, b, 0, st>>>(a);
  else if (tn == 1) gemm_pl_k<1, 2, EPI><<<g, b, 0, st>>>(a);
  else if (tn == 2 && tm == 4) gemm_pl_k<2, 4, EPI><<<g, b, 0, st>>>(a);
  else if (tn == 2) gemm_pl_k<2, 2, EPI><<<g, b, 0, st>>>(a);
  else gemm_pl_k<4, 2, EPI><<<g, b, 0, st>>>(a);
  MI_CHECK_LAUNCH(what);
  return MI_OK;
}

}  // namespace

extern "C" {

size_t mi_planes_bytes(int64_t rows, int32_t K) { return static_cast<size_t>((K + 15) >> 4) * static_cast<size_t>(rows) * PL_ROWB; }

int32_t mi_split_rows(const float* X, int64_t ldx, int64_t rows, int32_t K, int32_t transpose, const mi_planes_t* out,
                      float* amax_out, mi_stream_t stream) {
  MI_REQUIRE(rows >= 0 && K > 0 && X, "split_rows: rows=%lld K=%d", (long long)rows, K);
  if (rows == 0) return MI_OK;
  MI_REQUIRE(planes_ok(out, rows, K), "split_rows: output planes (16-byte aligned data, blk_stride >= 64 * rows and a multiple of 64, row_exp)");
  MI_REQUIRE(transpose ? ldx >= rows : ldx >= K, "split_rows: ldx=%lld", (long long)ldx);
  if (transpose) {
    const int64_t tb = mi::ceil_div(rows, 32);
    MI_REQUIRE(tb <= INT32_MAX, "split_rows: grid too large");
    split_t_k<<<dim3((unsigned)tb), dim3(256), 0, mi::as_stream(stream)>>>(X, ldx, rows, K, static_cast<char*>(out->data),
                                                                           out->blk_stride, out->row_exp, amax_out);
    MI_CHECK_LAUNCH("split_rows(transposed)");
    return MI_OK;
  }
  if ((K & 15) == 0 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0) {
    int lpr = 1;
    while (lpr < 64 && lpr < (K >> 4)) lpr <<= 1;
    const int64_t nb = mi::ceil_div(rows * lpr, 256);
    MI_REQUIRE(nb <= INT32_MAX, "split_rows: grid too large");
#define MI_SPLIT_BLK(L) split_rows_blk_k<L><<<dim3((unsigned)nb), dim3(256), 0, mi::as_stream(stream)>>>( \
        X, ldx, rows, K, static_cast<char*>(out->data), out->blk_stride, out->row_exp, amax_out)
    switch (lpr) {
      case 1: MI_SPLIT_BLK(1); break;
      case 2: MI_SPLIT_BLK(2); break;
      case 4: MI_SPLIT_BLK(4); break;
      case 8: MI_SPLIT_BLK(8); break;
      case 16: MI_SPLIT_BLK(16); break;
      case 32: MI_SPLIT_BLK(32); break;
      default: MI_SPLIT_BLK(64); break;
    }
#undef MI_SPLIT_BLK
    MI_CHECK_LAUNCH("split_rows(blocks)");
    return MI_OK;
  }
  const int64_t blocks = mi::ceil_div(rows, 4);
  MI_REQUIRE(blocks <= INT32_MAX, "split_rows: grid too large");
  split_rows_k<<<dim3((unsigned)blocks), dim3(256), 0, mi::as_stream(stream)>>>(
      X, ldx, rows, K, transpose, static_cast<char*>(out->data), out->blk_stride, out->row_exp, amax_out);
  MI_CHECK_LAUNCH("split_rows");
  return MI_OK;
}

int32_t mi_split_weights(const float* dense, const mi_weight_job_t* jobs, int32_t n_jobs, const float* amax,
                         mi_stream_t stream) {
  MI_REQUIRE(dense && jobs && amax && n_jobs > 0 && n_jobs <= MI_MAX_WEIGHT_JOBS, "split_weights: n_jobs=%d (1..%d)", n_jobs,
             MI_MAX_WEIGHT_JOBS);
  WJobs js{};
  js.n = n_jobs;
  int tiles = 0;
  for (int q = 0; q < n_jobs; ++q) {
    const mi_weight_job_t& u = jobs[q];
    MI_REQUIRE(u.K > 0 && u.N > 0 && u.offset >= 0 && (u.w.data || u.wt.data), "split_weights: job %d", q);
    MI_REQUIRE(!u.w.data || planes_ok(&u.w, u.K, u.N), "split_weights: job %d planes of W", q);
    MI_REQUIRE(!u.wt.data || planes_ok(&u.wt, u.N, u.K), "split_weights: job %d planes of W transposed", q);
    WJob& j = js.j[q];
    j.off = u.offset; j.K = u.K; j.N = u.N;
    j.pw = static_cast<char*>(u.w.data); j.bsw = u.w.blk_stride; j.ew = u.w.row_exp;
    j.pt = static_cast<char*>(u.wt.data); j.bst = u.wt.blk_stride; j.et = u.wt.row_exp;
    j.tile0 = tiles;
    tiles += ((u.K + 63) >> 6) * ((u.N + 63) >> 6);
  }
  split_weights_k<<<dim3((unsigned)tiles), dim3(256), 0, mi::as_stream(stream)>>>(dense, js, amax);
  MI_CHECK_LAUNCH("split_weights");
  return MI_OK;
}

int32_t mi_merge_rows(const mi_planes_t* in, int64_t rows, int32_t K, float* X, int64_t ldx, mi_stream_t stream) {
  MI_REQUIRE(rows >= 0 && K > 0 && X && ldx >= K, "merge_rows: rows=%lld K=%d", (long long)rows, K);
  if (rows == 0) return MI_OK;
  MI_REQUIRE(planes_ok(in, rows, K), "merge_rows: input planes");
  const int64_t blocks = mi::ceil_div(rows * K, 256);
  MI_REQUIRE(blocks <= INT32_MAX, "merge_rows: grid too large");
  merge_rows_k<<<dim3((unsigned)blocks), dim3(256), 0, mi::as_stream(stream)>>>(static_cast<const char*>(in->data), in->blk_stride,
                                                                                in->row_exp, rows, K, X, ldx);
  MI_CHECK_LAUNCH("merge_rows");
  return MI_OK;
}

int32_t mi_dense_fwd_planes(const mi_planes_t* X, const mi_planes_t* Wt, const float* bias, float* Y, int64_t ldy,
                            const mi_planes_t* Yp, int64_t M, int32_t N, int32_t K, int32_t relu, float keep_prob,
                            uint64_t seed, float* amax_out, uint32_t* mask_bits_out, int64_t mask_ld, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && M <= INT32_MAX && N > 0 && K > 0, "dense_fwd_planes: M=%lld N=%d K=%d", (long long)M, N, K);
  MI_REQUIRE(!mask_bits_out || mask_ld >= (N + 31) / 32, "dense_fwd_planes: mask_ld=%lld < ceil(N / 32)", (long long)mask_ld);
  if (M == 0) return MI_OK;
  MI_REQUIRE((K & 15) == 0 && (N & 15) == 0, "dense_fwd_planes: N=%d and K=%d must be multiples of 16 (use mi_dense_fwd)", N, K);
  MI_REQUIRE(planes_ok(X, M, K) && planes_ok(Wt, N, K), "dense_fwd_planes: operand planes");
  MI_REQUIRE(Y || Yp, "dense_fwd_planes: no output");
  MI_REQUIRE(!Y || (ldy >= N && (ldy & 3) == 0 && mi::aligned16(Y)), "dense_fwd_planes: Y leading dimension / alignment");
  MI_REQUIRE(!Yp || planes_ok(Yp, M, N), "dense_fwd_planes: output planes");
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dense_fwd_planes: keep_prob=%f", keep_prob);
  PlArgs a{};
  a.A = static_cast<const char*>(Wt->data); a.bsa = Wt->blk_stride; a.a_exp = Wt->row_exp;
  a.B = static_cast<const char*>(X->data); a.bsb = X->blk_stride; a.b_exp = X->row_exp;
  a.M = (int)M; a.N = N; a.K = K;
  a.C = Y; a.ldc = ldy;
  if (Yp) { a.Cp = static_cast<char*>(Yp->data); a.bsc = Yp->blk_stride; a.c_exp = Yp->row_exp; }
  a.bias = bias; a.relu = relu; a.keep_prob = keep_prob; a.keep_div = keep_prob; a.keep_rcp = 1.0f / keep_prob; a.seed = seed; a.st = mi::step_state();
  a.amax_c = amax_out;
  a.mbits = mask_bits_out; a.mbld = mask_ld;
  return launch_pl<PL_FWD>(a, mi::as_stream(stream), "dense_fwd_planes");
}

int32_t mi_dense_bwd_data_planes(const mi_planes_t* dY, const mi_planes_t* W, const mi_planes_t* Xact, float* dX,
                                 int64_t lddx, const mi_planes_t* dXp, int64_t M, int32_t N, int32_t K, float keep_prob,
                                 float* amax_out, const uint32_t* mask_bits, int64_t mask_ld, mi_stream_t stream) {
  // dX[M][K] = dY[M][N] * W[K][N]^T : output width K, reduction over N
  MI_REQUIRE(M >= 0 && M <= INT32_MAX && N > 0 && K > 0, "dense_bwd_data_planes: M=%lld N=%d K=%d", (long long)M, N, K);
  if (M == 0) return MI_OK;
  MI_REQUIRE((K & 15) == 0 && (N & 15) == 0, "dense_bwd_data_planes: N=%d and K=%d must be multiples of 16", N, K);
  MI_REQUIRE(planes_ok(dY, M, N) && planes_ok(W, K, N), "dense_bwd_data_planes: operand planes");
  MI_REQUIRE(!Xact || planes_ok(Xact, M, K), "dense_bwd_data_planes: activation planes");
  MI_REQUIRE(!mask_bits || mask_ld >= (K + 31) / 32, "dense_bwd_data_planes: mask_ld=%lld < ceil(K / 32)", (long long)mask_ld);
  MI_REQUIRE(dX || dXp, "dense_bwd_data_planes: no output");
  MI_REQUIRE(!dX || (lddx >= K && (lddx & 3) == 0 && mi::aligned16(dX)), "dense_bwd_data_planes: dX leading dimension / alignment");
  MI_REQUIRE(!dXp || planes_ok(dXp, M, K), "dense_bwd_data_planes: output planes");
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dense_bwd_data_planes: keep_prob=%f", keep_prob);
  PlArgs a{};
  a.A = static_cast<const char*>(W->data); a.bsa = W->blk_stride; a.a_exp = W->row_exp;
  a.B = static_cast<const char*>(dY->data); a.bsb = dY->blk_stride; a.b_exp = dY->row_exp;
  a.M = (int)M; a.N = K; a.K = N;
  a.C = dX; a.ldc = lddx;
  if (dXp) { a.Cp = static_cast<char*>(dXp->data); a.bsc = dXp->blk_stride; a.c_exp = dXp->row_exp; }
  a.keep_prob = keep_prob; a.keep_div = (Xact || mask_bits) ? keep_prob : 1.f; a.keep_rcp = 1.0f / a.keep_div;
  if (Xact) { a.mask = static_cast<const char*>(Xact->data); a.bsm = Xact->blk_stride; }
  a.mbits = const_cast<uint32_t*>(mask_bits); a.mbld = mask_ld;          // (read only in the DGRAD epilogue; preferred over Xact)
  a.amax_c = amax_out;
  return launch_pl<PL_DGRAD>(a, mi::as_stream(stream), "dense_bwd_data_planes");
}


size_t mi_hidden_logits_head_fused_workspace_bytes(int64_t M, int32_t N) {
  if (M <= 0 || N <= 0) return 256;
  return static_cast<size_t>(mi::ceil_div(M, 128)) * (N + 2) * sizeof(float) + 256;
}

int32_t mi_hidden_logits_head_fused(const mi_planes_t* X, const mi_planes_t* Wt, const float* bias, int64_t M, int32_t N, int32_t K,
                                    int32_t relu, float keep_prob, uint64_t seed, const float* w, const float* b, const float* lin,
                                    const float* lin_bias, const float* fm, const uint8_t* labels, float loss_scale, float* dnn,
                                    float* logits, float* loss_out, float* d_logit, float* d_logit_sum, float* dW, float* db,
                                    const mi_planes_t* dXp, float* amax_out, void* workspace, size_t workspace_bytes,
                                    mi_stream_t stream) {
  MI_REQUIRE(M > 0 && M <= INT32_MAX && N == 128 && K > 0 && (K & 15) == 0,
             "hidden_logits_head_fused: M=%lld N=%d (128) K=%d (a multiple of 16)", (long long)M, N, K);
  MI_REQUIRE(planes_ok(X, M, K) && planes_ok(Wt, N, K), "hidden_logits_head_fused: operand planes");
  MI_REQUIRE(w && labels && logits && d_logit && dW && workspace && mi::aligned16(w), "hidden_logits_head_fused: null buffer / alignment");
  MI_REQUIRE(planes_ok(dXp, M, N), "hidden_logits_head_fused: output planes");
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "hidden_logits_head_fused: keep_prob=%f", keep_prob);
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15u) == 0, "hidden_logits_head_fused: workspace alignment");
  if (workspace_bytes < mi_hidden_logits_head_fused_workspace_bytes(M, N)) {
    mi::set_error("hidden_logits_head_fused: workspace %zu < %zu", workspace_bytes, mi_hidden_logits_head_fused_workspace_bytes(M, N));
    return MI_ERR_WORKSPACE;
  }
  PlArgs a{};
  a.A = static_cast<const char*>(Wt->data); a.bsa = Wt->blk_stride; a.a_exp = Wt->row_exp;
  a.B = static_cast<const char*>(X->data); a.bsb = X->blk_stride; a.b_exp = X->row_exp;
  a.M = (int)M; a.N = N; a.K = K; a.tiles_n = 1;
  a.Cp = static_cast<char*>(dXp->data); a.bsc = dXp->blk_stride; a.c_exp = dXp->row_exp;
  a.bias = bias; a.relu = relu; a.keep_prob = keep_prob; a.keep_div = keep_prob; a.keep_rcp = 1.0f / keep_prob; a.seed = seed; a.st = mi::step_state();
  a.amax_c = amax_out;
  a.top_w = w; a.top_b = b; a.top_lin = lin; a.top_lin_bias = lin_bias; a.top_fm = fm; a.top_labels = labels; a.top_scale = loss_scale;
  a.top_dnn = dnn; a.top_logits = logits; a.top_dlogit = d_logit; a.top_part = static_cast<float*>(workspace);
  const int64_t blocks = mi::ceil_div(M, 128);
  hipStream_t st = mi::as_stream(stream);
  gemm_pl_k<1, 2, PL_TOP><<<dim3(static_cast<unsigned>(blocks)), dim3(PL_THREADS), 0, st>>>(a);
  MI_CHECK_LAUNCH("hidden_logits_head_fused");
  top_fold_k<<<dim3(static_cast<unsigned>(mi::ceil_div(N + 2, 16))), dim3(256), 0, st>>>(a.top_part, static_cast<int>(blocks), N, dW, db,
                                                                                         d_logit_sum, loss_out);
  MI_CHECK_LAUNCH("hidden_logits_head_fused(fold)");
  return MI_OK;
}

int32_t mi_dense_bwd_data_vec_planes(const float* dY, int64_t lddy, const float* W, const float* Xact, int64_t ldxa,
                                     float keep_prob, float* dX, int64_t lddx, const mi_planes_t* dXp, int64_t M, int32_t K,
                                     float* amax_out, const uint32_t* mask_bits, int64_t mask_ld, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K > 0 && (K & 15) == 0, "dense_bwd_data_vec_planes: M=%lld K=%d (K a multiple of 16)", (long long)M, K);
  MI_REQUIRE(!mask_bits || mask_ld >= (K + 31) / 32, "dense_bwd_data_vec_planes: mask_ld=%lld < ceil(K / 32)", (long long)mask_ld);
  if (M == 0) return MI_OK;
  MI_REQUIRE(dY && W && lddy >= 1 && mi::aligned16(W), "dense_bwd_data_vec_planes: dY / W");
  MI_REQUIRE(!Xact || (mi::aligned16(Xact) && ldxa >= K && (ldxa & 3) == 0), "dense_bwd_data_vec_planes: Xact leading dimension / alignment");
  MI_REQUIRE(!dX || (mi::aligned16(dX) && lddx >= K && (lddx & 3) == 0), "dense_bwd_data_vec_planes: dX leading dimension / alignment");
  MI_REQUIRE(planes_ok(dXp, M, K), "dense_bwd_data_vec_planes: output planes");
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dense_bwd_data_vec_planes: keep_prob=%f", keep_prob);
  int lpr = 4;                                   // lanes per row: one per float4, 4 .. 64
  while (lpr < 64 && lpr < (K >> 2)) lpr <<= 1;
  // rows per block: a 32 KB LDS image of their planes, at least one pass of the block's lanes, at most 256
  int rpb = 8192 / K;
  if (rpb < 256 / lpr) rpb = 256 / lpr;
  if (rpb > 256) rpb = 256;
  rpb -= rpb % (256 / lpr);
  const size_t lds = static_cast<size_t>(K >> 4) * rpb * PL_ROWB;
  MI_REQUIRE(lds <= 64 * 1024, "dense_bwd_data_vec_planes: K=%d too wide", K);
  const int64_t nb = mi::ceil_div(M, rpb);
  MI_REQUIRE(nb <= INT32_MAX, "dense_bwd_data_vec_planes: grid too large");
#define MI_VEC_DGRAD(L) vec_dgrad_planes_k<L><<<dim3((unsigned)nb), dim3(256), lds, mi::as_stream(stream)>>>( \
      dY, lddy, W, Xact, ldxa, (Xact || mask_bits) ? keep_prob : 1.f, (Xact || mask_bits) ? 1.0f / keep_prob : 1.f, M, K, rpb, dX, lddx, \
      static_cast<char*>(dXp->data), dXp->blk_stride, dXp->row_exp, amax_out, mask_bits, mask_ld)
  switch (lpr) {
    case 4: MI_VEC_DGRAD(4); break;
    case 8: MI_VEC_DGRAD(8); break;
    case 16: MI_VEC_DGRAD(16); break;
    case 32: MI_VEC_DGRAD(32); break;
    default: MI_VEC_DGRAD(64); break;
  }
#undef MI_VEC_DGRAD
  MI_CHECK_LAUNCH("dense_bwd_data_vec_planes");
  return MI_OK;
}

int32_t mi_selftest_div(float d, uint32_t first_bits, int64_t count, uint64_t* out, mi_stream_t stream) {
  MI_REQUIRE(count >= 0 && static_cast<uint64_t>(first_bits) + static_cast<uint64_t>(count) <= (1ull << 32), "selftest_div: range leaves 32 bits");
  MI_REQUIRE(d > 0.f, "selftest_div: d=%g", d);
  if (count == 0) return MI_OK;
  MI_REQUIRE(out, "selftest_div: null buffer");
  const int64_t blocks = mi::ceil_div(count, 256);
  MI_REQUIRE(blocks <= INT32_MAX, "selftest_div: grid too large");
  selftest_div_k<<<dim3((unsigned)blocks), dim3(256), 0, mi::as_stream(stream)>>>(d, 1.0f / d, first_bits, count,
                                                                                 reinterpret_cast<unsigned long long*>(out));
  MI_CHECK_LAUNCH("selftest_div");
  return MI_OK;
}

#ifdef MI_PL_STAMPS
int32_t mi_pl_stamps_read(void* dst, size_t nbytes) {
  return static_cast<int32_t>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pl_stamps), nbytes, 0, hipMemcpyDeviceToHost));
}
#endif

}  // extern "C"
