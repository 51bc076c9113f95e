// fp32 MFMA GEMMs of the [hidden_units] MLP with fused epilogues.
//
// Replaces tf.layers.dense / tf.layers.dropout (trainers/deep_fm.py:98-108) and their gradients.
// The 1e-5 logit bar forbids bf16 operands, so the matrix pipe runs v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulate: 64 FLOP/clk/SIMD, 157 TF/s chip peak).
//
// One kernel template, three operand layouts:
//   NN  Y  = X  * W      forward          A k-contiguous,  B n-contiguous
//   NT  dX = dY * W^T    data gradient    A k-contiguous,  B k-contiguous
//   TN  dW = X^T * dY    weight gradient  A m-contiguous,  B n-contiguous   (split-K over the batch)
// Tile 128x128x32, 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// VGPRs).  Operands are staged HBM -> registers -> LDS with the next tile's global loads in flight
// under the current tile's 64 MFMAs; two LDS buffers, one barrier per k-tile, two blocks per CU.
// k-contiguous operands sit in LDS as [mn][32+4] (ds_read_b128 conflict-free at the 36-float
// stride), mn-contiguous ones as [k][128] (ds_read_b32, one bank per lane).  A lane's four
// k-values per ds_read_b128 feed four consecutive MFMAs: lane half h owns k = 8s+4h+j.
// Workgroup ids are remapped so that the blocks sharing an A row-panel run on one XCD (shared L2).
#include "common.h"
#include "wgrad_pl.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KSTRIDE = BK + 4;           // k-contiguous LDS row stride (floats)
constexpr int TILE_FLOATS = BM * KSTRIDE;  // 4608 >= 32*128
constexpr int kThreads = 256;

enum { KC = 0, MC = 1 };                       // operand layouts
enum { EPI_BIAS_ACT = 0, EPI_MASK = 1, EPI_SLAB = 2 };

struct GemmArgs {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int M, N, K;
  int tiles_m, tiles_n, k_per_split;
  int vecA, vecB;
  int epi;
  const float* bias; int relu;
  float keep_prob; float keep_div; uint64_t seed;   // keep_div: what kept values are DIVIDED by (tf.nn.dropout: div(x, keep_prob) * mask)
  const float* mask_src; int64_t ldm;
  float* colsum_part;   // TN only: [splits][N] partial column sums of B (bias gradient), or NULL
  // Gathered A operand (layer 1 of the MLP): A is the embedding table and logical element
  // (example b, column c) is table[(g_off[c / g_E] + g_ids[b * g_F + c / g_E]) * g_E + c % g_E] —
  // the input_layer concat (deep_fm.py:54) read in place, never materialised.
  const int32_t* g_ids; const int64_t* g_off; int g_F, g_E;
  int64_t g_ts;         // floats between consecutive table rows (>= g_E: the rows may sit in [w | slot0 | slot1] records)
  // abs-max vectors (MI_AMAX_SLOTS floats each, value = largest entry): of the operands, for the
  // f16x2 split's scales; of the result, accumulated by the epilogue.  Any may be NULL.
  const float* amax_a; const float* amax_b; float* amax_c;
  const mi_step_state_t* st;   // device-resident step state of a captured step (seed += st->seed_term), or nullptr
};

// Dropout mask: mi_drop_* of common.h (element (row, col) is kept iff its 16 bits of the pair hash are below
// keep_prob * 2^16); the planes kernels (gemm_pl.hip) use the same functions, tests/util.py replays them on the host.

// Stage one operand tile HBM -> registers.  Branch-free and consumer-free on purpose: an
// out-of-range element reads the (always valid) first element of the buffer, and nothing touches
// the loaded registers until mask_tile() just before the LDS store, so the loads of a tile issue
// back to back and stay in flight under the MFMAs.  (With branches, or with the zero-select next
// to the load, hipcc parks an s_waitcnt vmcnt(0) behind every load and nothing overlaps.)
template <int L, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t ld, int MN, int mn0,
                                          int k0, int kend, float4 (&r)[4], int t) {
  if constexpr (L == KC) {
    const int kk = k0 + (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int mn = mn0 + (t >> 3) + 32 * p;
      const bool row_ok = mn < MN;
      const float* q = P + static_cast<int64_t>(row_ok ? mn : 0) * ld;
      if constexpr (VEC) {
        r[p] = *reinterpret_cast<const float4*>((row_ok && kk < kend) ? q + kk : P);
      } else {
        r[p].x = *((row_ok && kk < kend) ? q + kk : P);
        r[p].y = *((row_ok && kk + 1 < kend) ? q + kk + 1 : P);
        r[p].z = *((row_ok && kk + 2 < kend) ? q + kk + 2 : P);
        r[p].w = *((row_ok && kk + 3 < kend) ? q + kk + 3 : P);
      }
    }
  } else {
    const int mn = mn0 + (t & 31) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int kk = k0 + (t >> 5) + 8 * p;
      const bool k_ok = kk < kend;
      const float* q = P + static_cast<int64_t>(k_ok ? kk : 0) * ld;
      if constexpr (VEC) {
        r[p] = *reinterpret_cast<const float4*>((k_ok && mn < MN) ? q + mn : P);
      } else {
        r[p].x = *((k_ok && mn < MN) ? q + mn : P);
        r[p].y = *((k_ok && mn + 1 < MN) ? q + mn + 1 : P);
        r[p].z = *((k_ok && mn + 2 < MN) ? q + mn + 2 : P);
        r[p].w = *((k_ok && mn + 3 < MN) ? q + mn + 3 : P);
      }
    }
  }
}

// zero the elements load_tile fetched from the dummy address (same predicates, evaluated late)
template <int L, bool VEC>
__device__ __forceinline__ void mask_tile(int MN, int mn0, int k0, int kend, float4 (&r)[4], int t) {
  if constexpr (L == KC) {
    const int kk = k0 + (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool row_ok = mn0 + (t >> 3) + 32 * p < MN;
      if constexpr (VEC) {
        if (!(row_ok && kk < kend)) r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        r[p].x = (row_ok && kk < kend) ? r[p].x : 0.f;
        r[p].y = (row_ok && kk + 1 < kend) ? r[p].y : 0.f;
        r[p].z = (row_ok && kk + 2 < kend) ? r[p].z : 0.f;
        r[p].w = (row_ok && kk + 3 < kend) ? r[p].w : 0.f;
      }
    }
  } else {
    const int mn = mn0 + (t & 31) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool k_ok = k0 + (t >> 5) + 8 * p < kend;
      if constexpr (VEC) {
        if (!(k_ok && mn < MN)) r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        r[p].x = (k_ok && mn < MN) ? r[p].x : 0.f;
        r[p].y = (k_ok && mn + 1 < MN) ? r[p].y : 0.f;
        r[p].z = (k_ok && mn + 2 < MN) ? r[p].z : 0.f;
        r[p].w = (k_ok && mn + 3 < MN) ? r[p].w : 0.f;
      }
    }
  }
}

// ---- gathered A operand ------------------------------------------------------------------
// Row numbers of one A tile (4 per thread, one per staging slot p), fetched ONE TILE AHEAD of the
// row loads that use them so the id -> row dependent chain never sits in front of the MFMAs.
// Out-of-range slots get row 0 (masked to zero later by mask_tile).
template <int L>
__device__ __forceinline__ void gather_rows_for_tile(const GemmArgs& a, int MN, int mn0, int k0, int kend,
                                                     int (&row)[4], int t) {
  if constexpr (L == KC) {           // forward: M = examples, K = concat columns
    const int kk = k0 + (t & 7) * 4;
    const bool k_ok = kk < kend;
    const int f = k_ok ? kk / a.g_E : 0;
    const int64_t off = a.g_off[f];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = mn0 + (t >> 3) + 32 * p;
      const bool ok = k_ok && m < MN;
      row[p] = static_cast<int>(off + a.g_ids[ok ? static_cast<int64_t>(m) * a.g_F + f : 0]);
    }
  } else {                           // weight gradient: M = concat columns, K = examples
    const int mn = mn0 + (t & 31) * 4;
    const bool m_ok = mn < MN;
    const int f = m_ok ? mn / a.g_E : 0;
    const int64_t off = a.g_off[f];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int b = k0 + (t >> 5) + 8 * p;
      const bool ok = m_ok && b < kend;
      row[p] = static_cast<int>(off + a.g_ids[ok ? static_cast<int64_t>(b) * a.g_F + f : 0]);
    }
  }
}

template <int L>
__device__ __forceinline__ void load_tile_gathered(const GemmArgs& a, int mn0, int k0, const int (&row)[4],
                                                   float4 (&r)[4], int t) {
  const int c = (L == KC) ? k0 + (t & 7) * 4 : mn0 + (t & 31) * 4;   // concat column of this thread's float4
  const int e = c % a.g_E;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    r[p] = *reinterpret_cast<const float4*>(a.A + static_cast<int64_t>(row[p]) * a.g_ts + e);
}

template <int L>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4 (&r)[4], int t) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if constexpr (L == KC)
      *reinterpret_cast<float4*>(S + ((t >> 3) + 32 * p) * KSTRIDE + (t & 7) * 4) = r[p];
    else
      *reinterpret_cast<float4*>(S + ((t >> 5) + 8 * p) * BM + (t & 31) * 4) = r[p];
  }
}

// fragment of one 32-row MFMA tile for k-group s: four k values (k = 8s + 4h + j, j = 0..3)
template <int L>
__device__ __forceinline__ void read_frag(const float* __restrict__ S, int mn, int s, int h,
                                          float (&f)[4]) {
  if constexpr (L == KC) {
    const float4 v = *reinterpret_cast<const float4*>(S + mn * KSTRIDE + 8 * s + 4 * h);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = S[(8 * s + 4 * h + j) * BM + mn];
  }
}

// activation of tf.layers.dense (deep_fm.py:22,100: params["activation"], default tf.nn.relu): 0 none, 1 relu,
// 2 sigmoid, 3 tanh; and its derivative expressed through the activation's OUTPUT y (what the forward stored)
__device__ __forceinline__ float act_apply(int kind, float v) {
  switch (kind) {
    case 1: return fmaxf(v, 0.f);
    case 2: return 1.f / (1.f + __expf(-v));
    case 3: return tanhf(v);
    default: return v;
  }
}
__device__ __forceinline__ float act_deriv_from_output(int kind, float y) {
  switch (kind) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y * (1.f - y);
    case 3: return 1.f - y * y;
    default: return 1.f;
  }
}

// ---- epilogue shared by the fp32-MFMA and the bf16x3-split kernels --------------------------
// C/D register map of every 32x32 MFMA (dtype independent): col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  i = lane&31, h = lane>>5.
// sa * sb undoes the operand scales of the f16x2 split (powers of two; 1 elsewhere).  When a.amax_c
// is set the largest |value stored| goes into one of its MI_AMAX_SLOTS entries (atomic max on the
// bit pattern; spread over slots so that 8k waves do not queue on one address).
__device__ __forceinline__ void store_tile_c(const GemmArgs& a, const f32x16 (&acc)[2][2], int m0, int n0, int wm,
                                             int wn, int i, int h, int split, float sa, float sb) {
  float* Cb = a.C;
  float mx = 0.f;
  if (a.epi == EPI_SLAB) Cb += static_cast<int64_t>(split) * a.M * a.ldc;
  const uint32_t thresh = mi_drop_thresh16(a.keep_prob);
  const uint64_t seed = a.seed + (a.st ? a.st->seed_term : 0ull);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + i;
    if (col >= a.N) continue;
    const float bv = (a.epi == EPI_BIAS_ACT && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= a.M) continue;
        float v = acc[mi][ni][r] * sa * sb;
        if (a.epi == EPI_BIAS_ACT) {
          v += bv;
          v = act_apply(a.relu, v);
          if (a.keep_prob < 1.f) v = mi_drop_keep_at(seed, row, col, thresh) ? v / a.keep_div : 0.f;
        } else if (a.epi == EPI_MASK) {
          if (a.mask_src) {
            const float x = a.mask_src[static_cast<int64_t>(row) * a.ldm + col];
            if (a.relu == 1) {
              v = (x > 0.f) ? v / a.keep_div : 0.f;          // active and kept
            } else {
              // stored output x = act(pre) / keep or 0 (dropped; with dropout an output that is exactly 0 counts as dropped)
              const bool dropped = a.keep_div < 1.f && x == 0.f;
              v = dropped ? 0.f : (v / a.keep_div) * act_deriv_from_output(a.relu, x * a.keep_div);
            }
          }
        }
        Cb[static_cast<int64_t>(row) * a.ldc + col] = v;
        mx = fmaxf(mx, fabsf(v));
      }
    }
  }
  if (a.amax_c) mi_amax_publish(a.amax_c, mx);
}

template <int LA, int LB, bool VA, bool VB, bool COLSUM, bool GATHER>
#ifndef GEMM_LB_WAVES
#define GEMM_LB_WAVES 2
#endif
__global__ __launch_bounds__(kThreads, GEMM_LB_WAVES) void gemm_f32_k(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][TILE_FLOATS];

  // XCD-aware bijective remap: the 8 XCDs receive blocks round-robin; give each XCD a contiguous
  // run of logical tiles so that the tiles_n column blocks of one A row-panel share an L2.
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < rr) ? xcd * (q + 1) + idx : rr * (q + 1) + (xcd - rr) * q + idx;

  const int tn = lid % a.tiles_n;
  const int tm = (lid / a.tiles_n) % a.tiles_m;
  const int split = lid / (a.tiles_n * a.tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * a.k_per_split;
  const int kend = min(a.K, kbeg + a.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const int t = threadIdx.x;
  const int w = t >> 6, lane = t & 63;
  const int wm = w >> 1, wn = w & 1;
  const int i = lane & 31, h = lane >> 5;
  const int arow = wm * 64 + i, brow = wn * 64 + i;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  float4 ra[4], rb[4];
  // COLSUM (weight gradient only): the bias gradient rides along — the tm == 0 blocks add up the
  // rows of the B (= dY) tiles they stage anyway; thread t owns columns 4*(t&31).. and k rows
  // (t>>5)+8p.
  const bool do_colsum = COLSUM && a.colsum_part != nullptr && tm == 0;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  int grow[4] = {0, 0, 0, 0};        // GATHER: rows of the NEXT A tile to load
  if (nk > 0) {
    if constexpr (GATHER) {
      gather_rows_for_tile<LA>(a, a.M, m0, kbeg, kend, grow, t);
      load_tile_gathered<LA>(a, m0, kbeg, grow, ra, t);
      if (nk > 1) gather_rows_for_tile<LA>(a, a.M, m0, kbeg + BK, kend, grow, t);
    } else {
      load_tile<LA, VA>(a.A, a.lda, a.M, m0, kbeg, kend, ra, t);
    }
    load_tile<LB, VB>(a.B, a.ldb, a.N, n0, kbeg, kend, rb, t);
    mask_tile<LA, VA>(a.M, m0, kbeg, kend, ra, t);
    mask_tile<LB, VB>(a.N, n0, kbeg, kend, rb, t);
    store_tile<LA>(smem[0][0], ra, t);
    store_tile<LB>(smem[0][1], rb, t);
    if (COLSUM && do_colsum) {
#pragma unroll
      for (int p = 0; p < 4; ++p) { cs.x += rb[p].x; cs.y += rb[p].y; cs.z += rb[p].z; cs.w += rb[p].w; }
    }
  }
  __syncthreads();

  // Per k-tile: the next tile's global loads are issued first and land in registers under the
  // MFMAs; the staged registers go to the other LDS buffer (idle since the previous barrier)
  // at GEMM_STORE_POS; one barrier per tile.
#ifndef GEMM_STORE_POS
#define GEMM_STORE_POS 2   /* after k-group n of 4 (4 = end of tile); A/B in tools/gemm_bench.py */
#endif
#ifndef GEMM_FRAG_DB
#define GEMM_FRAG_DB 1
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      if constexpr (GATHER) {
        load_tile_gathered<LA>(a, m0, kbeg + (kt + 1) * BK, grow, ra, t);
        if (kt + 2 < nk) gather_rows_for_tile<LA>(a, a.M, m0, kbeg + (kt + 2) * BK, kend, grow, t);
      } else {
        load_tile<LA, VA>(a.A, a.lda, a.M, m0, kbeg + (kt + 1) * BK, kend, ra, t);
      }
      load_tile<LB, VB>(a.B, a.ldb, a.N, n0, kbeg + (kt + 1) * BK, kend, rb, t);
    }
    const float* As = smem[cur][0];
    const float* Bs = smem[cur][1];
    auto stage = [&]() {
      if (more) {
        mask_tile<LA, VA>(a.M, m0, kbeg + (kt + 1) * BK, kend, ra, t);
        mask_tile<LB, VB>(a.N, n0, kbeg + (kt + 1) * BK, kend, rb, t);
        store_tile<LA>(smem[cur ^ 1][0], ra, t);
        store_tile<LB>(smem[cur ^ 1][1], rb, t);
        if (COLSUM && do_colsum) {
#pragma unroll
          for (int p = 0; p < 4; ++p) { cs.x += rb[p].x; cs.y += rb[p].y; cs.z += rb[p].z; cs.w += rb[p].w; }
        }
      }
    };
#if GEMM_FRAG_DB
    float fa[2][2][4], fb[2][2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) read_frag<LA>(As, arow + mi * 32, 0, h, fa[0][mi]);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) read_frag<LB>(Bs, brow + ni * 32, 0, h, fb[0][ni]);
#endif
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
#if GEMM_FRAG_DB
      const int c = s & 1;
      if (s + 1 < BK / 8) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) read_frag<LA>(As, arow + mi * 32, s + 1, h, fa[c ^ 1][mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) read_frag<LB>(Bs, brow + ni * 32, s + 1, h, fb[c ^ 1][ni]);
      }
#define FA(mi, j) fa[c][mi][j]
#define FB(ni, j) fb[c][ni][j]
#else
      float fa[2][4], fb[2][4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) read_frag<LA>(As, arow + mi * 32, s, h, fa[mi]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) read_frag<LB>(Bs, brow + ni * 32, s, h, fb[ni]);
#define FA(mi, j) fa[mi][j]
#define FB(ni, j) fb[ni][j]
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA(mi, j), FB(ni, j), acc[mi][ni], 0, 0, 0);
#undef FA
#undef FB
      if (s + 1 == GEMM_STORE_POS) stage();
    }
    __syncthreads();
  }

  if (COLSUM && do_colsum) {   // fold the 8 k-row groups in a fixed order, one partial row per split
    float* red = &smem[0][0][0];
    *reinterpret_cast<float4*>(red + (t >> 5) * BN + (t & 31) * 4) = cs;
    __syncthreads();
    if (t < BN) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += red[g * BN + t];
      if (n0 + t < a.N) a.colsum_part[static_cast<int64_t>(split) * a.N + n0 + t] = v;
    }
  }

  store_tile_c(a, acc, m0, n0, wm, wn, i, h, split, 1.f, 1.f);
}

#include "gemm_split.inc"
#include "gemm_wgrad_pl.inc"

// which matrix-pipe path the vectorisable GEMMs take: 1 = 16-bit operand split (default; f16x2 when
// the call carries the operands' abs-max, bf16x3 otherwise), 0 = fp32-input MFMA
int g_gemm_mode = 1;

// abs-max of a buffer into an abs-max vector (grid-stride, float4 body; one atomic per wave)
__global__ __launch_bounds__(kThreads) void absmax_k(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  float mx = 0.f;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 ? n >> 2 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < n4; j += stride) {
    const float4 v = x4[j];
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t j = 4 * n4 + static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < n; j += stride)
    mx = fmaxf(mx, fabsf(x[j]));
  mi_amax_publish(out, mx);
}

// out[i] = sum_s slab[s][i] in a fixed order (bitwise reproducible split-K): the 4 waves of a block take every 4th slab
// each (wave g: s = g, g + 4, ...), then fold through LDS as (w0 + w1) + (w2 + w3).  A lane owns VEC consecutive outputs
// (a block 64 VEC of them): VEC = 4 — 16-byte loads, eight in flight per lane — where that still leaves >= 512 blocks
// (the layer-1 fold: 65 MB of slabs), VEC = 1 for the small matrices, whose parallelism is the slabs.
// A second vector (the bias gradient's partials) rides in the same launch: blocks past the first vector's reduce it (VEC = 1).
template <int VEC>
__device__ __forceinline__ void slab_reduce_body(const float* __restrict__ slab, int nsplit, int64_t n, float* __restrict__ out,
                                                 int64_t blk, float (*red)[64 * 4]) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (blk * 64 + c) * VEC;
  float acc[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
  if (i < n) {
    const float* p = slab + i;
    int s = g;
    if constexpr (VEC == 4) {
      for (; s + 28 < nsplit; s += 32) {               // eight slabs of this wave at once
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(p + static_cast<int64_t>(s + 4 * q) * n);
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc[0] += v[q].x; acc[1] += v[q].y; acc[2] += v[q].z; acc[3] += v[q].w; }
      }
      for (; s < nsplit; s += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + static_cast<int64_t>(s) * n);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      }
    } else {
#pragma unroll 8
      for (; s < nsplit; s += 4) acc[0] += p[static_cast<int64_t>(s) * n];
    }
  }
#pragma unroll
  for (int q = 0; q < VEC; ++q) red[g][c * VEC + q] = acc[q];
  __syncthreads();
  if (g == 0 && i < n) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) out[i + q] = (red[0][c * VEC + q] + red[1][c * VEC + q]) + (red[2][c * VEC + q] + red[3][c * VEC + q]);
  }
}
template <int VEC>
__global__ __launch_bounds__(kThreads) void slab_reduce_k(const float* __restrict__ slab, int nsplit,
                                                          int64_t n, float* __restrict__ out,
                                                          const float* __restrict__ slab2 = nullptr, int64_t n2 = 0,
                                                          float* __restrict__ out2 = nullptr) {
  __shared__ float red[4][64 * 4];
  const int64_t nb1 = (n + 64 * VEC - 1) / (64 * VEC);
  const int64_t blk = blockIdx.x;
  if (blk >= nb1) slab_reduce_body<1>(slab2, nsplit, n2, out2, blk - nb1, red);        // (block-uniform)
  else slab_reduce_body<VEC>(slab, nsplit, n, out, blk, red);
}
// the folds of several layers' slabs in ONE launch (mi_dense_bwd_weight_planes_batch): blocks [block0, ...) are job j's
struct FoldJob { const float* slab; int nsplit; int64_t n; float* out; const float* slab2; int64_t n2; float* out2; int vec; int block0; };
struct FoldJobs { FoldJob j[MI_MAX_WEIGHT_JOBS]; int n; };
inline bool fold_vec_ok(const float* slab, int64_t n, const float* out) {
  return (n & 3) == 0 && n >= 4 * 64 * 512 && (reinterpret_cast<uintptr_t>(slab) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
}
inline int64_t fold_blocks(const FoldJob& f) { return mi::ceil_div(f.n, f.vec ? 256 : 64) + (f.n2 > 0 ? mi::ceil_div(f.n2, 64) : 0); }
__global__ __launch_bounds__(kThreads) void slab_reduce_multi_k(const FoldJobs js) {
  __shared__ float red[4][64 * 4];
  int ji = 0;
#pragma unroll
  for (int q = 1; q < MI_MAX_WEIGHT_JOBS; ++q)
    if (q < js.n && static_cast<int>(blockIdx.x) >= js.j[q].block0) ji = q;
  const FoldJob& f = js.j[ji];
  const int64_t blk = static_cast<int>(blockIdx.x) - f.block0;
  const int64_t nb1 = (f.n + (f.vec ? 256 : 64) - 1) / (f.vec ? 256 : 64);
  if (blk >= nb1) slab_reduce_body<1>(f.slab2, f.nsplit, f.n2, f.out2, blk - nb1, red);        // (block-uniform branches)
  else if (f.vec) slab_reduce_body<4>(f.slab, f.nsplit, f.n, f.out, blk, red);
  else slab_reduce_body<1>(f.slab, f.nsplit, f.n, f.out, blk, red);
}

inline void slab_reduce(const float* slab, int nsplit, int64_t n, float* out, const float* slab2, int64_t n2, float* out2, hipStream_t st) {
  const bool vec = fold_vec_ok(slab, n, out);
  const int64_t nb2 = n2 > 0 ? mi::ceil_div(n2, 64) : 0;
  if (vec) slab_reduce_k<4><<<dim3((unsigned)(mi::ceil_div(n, 256) + nb2)), dim3(kThreads), 0, st>>>(slab, nsplit, n, out, slab2, n2, out2);
  else slab_reduce_k<1><<<dim3((unsigned)(mi::ceil_div(n, 64) + nb2)), dim3(kThreads), 0, st>>>(slab, nsplit, n, out, slab2, n2, out2);
}

// column sums, stage 1: block = 64 columns x a slab of rows; thread (c, g) strides rows by 4.
constexpr int kColsumRows = 512;
__global__ __launch_bounds__(kThreads) void colsum_part_k(const float* __restrict__ X, int64_t ldx,
                                                          int64_t M, int N, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * kColsumRows;
  const int64_t r1 = min(M, r0 + kColsumRows);
  float acc = 0.f;
  if (col < N)
    for (int64_t r = r0 + g; r < r1; r += 4) acc += X[r * ldx + col];
  red[g][c] = acc;
  __syncthreads();
  if (g == 0 && col < N)
    part[static_cast<int64_t>(blockIdx.y) * N + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ __launch_bounds__(kThreads) void colsum_final_k(const float* __restrict__ part, int nparts,
                                                           int N, float* __restrict__ out) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= N) return;
  float acc = 0.f;
  for (int p = 0; p < nparts; ++p) acc += part[static_cast<int64_t>(p) * N + c];
  out[c] = acc;
}

// ---- the N = 1 logits layer (deep_fm.py:108): matrix-vector forms -----------------------------
// A 128x128 MFMA tile for one output column wastes 127/128 of the matrix pipe; these are plain
// HBM-bound kernels instead (read X once / write dX once).  16 lanes per example.
constexpr int kGvLanes = 16;

__global__ __launch_bounds__(kThreads) void gemv_fwd_k(const float* __restrict__ X, int64_t ldx,
                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                       float* __restrict__ Y, int64_t ldy, int64_t M, int K, int relu,
                                                       float keep_prob, float keep_div, uint64_t seed,
                                                       float* __restrict__ amax_out, const mi_step_state_t* __restrict__ st) {
  if (st) seed += st->seed_term;
  const int l = threadIdx.x & (kGvLanes - 1);
  const int64_t groups = static_cast<int64_t>(gridDim.x) * (kThreads / kGvLanes);
  const uint32_t thresh = mi_drop_thresh16(keep_prob);
  const float b0 = bias ? bias[0] : 0.f;
  float mx = 0.f;
  for (int64_t m0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / kGvLanes; m0 < M; m0 += 2 * groups) {
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {                  // two rows in flight per lane group
      const int64_t m = min(m0 + j * groups, M - 1);
      const float* x = X + m * ldx;
      for (int k = 4 * l; k < K; k += 4 * kGvLanes) {
        const float4 a = *reinterpret_cast<const float4*>(x + k);
        const float4 w = *reinterpret_cast<const float4*>(W + k);
        acc[j] += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int o = kGvLanes / 2; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o, kGvLanes);
      const int64_t m = m0 + j * groups;
      if (m < M && l == 0) {
        float v = acc[j] + b0;
        v = act_apply(relu, v);
        if (keep_prob < 1.f) v = mi_drop_keep_at(seed, static_cast<uint32_t>(m), 0u, thresh) ? v / keep_div : 0.f;
        Y[m * ldy] = v;
        mx = fmaxf(mx, fabsf(v));
      }
    }
  }
  const float v = mx;
  if (amax_out) mi_amax_publish(amax_out, fabsf(v));
}

// dX[m,k] = dY[m] * W[k], masked by the stored activation (see mi_dense_bwd_data)
__global__ __launch_bounds__(kThreads) void gemv_dgrad_k(const float* __restrict__ dY, int64_t lddy,
                                                         const float* __restrict__ W, const float* __restrict__ Xact,
                                                         int64_t ldxa, float* __restrict__ dX, int64_t lddx, int64_t M,
                                                         int K, float keep_div, int act, float* __restrict__ amax_out) {
  const int kq = K >> 2;
  const int64_t total = M * kq, stride = static_cast<int64_t>(gridDim.x) * kThreads;
  float mx = 0.f;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; base < total; base += 4 * stride) {
    float4 xv[4], wv[4];
    float gv[4];
    int64_t mm[4];
    int kk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                  // loads of four elements first, then the arithmetic
      const int64_t idx = min(base + j * stride, total - 1);
      mm[j] = idx / kq;
      kk[j] = static_cast<int>(idx - mm[j] * kq) * 4;
      gv[j] = dY[mm[j] * lddy];
      wv[j] = *reinterpret_cast<const float4*>(W + kk[j]);
      if (Xact) xv[j] = *reinterpret_cast<const float4*>(Xact + mm[j] * ldxa + kk[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (base + j * stride >= total) break;
      float4 v = make_float4(gv[j] * wv[j].x, gv[j] * wv[j].y, gv[j] * wv[j].z, gv[j] * wv[j].w);
      if (Xact) {
        if (act == 1) {
          v.x = xv[j].x > 0.f ? v.x / keep_div : 0.f; v.y = xv[j].y > 0.f ? v.y / keep_div : 0.f;
          v.z = xv[j].z > 0.f ? v.z / keep_div : 0.f; v.w = xv[j].w > 0.f ? v.w / keep_div : 0.f;
        } else {
          const bool dr = keep_div < 1.f;
          v.x = (dr && xv[j].x == 0.f) ? 0.f : (v.x / keep_div) * act_deriv_from_output(act, xv[j].x * keep_div);
          v.y = (dr && xv[j].y == 0.f) ? 0.f : (v.y / keep_div) * act_deriv_from_output(act, xv[j].y * keep_div);
          v.z = (dr && xv[j].z == 0.f) ? 0.f : (v.z / keep_div) * act_deriv_from_output(act, xv[j].z * keep_div);
          v.w = (dr && xv[j].w == 0.f) ? 0.f : (v.w / keep_div) * act_deriv_from_output(act, xv[j].w * keep_div);
        }
      }
      *reinterpret_cast<float4*>(dX + mm[j] * lddx + kk[j]) = v;
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
  if (amax_out) mi_amax_publish(amax_out, mx);
}

// slab s of rows: part[s][k] = sum_m X[m,k] dY[m], bpart[s] = sum_m dY[m]; rows in ascending order per
// thread column, the 64 row-lanes of a column folded in a fixed tree: reproducible.  slab_reduce_k adds
// the slabs.  Block = 4 column-float4 x 64 row lanes.
__global__ __launch_bounds__(kThreads) void gemv_wgrad_k(const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ dY, int64_t lddy, int64_t M, int K,
                                                         int64_t rows_per_slab, float* __restrict__ part,
                                                         float* __restrict__ bpart) {
  __shared__ float4 red[kThreads];
  __shared__ float redb[64];
  const int c = threadIdx.x & 3, rl = threadIdx.x >> 2;            // column group inside the block, row lane
  const int k = (blockIdx.x * 4 + c) * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float accb = 0.f;
  for (int64_t m = r0 + rl; m < r1; m += 64) {
    const float g = dY[m * lddy];
    if (k < K) {
      const float4 x = *reinterpret_cast<const float4*>(X + m * ldx + k);
      acc.x += x.x * g; acc.y += x.y * g; acc.z += x.z * g; acc.w += x.w * g;
    }
    if (c == 0) accb += g;
  }
  red[threadIdx.x] = acc;
  if (c == 0) redb[rl] = accb;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (rl < o) {
      const float4 b = red[threadIdx.x + 4 * o];
      float4 a = red[threadIdx.x];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      red[threadIdx.x] = a;
      if (c == 0) redb[rl] += redb[rl + o];
    }
    __syncthreads();
  }
  if (rl == 0 && k < K) *reinterpret_cast<float4*>(part + static_cast<int64_t>(blockIdx.y) * K + k) = red[c];
  if (threadIdx.x == 0 && blockIdx.x == 0 && bpart) bpart[blockIdx.y] = redb[0];
}

bool gemv_ok(const float* X, int64_t ldx, const float* W, int K) {
  return mi::aligned16(X) && mi::aligned16(W) && (ldx & 3) == 0 && (K & 3) == 0;
}

bool vec_ok(const float* p, int64_t ld, int contiguous_extent) {
  return mi::aligned16(p) && (ld & 3) == 0 && (contiguous_extent & 3) == 0;
}

template <int LA, int LB, bool COLSUM = false, bool GATHER = false>
int32_t launch(GemmArgs& a, int splits, hipStream_t st, const char* what) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const int64_t nblocks = static_cast<int64_t>(a.tiles_m) * a.tiles_n * splits;
  if (nblocks <= 0 || nblocks > INT32_MAX) {
    mi::set_error("%s: bad grid (%lld blocks)", what, (long long)nblocks);
    return MI_ERR_INVALID;
  }
  const dim3 g((unsigned)nblocks), b(kThreads);
  if (g_gemm_mode == 1 && a.vecA && a.vecB) {
    // The f16x2 split scales an operand by ONE power of two (its abs-max): rows far below the matrix abs-max
    // lose their low bits.  That is harmless only where the reduction runs over those rows — the weight
    // gradient (both operands mn-contiguous: k = examples) — so only it takes the matrix-wide scales.  The
    // forward pass and the data gradient take per-row exponents (mi_dense_fwd_planes / mi_dense_bwd_data_planes)
    // or, through these any-shape entries, the scale-free bf16x3 split.
    if (a.amax_a && a.amax_b && LA == MC && LB == MC) {
      // whole tiles, lane offsets that fit 32 bits, and (forward gather) k-tiles inside one field
      const bool whole = a.M % BM == 0 && a.N % BN == 0 && a.K % BK == 0 && a.k_per_split % BK == 0 &&
                         a.lda < (1 << 22) && a.ldb < (1 << 22) &&
                         (!GATHER || (a.g_F < (1 << 20) && (LA != KC || a.g_E % BK == 0)));
      if (whole) gemm_split_k<FMT_F16X2, LA, LB, COLSUM, GATHER, false><<<g, b, 0, st>>>(a);
      else gemm_split_k<FMT_F16X2, LA, LB, COLSUM, GATHER, true><<<g, b, 0, st>>>(a);
    } else {
      gemm_split_k<FMT_BF16X3, LA, LB, COLSUM, GATHER, true><<<g, b, 0, st>>>(a);
    }
    MI_CHECK_LAUNCH(what);
    return MI_OK;
  }
  if constexpr (GATHER) {       // the gathered operand is always float4-addressable (E % 4 == 0)
    if (a.vecB) gemm_f32_k<LA, LB, true, true, COLSUM, true><<<g, b, 0, st>>>(a);
    else gemm_f32_k<LA, LB, true, false, COLSUM, true><<<g, b, 0, st>>>(a);
  } else {
    if (a.vecA && a.vecB) gemm_f32_k<LA, LB, true, true, COLSUM, false><<<g, b, 0, st>>>(a);
    else if (a.vecA) gemm_f32_k<LA, LB, true, false, COLSUM, false><<<g, b, 0, st>>>(a);
    else if (a.vecB) gemm_f32_k<LA, LB, false, true, COLSUM, false><<<g, b, 0, st>>>(a);
    else gemm_f32_k<LA, LB, false, false, COLSUM, false><<<g, b, 0, st>>>(a);
  }
  MI_CHECK_LAUNCH(what);
  return MI_OK;
}

// split-K policy for the weight gradient: about two full rounds of resident workgroups (2 per CU),
// k slices a multiple of BK
int wgrad_splits(int64_t M, int N, int K) {
  const int64_t tiles = mi::ceil_div(K, BM) * mi::ceil_div(N, BN);
  int64_t s = 1024 / tiles;
  const int64_t max_s = mi::ceil_div(M, 4 * BK);  // at least 4 k-tiles per slice
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return static_cast<int>(s);
}

int64_t wgrad_k_per_split(int64_t M, int splits) {
  return mi::ceil_div(mi::ceil_div(M, splits), BK) * BK;
}

void set_amax(GemmArgs& a, const mi_gemm_amax_t* amax) {
  if (!amax) return;
  a.amax_a = amax->a; a.amax_b = amax->b; a.amax_c = amax->out;
}

}  // namespace

extern "C" {

int32_t mi_absmax(const float* x, int64_t n, float* amax_out, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && amax_out && (n == 0 || x), "absmax: n=%lld", (long long)n);
  if (n == 0) return MI_OK;
  const int64_t blocks = std::min<int64_t>(mi::ceil_div(n, 16 * kThreads), 1024);
  absmax_k<<<dim3((unsigned)blocks), dim3(kThreads), 0, mi::as_stream(stream)>>>(x, n, amax_out);
  MI_CHECK_LAUNCH("absmax");
  return MI_OK;
}

int32_t mi_set_gemm_mode(int32_t mode) {
  MI_REQUIRE(mode == 0 || mode == 1, "set_gemm_mode: %d (0 = fp32-input MFMA, 1 = 16-bit operand split)", mode);
  g_gemm_mode = mode;
  return MI_OK;
}

int32_t mi_get_gemm_mode(void) { return g_gemm_mode; }

int32_t mi_dense_fwd(const float* X, int64_t ldx, const float* W, const float* bias, float* Y,
                     int64_t ldy, int64_t M, int32_t N, int32_t K, int32_t relu, float keep_prob,
                     uint64_t seed, const mi_gemm_amax_t* amax, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && M <= INT32_MAX && N > 0 && K > 0, "dense_fwd: M=%lld N=%d K=%d", (long long)M, N, K);
  if (M == 0) return MI_OK;
  MI_REQUIRE(X && W && Y, "dense_fwd: null buffer");
  MI_REQUIRE(ldx >= K && ldy >= N, "dense_fwd: ldx=%lld ldy=%lld", (long long)ldx, (long long)ldy);
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dense_fwd: keep_prob=%f", keep_prob);
  MI_REQUIRE(relu >= 0 && relu <= 3, "dense_fwd: activation=%d (0 none, 1 relu, 2 sigmoid, 3 tanh)", relu);
  if (N == 1 && gemv_ok(X, ldx, W, K)) {
    const int64_t blocks = std::min<int64_t>(mi::ceil_div(M * kGvLanes, 2 * kThreads), 2048);
    gemv_fwd_k<<<dim3((unsigned)blocks), dim3(kThreads), 0, mi::as_stream(stream)>>>(
        X, ldx, W, bias, Y, ldy, M, K, relu, keep_prob, keep_prob, seed, amax ? amax->out : nullptr, mi::step_state());
    MI_CHECK_LAUNCH("dense_fwd(N = 1)");
    return MI_OK;
  }
  GemmArgs a{};
  a.A = X; a.lda = ldx; a.B = W; a.ldb = N; a.C = Y; a.ldc = ldy;
  a.M = (int)M; a.N = N; a.K = K; a.k_per_split = ((K + BK - 1) / BK) * BK;
  a.vecA = vec_ok(X, ldx, K); a.vecB = vec_ok(W, N, N);
  a.epi = EPI_BIAS_ACT; a.bias = bias; a.relu = relu;
  a.keep_prob = keep_prob; a.keep_div = keep_prob; a.seed = seed; a.st = mi::step_state();
  set_amax(a, amax);
  return launch<KC, MC>(a, 1, mi::as_stream(stream), "dense_fwd");
}

int32_t mi_dense_bwd_data(const float* dY, int64_t lddy, const float* W, const float* Xact,
                          int64_t ldxa, float* dX, int64_t lddx, int64_t M, int32_t N, int32_t K,
                          float keep_prob, int32_t activation, const mi_gemm_amax_t* amax, mi_stream_t stream) {
  MI_REQUIRE(activation >= 0 && activation <= 3, "dense_bwd_data: activation=%d (0 none, 1 relu, 2 sigmoid, 3 tanh)", activation);
  MI_REQUIRE(M >= 0 && M <= INT32_MAX && N > 0 && K > 0, "dense_bwd_data: M=%lld N=%d K=%d", (long long)M, N, K);
  if (M == 0) return MI_OK;
  MI_REQUIRE(dY && W && dX, "dense_bwd_data: null buffer");
  MI_REQUIRE(lddy >= N && lddx >= K && (!Xact || ldxa >= K), "dense_bwd_data: leading dimensions");
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dense_bwd_data: keep_prob=%f", keep_prob);
  if (N == 1 && mi::aligned16(W) && mi::aligned16(dX) && (lddx & 3) == 0 && (K & 3) == 0 &&
      (!Xact || (mi::aligned16(Xact) && (ldxa & 3) == 0))) {
    const int64_t blocks = std::min<int64_t>(mi::ceil_div(M * (K >> 2), 4 * kThreads), 4096);
    gemv_dgrad_k<<<dim3((unsigned)blocks), dim3(kThreads), 0, mi::as_stream(stream)>>>(
        dY, lddy, W, Xact, ldxa, dX, lddx, M, K, Xact ? keep_prob : 1.f, activation, amax ? amax->out : nullptr);
    MI_CHECK_LAUNCH("dense_bwd_data(N = 1)");
    return MI_OK;
  }
  GemmArgs a{};                       // dX[M,K] = dY[M,N] * W[K,N]^T : gemm M x K x (reduce N)
  a.A = dY; a.lda = lddy; a.B = W; a.ldb = N; a.C = dX; a.ldc = lddx;
  a.M = (int)M; a.N = K; a.K = N; a.k_per_split = ((N + BK - 1) / BK) * BK;
  a.vecA = vec_ok(dY, lddy, N); a.vecB = vec_ok(W, N, N);
  a.epi = EPI_MASK; a.mask_src = Xact; a.ldm = ldxa; a.relu = activation;
  a.keep_prob = keep_prob; a.keep_div = Xact ? keep_prob : 1.f;
  set_amax(a, amax);
  return launch<KC, KC>(a, 1, mi::as_stream(stream), "dense_bwd_data");
}

static int32_t check_gather(const char* who, const float* table, const int64_t* field_off, const int32_t* ids,
                            int32_t F, int32_t E) {
  MI_REQUIRE(table && field_off && ids && F > 0, "%s: null gather operand", who);
  MI_REQUIRE(E >= 4 && (E & 3) == 0 && mi::aligned16(table), "%s: embedding size %d must be a multiple of 4", who, E);
  return MI_OK;
}

int32_t mi_dense_fwd_gathered(const float* table, const int64_t* field_off, const int32_t* ids, int32_t F,
                              int32_t E, const float* W, const float* bias, float* Y, int64_t ldy, int64_t M,
                              int32_t N, int32_t relu, float keep_prob, uint64_t seed, const mi_gemm_amax_t* amax,
                              int64_t table_stride, mi_stream_t stream) {
  if (int32_t rc = check_gather("dense_fwd_gathered", table, field_off, ids, F, E)) return rc;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "dense_fwd_gathered", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;

  MI_REQUIRE(M >= 0 && M <= INT32_MAX && N > 0, "dense_fwd_gathered: M=%lld N=%d", (long long)M, N);
  if (M == 0) return MI_OK;
  MI_REQUIRE(W && Y && ldy >= N, "dense_fwd_gathered: null buffer / ldy");
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dense_fwd_gathered: keep_prob=%f", keep_prob);
  const int K = F * E;
  GemmArgs a{};
  a.A = table; a.lda = 0; a.B = W; a.ldb = N; a.C = Y; a.ldc = ldy;
  a.M = (int)M; a.N = N; a.K = K; a.k_per_split = ((K + BK - 1) / BK) * BK;
  a.vecA = 1; a.vecB = vec_ok(W, N, N);
  a.epi = EPI_BIAS_ACT; a.bias = bias; a.relu = relu;
  a.keep_prob = keep_prob; a.keep_div = keep_prob; a.seed = seed; a.st = mi::step_state();
  a.g_ids = ids; a.g_off = field_off; a.g_F = F; a.g_E = E; a.g_ts = ts;
  set_amax(a, amax);
  return launch<KC, MC, false, true>(a, 1, mi::as_stream(stream), "dense_fwd_gathered");
}

size_t mi_colsum_workspace_bytes(int64_t M, int32_t N) {
  return static_cast<size_t>(mi::ceil_div(M > 0 ? M : 1, kColsumRows)) * N * sizeof(float);
}

int32_t mi_colsum(const float* X, int64_t ldx, int64_t M, int32_t N, float* out, void* workspace,
                  size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(M > 0 && N > 0 && X && out && workspace && ldx >= N, "colsum: M=%lld N=%d", (long long)M, N);
  if (workspace_bytes < mi_colsum_workspace_bytes(M, N)) {
    mi::set_error("colsum: workspace %zu < %zu", workspace_bytes, mi_colsum_workspace_bytes(M, N));
    return MI_ERR_WORKSPACE;
  }
  const int64_t nparts = mi::ceil_div(M, kColsumRows);
  MI_REQUIRE(nparts <= 65535, "colsum: M=%lld too large", (long long)M);
  hipStream_t st = mi::as_stream(stream);
  float* part = static_cast<float*>(workspace);
  colsum_part_k<<<dim3((unsigned)mi::ceil_div(N, 64), (unsigned)nparts), dim3(kThreads), 0, st>>>(X, ldx, M, N, part);
  MI_CHECK_LAUNCH("colsum(part)");
  colsum_final_k<<<dim3((unsigned)mi::ceil_div(N, kThreads)), dim3(kThreads), 0, st>>>(part, (int)nparts, N, out);
  MI_CHECK_LAUNCH("colsum(final)");
  return MI_OK;
}

size_t mi_dense_bwd_weight_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  const int splits = wgrad_splits(M, N, K);
  return (static_cast<size_t>(splits) * K * N + static_cast<size_t>(splits) * N) * sizeof(float) + 256;
}

static int32_t bwd_weight_impl(const float* X, int64_t ldx, const float* dY, int64_t lddy, float* dW, float* db,
                               int64_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes,
                               const mi_gemm_amax_t* amax, mi_stream_t stream, const int32_t* g_ids,
                               const int64_t* g_off, int32_t g_F, int32_t g_E, int64_t g_ts);

int32_t mi_dense_bwd_weight(const float* X, int64_t ldx, const float* dY, int64_t lddy, float* dW,
                            float* db, int64_t M, int32_t N, int32_t K, void* workspace,
                            size_t workspace_bytes, const mi_gemm_amax_t* amax, mi_stream_t stream) {
  MI_REQUIRE(X && ldx >= K, "dense_bwd_weight: X / ldx");
  return bwd_weight_impl(X, ldx, dY, lddy, dW, db, M, N, K, workspace, workspace_bytes, amax, stream, nullptr, nullptr,
                         0, 0, 0);
}

int32_t mi_dense_bwd_weight_gathered(const float* table, const int64_t* field_off, const int32_t* ids, int32_t F,
                                     int32_t E, const float* dY, int64_t lddy, float* dW, float* db, int64_t M,
                                     int32_t N, void* workspace, size_t workspace_bytes,
                                     const mi_gemm_amax_t* amax, int64_t table_stride, mi_stream_t stream) {
  if (int32_t rc = check_gather("dense_bwd_weight_gathered", table, field_off, ids, F, E)) return rc;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "dense_bwd_weight_gathered", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;
  return bwd_weight_impl(table, 0, dY, lddy, dW, db, M, N, F * E, workspace, workspace_bytes, amax, stream, ids,
                         field_off, F, E, ts);
}

static int32_t bwd_weight_impl(const float* X, int64_t ldx, const float* dY, int64_t lddy, float* dW, float* db,
                               int64_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes,
                               const mi_gemm_amax_t* amax, mi_stream_t stream, const int32_t* g_ids,
                               const int64_t* g_off, int32_t g_F, int32_t g_E, int64_t g_ts) {
  MI_REQUIRE(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "dense_bwd_weight: M=%lld N=%d K=%d", (long long)M, N, K);
  MI_REQUIRE(X && dY && dW && workspace, "dense_bwd_weight: null buffer");
  MI_REQUIRE(lddy >= N, "dense_bwd_weight: leading dimensions");
  MI_REQUIRE(mi::aligned16(workspace), "dense_bwd_weight: workspace must be 16-byte aligned");
  if (workspace_bytes < mi_dense_bwd_weight_workspace_bytes(M, N, K)) {
    mi::set_error("dense_bwd_weight: workspace %zu < %zu", workspace_bytes,
                  mi_dense_bwd_weight_workspace_bytes(M, N, K));
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  const int splits = wgrad_splits(M, N, K);
  const int64_t n = static_cast<int64_t>(K) * N;
  float* slab = static_cast<float*>(workspace);
  float* cpart = slab + static_cast<int64_t>(splits) * n;   // [splits][N] bias-gradient partials
  if (N == 1 && !g_ids && gemv_ok(X, ldx, X, K)) {          // logits layer: slabs of rows, then the same reduce
    const int64_t rows = mi::ceil_div(M, splits);
    const bool one = splits == 1;       // one slab IS the result (the small-batch step saves the reduce launch)
    gemv_wgrad_k<<<dim3((unsigned)mi::ceil_div(K, 16), (unsigned)splits), dim3(kThreads), 0, st>>>(
        X, ldx, dY, lddy, M, K, rows, one ? dW : slab, db ? (one ? db : cpart) : nullptr);
    MI_CHECK_LAUNCH("dense_bwd_weight(N = 1)");
    if (one) return MI_OK;
    slab_reduce(slab, splits, n, dW, cpart, db ? 1 : 0, db, st);
    MI_CHECK_LAUNCH("dense_bwd_weight(reduce)");
    return MI_OK;
  }
  GemmArgs a{};                       // dW[K,N] = X[M,K]^T * dY[M,N] : gemm K x N x (reduce M)
  a.A = X; a.lda = ldx; a.B = dY; a.ldb = lddy;
  a.M = K; a.N = N; a.K = (int)M; a.k_per_split = (int)wgrad_k_per_split(M, splits);
  a.vecA = vec_ok(X, ldx, K); a.vecB = vec_ok(dY, lddy, N);
  // one split: the "slab" IS the result (the small-batch step saves the two reduce launches per layer)
  const bool direct = splits == 1;
  a.C = direct ? dW : slab; a.ldc = N; a.epi = EPI_SLAB; a.keep_prob = 1.f; a.keep_div = 1.f;
  a.colsum_part = db ? (direct ? db : cpart) : nullptr;
  set_amax(a, amax);
  a.amax_c = nullptr;                 // the slabs are partial sums; dW is nobody's matrix operand
  if (g_ids) {
    a.vecA = 1; a.g_ids = g_ids; a.g_off = g_off; a.g_F = g_F; a.g_E = g_E; a.g_ts = g_ts;
    if (int32_t rc = launch<MC, MC, true, true>(a, splits, st, "dense_bwd_weight_gathered(split-K)")) return rc;
  } else if (int32_t rc = launch<MC, MC, true>(a, splits, st, "dense_bwd_weight(split-K)")) return rc;
  if (direct) return MI_OK;
  slab_reduce(slab, splits, n, dW, cpart, db ? N : 0, db, st);
  MI_CHECK_LAUNCH("dense_bwd_weight(reduce)");
  return MI_OK;
}


// two kernels: wgrad_pl.hip (LDS-DMA staging, transposed LDS reads; N = 128 / 256 / 512) and, for the other whole-tile
// shapes, gemm_wgrad_pl_k (register staging).  (MI_WGRAD_PL=0 in the tools' build forces the second: A/B runs.)
static bool wgrad_plan(int64_t M, int32_t N, int32_t K, mi::WgradPlPlan* p) {
  return mi::env_int("MI_WGRAD_PL", 1) != 0 && mi::wgrad_pl_plan(M, N, K, p);
}

size_t mi_dense_bwd_weight_planes_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  mi::WgradPlPlan p;
  size_t slabs = mi_dense_bwd_weight_workspace_bytes(M, N, K);
  if (wgrad_plan(M, N, K, &p)) {
    const size_t s2 = (static_cast<size_t>(p.splits) * K * N + static_cast<size_t>(p.splits) * N) * sizeof(float) + 256;
    if (s2 > slabs) slabs = s2;
  }
  return slabs + static_cast<size_t>(M > 0 ? M : 0) * 12 + 256;   // + the per-example factors
}

// One layer's weight gradient from planes, in three stages that a batch of layers shares launch by launch: the per-example
// factors (wgrad_scale_k: ONE launch for all layers), the split-K GEMM (one launch per layer), the slab folds (ONE launch).
struct WgradJobPlan {
  const mi_planes_t* X; const mi_planes_t* dY; float* dW; float* db; int64_t M; int N, K; const float* amax_a; const float* amax_b;
  bool dma; mi::WgradPlPlan plan; int splits; int64_t n; bool direct;
  char* fac; float* slab; float* cpart; float* out; float* cout;
};
static int32_t wgrad_job_plan(const mi_planes_t* X, const mi_planes_t* dY, float* dW, float* db, int64_t M, int32_t N, int32_t K,
                              void* workspace, size_t workspace_bytes, const mi_gemm_amax_t* amax, WgradJobPlan* jp) {
  MI_REQUIRE(M > 0 && M <= INT32_MAX && N > 0 && K > 0, "dense_bwd_weight_planes: M=%lld N=%d K=%d", (long long)M, N, K);
  MI_REQUIRE(M % BK == 0 && N % BN == 0 && K % BM == 0,
             "dense_bwd_weight_planes: M=%lld N=%d K=%d (examples a multiple of 32, N and K multiples of 128)", (long long)M, N, K);
  MI_REQUIRE(X && dY && X->data && dY->data && X->row_exp && dY->row_exp && dW && workspace, "dense_bwd_weight_planes: null buffer");
  MI_REQUIRE(mi::aligned16(X->data) && mi::aligned16(dY->data) && X->blk_stride >= 64 * M && dY->blk_stride >= 64 * M &&
                 X->blk_stride % 64 == 0 && dY->blk_stride % 64 == 0 && X->blk_stride < (1 << 28) && dY->blk_stride < (1 << 28),
             "dense_bwd_weight_planes: planes (16-byte aligned, 64 M <= blk_stride < 2^28, a multiple of 64)");
  MI_REQUIRE(amax && amax->a && amax->b, "dense_bwd_weight_planes: needs the abs-max vectors of X and dY");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 31u) == 0, "dense_bwd_weight_planes: workspace must be 32-byte aligned");
  if (workspace_bytes < mi_dense_bwd_weight_planes_workspace_bytes(M, N, K)) {
    mi::set_error("dense_bwd_weight_planes: workspace %zu < %zu", workspace_bytes,
                  mi_dense_bwd_weight_planes_workspace_bytes(M, N, K));
    return MI_ERR_WORKSPACE;
  }
  WgradJobPlan& p = *jp;
  p.X = X; p.dY = dY; p.dW = dW; p.db = db; p.M = M; p.N = N; p.K = K; p.amax_a = amax->a; p.amax_b = amax->b;
  p.dma = wgrad_plan(M, N, K, &p.plan);
  p.splits = p.dma ? p.plan.splits : wgrad_splits(M, N, K);
  p.n = static_cast<int64_t>(K) * N;
  p.fac = static_cast<char*>(workspace);                                   // per-example factors first (alignment)
  p.slab = reinterpret_cast<float*>(p.fac + ((static_cast<size_t>(M) * 12 + 255) & ~size_t(255)));
  p.cpart = p.slab + static_cast<int64_t>(p.splits) * p.n;                 // [splits][N] bias-gradient partials
  p.direct = p.splits == 1;           // one split: the "slab" IS the result
  p.out = p.direct ? dW : p.slab;
  p.cout = db ? (p.direct ? db : p.cpart) : nullptr;
  return MI_OK;
}
static WgScaleJob wgrad_scale_job(const WgradJobPlan& p, int block0) {
  WgScaleJob j{};
  j.x_exp = p.X->row_exp; j.y_exp = p.dY->row_exp; j.amax_a = p.amax_a; j.amax_b = p.amax_b; j.M = p.M; j.block0 = block0;
  if (p.dma) {
    j.sx16 = reinterpret_cast<uint16_t*>(p.fac); j.sy16 = j.sx16 + p.M; j.sc16 = j.sy16 + p.M;
    j.kflag = reinterpret_cast<int32_t*>(j.sc16 + p.M);                    // [M / 16]
  } else {
    j.scw = reinterpret_cast<uint32_t*>(p.fac); j.scy = j.scw + p.M; j.yf = reinterpret_cast<float*>(j.scy + p.M);
  }
  return j;
}
static int32_t wgrad_job_gemm(const WgradJobPlan& p, hipStream_t st) {
  const int64_t M = p.M;
  if (p.dma) {
    const uint16_t* sx16 = reinterpret_cast<const uint16_t*>(p.fac);
    const uint16_t* sy16 = sx16 + M;
    const uint16_t* sc16 = sy16 + M;
    const int32_t* kflag = reinterpret_cast<const int32_t*>(sc16 + M);
    return mi::wgrad_pl_launch(p.plan, p.X, p.dY, sx16, sy16, sc16, kflag, p.amax_a, p.amax_b, p.out, p.cout, M, p.N, p.K, st);
  }
  WgPlArgs wa{};
  wa.A = static_cast<const char*>(p.X->data); wa.bsa = p.X->blk_stride;
  wa.B = static_cast<const char*>(p.dY->data); wa.bsb = p.dY->blk_stride;
  wa.scw = reinterpret_cast<const uint32_t*>(p.fac); wa.scy = wa.scw + M; wa.yf = reinterpret_cast<const float*>(wa.scy + M);
  GemmArgs& a = wa.g;                 // dW[K,N] = X[M,K]^T * dY[M,N] : gemm K x N x (reduce M)
  a.M = p.K; a.N = p.N; a.K = (int)M; a.k_per_split = (int)wgrad_k_per_split(M, p.splits);
  a.C = p.out; a.ldc = p.N; a.epi = EPI_SLAB; a.keep_prob = 1.f; a.keep_div = 1.f;
  a.colsum_part = p.cout;
  a.amax_a = p.amax_a; a.amax_b = p.amax_b; a.amax_c = nullptr;
  a.tiles_m = p.K / BM; a.tiles_n = p.N / BN;
  const int64_t nblocks = static_cast<int64_t>(a.tiles_m) * a.tiles_n * p.splits;
  MI_REQUIRE(nblocks <= INT32_MAX, "dense_bwd_weight_planes: grid too large");
  if (p.db) gemm_wgrad_pl_k<true><<<dim3((unsigned)nblocks), dim3(kThreads), 0, st>>>(wa);
  else gemm_wgrad_pl_k<false><<<dim3((unsigned)nblocks), dim3(kThreads), 0, st>>>(wa);
  MI_CHECK_LAUNCH("dense_bwd_weight_planes(split-K)");
  return MI_OK;
}

int32_t mi_dense_bwd_weight_planes_batch(const mi_wgrad_job_t* jobs, int32_t n_jobs, int64_t M, void* workspace, size_t workspace_bytes,
                                         mi_stream_t stream) {
  MI_REQUIRE(jobs && n_jobs > 0 && n_jobs <= MI_MAX_WEIGHT_JOBS, "dense_bwd_weight_planes_batch: n_jobs=%d (1..%d)", n_jobs, MI_MAX_WEIGHT_JOBS);
  hipStream_t st = mi::as_stream(stream);
  WgradJobPlan plans[MI_MAX_WEIGHT_JOBS];
  WgScaleJobs sj{};
  FoldJobs fj{};
  size_t off = 0;
  int sblocks = 0, fblocks = 0;
  for (int q = 0; q < n_jobs; ++q) {
    const mi_wgrad_job_t& u = jobs[q];
    const size_t need = mi_dense_bwd_weight_planes_workspace_bytes(M, u.N, u.K);
    if (off + need > workspace_bytes) {
      mi::set_error("dense_bwd_weight_planes_batch: workspace %zu too small (job %d ends at %zu)", workspace_bytes, q, off + need);
      return MI_ERR_WORKSPACE;
    }
    if (int32_t rc = wgrad_job_plan(&u.X, &u.dY, u.dW, u.db, M, u.N, u.K, static_cast<char*>(workspace) + off, need, &u.amax, &plans[q])) return rc;
    off += (need + 255) & ~size_t(255);
    sj.j[q] = wgrad_scale_job(plans[q], sblocks);
    sblocks += static_cast<int>(mi::ceil_div(M, kThreads));
    if (!plans[q].direct) {
      FoldJob& f = fj.j[fj.n];
      f.slab = plans[q].slab; f.nsplit = plans[q].splits; f.n = plans[q].n; f.out = u.dW;
      f.slab2 = u.db ? plans[q].cpart : nullptr; f.n2 = u.db ? u.N : 0; f.out2 = u.db;
      f.vec = fold_vec_ok(f.slab, f.n, f.out) ? 1 : 0; f.block0 = fblocks;
      fblocks += static_cast<int>(fold_blocks(f));
      ++fj.n;
    }
  }
  sj.n = n_jobs;
  wgrad_scale_k<<<dim3((unsigned)sblocks), dim3(kThreads), 0, st>>>(sj);
  MI_CHECK_LAUNCH("dense_bwd_weight_planes(scales)");
  for (int q = 0; q < n_jobs; ++q)
    if (int32_t rc = wgrad_job_gemm(plans[q], st)) return rc;
  if (fj.n > 0) {
    slab_reduce_multi_k<<<dim3((unsigned)fblocks), dim3(kThreads), 0, st>>>(fj);
    MI_CHECK_LAUNCH("dense_bwd_weight_planes(reduce)");
  }
  return MI_OK;
}

size_t mi_dense_bwd_weight_planes_batch_workspace_bytes(const mi_wgrad_job_t* jobs, int32_t n_jobs, int64_t M) {
  size_t tot = 0;
  for (int q = 0; jobs && q < n_jobs; ++q) tot += (mi_dense_bwd_weight_planes_workspace_bytes(M, jobs[q].N, jobs[q].K) + 255) & ~size_t(255);
  return tot + 256;
}

int32_t mi_dense_bwd_weight_planes(const mi_planes_t* X, const mi_planes_t* dY, float* dW, float* db, int64_t M,
                                   int32_t N, int32_t K, void* workspace, size_t workspace_bytes,
                                   const mi_gemm_amax_t* amax, mi_stream_t stream) {
  MI_REQUIRE(X && dY && amax, "dense_bwd_weight_planes: null argument");
  mi_wgrad_job_t job{};
  job.X = *X; job.dY = *dY; job.dW = dW; job.db = db; job.N = N; job.K = K; job.amax = *amax;
  return mi_dense_bwd_weight_planes_batch(&job, 1, M, workspace, workspace_bytes, stream);
}

}  // extern "C"


#ifdef MI_GEMM_STAMPS
extern "C" int32_t mi_gemm_stamps_read(void* dst, size_t nbytes) {
  return static_cast<int32_t>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), nbytes, 0, hipMemcpyDeviceToHost));
}
#endif
