// The logits layer and the head of a TRAIN step in one pass over the last hidden layer's output.
//
// Replaces, for the training step, the five launches round 3 spent on trainers/deep_fm.py:108 (`tf.layers.dense(net, 1)`),
// :111 (`logits += dnn_logits`), :118-125 (the sigmoid cross-entropy head) and their gradients: mi_dense_fwd's N = 1
// form (gemv_fwd_k), mi_sigmoid_ce_head (head_k + sum_partials_k), mi_dense_bwd_weight's N = 1 form (gemv_wgrad_k +
// slab_reduce_k) and mi_dense_bwd_data_vec_planes — ~55 us and four launch gaps at config 3 for 33 MB read twice and
// 33 MB written.  Here the layer's input X [M][K] is read ONCE: a group of 16 lanes owns an example,
//     dnn[m]     = X[m,:] . w + b                         (the order of gemv_fwd_k: 16 lanes, float4 pieces 64 k apart, xor tree)
//     logits[m]  = lin[m] + lin_bias + fm[m] + dnn[m]     (head_k's order)
//     loss      += (max(x, 0) - x y + log1p(exp(-|x|))) * scale,   d[m] = (sigmoid(x) - y) * scale
//     dX[m][k]   = d[m] * w[k], kept where mask bit k is set, divided by keep_prob  -> planes (vec_dgrad_planes_k's arithmetic)
//     dW[k]     += X[m][k] * d[m],  db += d[m]            (per-block partial sums, folded in block order by tail_fold_k)
// Two launches (this kernel + a fold of the blocks' (K + 2) partials).  HBM bound: 4 K + K / 8 + ~20 bytes read and 4 K + ~16
// written per example.  Results are reproducible (fixed orders); the loss / dW sums associate differently from the
// unfused sequence (tests hold them to 1e-6 of it), everything per-example is the unfused sequence's bits.
#include "common.h"
#include <algorithm>

namespace {

constexpr int kBlock = 256;
constexpr int kLpr = 16;                      // lanes per example
constexpr int kGroups = kBlock / kLpr;        // examples per pass of a block
constexpr int ROWB = 64;                      // bytes of a plane row piece: 16 hi | 16 lo

typedef _Float16 t_h16x2 __attribute__((ext_vector_type(2)));
typedef float t_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float t_pow2(int s) { return __uint_as_float(static_cast<uint32_t>(127 + s) << 23); }
__device__ __forceinline__ int t_exp_for(float amax) {      // gemm_pl.hip's pl_exp_for
  const int e = static_cast<int>((__float_as_uint(amax) >> 23) & 0xffu);
  return max(-100, min(100, 141 - e));
}
__device__ __forceinline__ float t_sigmoid(float x) {
  const float e = expf(-fabsf(x));
  return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

struct TailArgs {
  const float* X; int64_t ldx; const float* w; const float* b;
  const float* lin; const float* lin_bias; const float* fm; const uint8_t* labels;
  int64_t M; int K; float scale;
  const uint32_t* mbits; int64_t mbld; float keep_div, keep_rcp;
  float* dnn; float* logits; float* d_logit;
  char* out; int64_t ldo_b; int32_t* row_exp;     // planes of dX
  float* dX; int64_t lddx; float* amax_out;
  float* part;                                     // [blocks][K + 2]: dW partial, sum d, sum loss
  int rpb;                                         // examples per block (a multiple of kGroups)
};

// A block walks kTiles consecutive tiles of `rpb` examples (rpb = 4096 / K: a 16-KB planes image, P = rpb / 16 passes of its
// 16 lane groups); ALL of a tile's loads — its P examples' X pieces, lin, fm, labels and mask words per lane group — are
// issued before anything is computed (one memory latency per tile, not per pass), the dW / loss / d sums stay in registers
// across the tiles.  (First version: one 80-example tile per block, 48 KB of LDS = 3 blocks per CU, loads pass by pass:
// 43 us alone at config 3 against 51 for the five launches it replaces; this form: see tools/mlp_tail_bench.py.)
constexpr int kTiles = 2;

template <int Q, int P>                            // float4 pieces per lane: K = 64 Q; passes per tile: rpb = 16 P
__global__ __launch_bounds__(kBlock) void logits_head_tail_k(const TailArgs a) {
  constexpr int K = 64 * Q, RPB = kGroups * P;
  // [K / 16][RPB][64 B] planes image of a tile; the 16-k blocks 32 bytes further apart than their rows need, so that the four
  // blocks a lane group writes at once fall on different banks
  constexpr int BLK = RPB * ROWB + 32;
  __shared__ __attribute__((aligned(16))) char stage[(K >> 4) * BLK];
  __shared__ float red[kGroups][K + 2];
  const int t = threadIdx.x, l = t & (kLpr - 1), grp = t / kLpr;
  float4 w4[Q], accw[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { w4[q] = *reinterpret_cast<const float4*>(a.w + 4 * l + 64 * q); accw[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
  const float b0 = a.b ? a.b[0] : 0.f;
  const float lb = (a.lin && a.lin_bias) ? a.lin_bias[0] : 0.f;
  float acc_l = 0.f, acc_d = 0.f, bmx = 0.f;
  struct TileIn { float4 x[P][Q]; uint32_t mw[P][Q]; float lin_v[P], fm_v[P], y_v[P]; };
  // every load of a tile, issued together (rows past the end of the batch: row 0 of the tile, unused)
  auto load_tile = [&](int tile, TileIn& in) {
    const int64_t r0 = min((static_cast<int64_t>(blockIdx.x) * kTiles + tile) * RPB, a.M - 1);
    const int nrows = static_cast<int>(min(static_cast<int64_t>(RPB), a.M - r0));
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rl = grp + kGroups * p;
      const int64_t r = r0 + (rl < nrows ? rl : 0);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int k = 4 * l + 64 * q;
        in.x[p][q] = *reinterpret_cast<const float4*>(a.X + r * a.ldx + k);
        in.mw[p][q] = a.mbits ? a.mbits[r * a.mbld + (k >> 5)] >> (k & 31) : 0xfu;
      }
      in.lin_v[p] = a.lin ? a.lin[r] : 0.f;
      in.fm_v[p] = a.fm ? a.fm[r] : 0.f;
      in.y_v[p] = a.labels[r] ? 1.f : 0.f;
    }
  };
  TileIn cur, nxt;
  load_tile(0, cur);
#pragma unroll
  for (int tile = 0; tile < kTiles; ++tile) {
    const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * kTiles + tile) * RPB;
    if (r0 >= a.M) break;                                            // (block-uniform)
    const int nrows = static_cast<int>(min(static_cast<int64_t>(RPB), a.M - r0));
    if (tile + 1 < kTiles) load_tile(tile + 1, nxt);                 // the next tile's loads travel under this tile's arithmetic
    const auto& x = cur.x; const auto& mw = cur.mw; const auto& lin_v = cur.lin_v; const auto& fm_v = cur.fm_v; const auto& y_v = cur.y_v;
    if (tile > 0) __syncthreads();                                   // the previous tile's planes have left the stage
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rl = grp + kGroups * p;
      const bool on = rl < nrows;
      const int64_t r = r0 + (on ? rl : 0);
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) acc += (x[p][q].x * w4[q].x + x[p][q].y * w4[q].y) + (x[p][q].z * w4[q].z + x[p][q].w * w4[q].w);
#pragma unroll
      for (int o = kLpr / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kLpr);
      const float dnn = acc + b0;
      float z = 0.f;
      if (a.lin) z += lin_v[p] + lb;
      if (a.fm) z += fm_v[p];
      z += dnn;
      const float y = y_v[p];
      const float loss = (fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)))) * a.scale;
      const float g = (t_sigmoid(z) - y) * a.scale;
      if (on && l == 0) {
        if (a.dnn) a.dnn[r] = dnn;
        a.logits[r] = z;
        a.d_logit[r] = g;
        acc_l += loss; acc_d += g;
      }
      // the layer's data gradient, masked, and its abs-max over the example
      float4 v[Q];
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int k = 4 * l + 64 * q;
        v[q] = make_float4(g * w4[q].x, g * w4[q].y, g * w4[q].z, g * w4[q].w);
        if (a.mbits) {
          const uint32_t m = mw[p][q];
          v[q].x = (m & 1u) ? mi_div_const(v[q].x, a.keep_div, a.keep_rcp) : 0.f; v[q].y = (m & 2u) ? mi_div_const(v[q].y, a.keep_div, a.keep_rcp) : 0.f;
          v[q].z = (m & 4u) ? mi_div_const(v[q].z, a.keep_div, a.keep_rcp) : 0.f; v[q].w = (m & 8u) ? mi_div_const(v[q].w, a.keep_div, a.keep_rcp) : 0.f;
        }
        if (on && a.dX) *reinterpret_cast<float4*>(a.dX + r * a.lddx + k) = v[q];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[q].x), fabsf(v[q].y))), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
        if (on) { accw[q].x += x[p][q].x * g; accw[q].y += x[p][q].y * g; accw[q].z += x[p][q].z * g; accw[q].w += x[p][q].w * g; }
      }
#pragma unroll
      for (int o = kLpr / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, kLpr));
      if (on) {
        bmx = fmaxf(bmx, mx);
        const int s = t_exp_for(mx);
        const float sc = t_pow2(s);
        if (l == 0) a.row_exp[r] = s;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float u[4] = {v[q].x * sc, v[q].y * sc, v[q].z * sc, v[q].w * sc};
          uint32_t ph[2], pq[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const t_f32x2 uu = {u[2 * e], u[2 * e + 1]};
            t_h16x2 hh = __builtin_convertvector(uu, t_h16x2);
            uint32_t hb = __builtin_bit_cast(uint32_t, hh);
            if (uu[0] > 0.f && (hb & 0xffffu) == 0u) hb |= 1u;          // positive stays positive in the high plane
            if (uu[1] > 0.f && (hb >> 16) == 0u) hb |= 0x10000u;
            hh = __builtin_bit_cast(t_h16x2, hb);
            const t_f32x2 rr2 = {uu[0] - static_cast<float>(hh[0]), uu[1] - static_cast<float>(hh[1])};
            ph[e] = hb;
            pq[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr2, t_h16x2));
          }
          const int qq = l + kLpr * q;                                   // float4 index inside the row: k = 4 qq
          char* d = stage + (qq >> 2) * BLK + rl * ROWB + (qq & 3) * 8;
          *reinterpret_cast<uint2*>(d) = make_uint2(ph[0], ph[1]);
          *reinterpret_cast<uint2*>(d + 32) = make_uint2(pq[0], pq[1]);
        }
      }
    }
    __syncthreads();
    // the planes of the tile's rows: one contiguous run per 16-k block
    // (all of a thread's pieces are read from LDS first, then stored: (K / 16) * RPB * 4 / 256 independent 16-byte stores)
    const int run16 = nrows * 4;
    constexpr int NPC = (K >> 4) * RPB * 4 / kBlock;
    uint4 pc[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int idx = j * kBlock + t, kb = idx / (RPB * 4), pp = idx % (RPB * 4);
      pc[j] = *reinterpret_cast<const uint4*>(stage + kb * BLK + pp * 16);
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int idx = j * kBlock + t, kb = idx / (RPB * 4), pp = idx % (RPB * 4);
      if (pp < run16) *reinterpret_cast<uint4*>(a.out + kb * a.ldo_b + r0 * ROWB + pp * 16) = pc[j];
    }
    if (tile + 1 < kTiles) cur = nxt;
  }
  // the block's partial sums: dW over its examples (the 16 lane groups in group order), sum d, sum loss
#pragma unroll
  for (int q = 0; q < Q; ++q) *reinterpret_cast<float4*>(&red[grp][4 * l + 64 * q]) = accw[q];
  if (l == 0) { red[grp][K] = acc_d; red[grp][K + 1] = acc_l; }
  __syncthreads();
  for (int k = t; k < K + 2; k += kBlock) {
    float s = 0.f;
#pragma unroll
    for (int gq = 0; gq < kGroups; ++gq) s += red[gq][k];
    a.part[static_cast<int64_t>(blockIdx.x) * (K + 2) + k] = s;
  }
  if (a.amax_out) mi_amax_publish(a.amax_out, bmx);                  // (every thread of the block reaches this)
}

// out[k] = the sum of the blocks' partials: 16 columns x 16 slices of the partials per workgroup (slice s: partials s, s + 16,
// ... in order), the slices folded in order — dW [K], then the sum of d (into db and / or d_sum) and the loss
__global__ __launch_bounds__(kBlock) void tail_fold_k(const float* __restrict__ part, int nparts, int K, float* __restrict__ dW,
                                                      float* __restrict__ db, float* __restrict__ d_sum, float* __restrict__ loss) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + c;
  float acc = 0.f;
  if (k < K + 2) {
#pragma unroll 8
    for (int s = sl; s < nparts; s += 16) acc += part[static_cast<int64_t>(s) * (K + 2) + k];
  }
  red[sl][c] = acc;
  __syncthreads();
  if (sl == 0 && k < K + 2) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][c];
    if (k < K) dW[k] = v;
    else if (k == K) { if (db) db[0] = v; if (d_sum) d_sum[0] = v; }
    else if (loss) loss[0] = v;
  }
}

int tail_rows_per_block(int K) { return kTiles * (4096 / K); }     // kTiles tiles of 4096 / K examples (a 16-KB planes image)

}  // namespace

extern "C" {

size_t mi_logits_head_fused_workspace_bytes(int64_t M, int32_t K) {
  if (M <= 0 || K <= 0) return 256;
  const int64_t nb = mi::ceil_div(M, tail_rows_per_block(K));
  return static_cast<size_t>(nb) * (K + 2) * sizeof(float) + 256;
}

int32_t mi_logits_head_fused(const float* X, int64_t ldx, const float* w, const float* b, const float* lin, const float* lin_bias,
                             const float* fm, const uint8_t* labels, int64_t M, int32_t K, float loss_scale,
                             const uint32_t* mask_bits, int64_t mask_ld, float keep_prob, float* dnn, float* logits,
                             float* loss_out, float* d_logit, float* d_logit_sum, float* dW, float* db, const mi_planes_t* dXp,
                             float* dX, int64_t lddx, float* amax_out, void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(M > 0 && (K == 64 || K == 128 || K == 256), "logits_head_fused: M=%lld K=%d (K in {64, 128, 256})", (long long)M, K);
  MI_REQUIRE(X && w && labels && logits && d_logit && dW && dXp && workspace, "logits_head_fused: null buffer");
  MI_REQUIRE(ldx >= K && (ldx & 3) == 0 && mi::aligned16(X) && mi::aligned16(w), "logits_head_fused: X / w leading dimension or alignment");
  MI_REQUIRE(!dX || (mi::aligned16(dX) && lddx >= K && (lddx & 3) == 0), "logits_head_fused: dX leading dimension / alignment");
  MI_REQUIRE(dXp->data && dXp->row_exp && mi::aligned16(dXp->data) && dXp->blk_stride >= M * ROWB && (dXp->blk_stride & 63) == 0,
             "logits_head_fused: output planes");
  MI_REQUIRE(!mask_bits || mask_ld >= (K + 31) / 32, "logits_head_fused: mask_ld=%lld", (long long)mask_ld);
  MI_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "logits_head_fused: keep_prob=%f", keep_prob);
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15u) == 0, "logits_head_fused: workspace alignment");
  if (workspace_bytes < mi_logits_head_fused_workspace_bytes(M, K)) {
    mi::set_error("logits_head_fused: workspace %zu < %zu", workspace_bytes, mi_logits_head_fused_workspace_bytes(M, K));
    return MI_ERR_WORKSPACE;
  }
  TailArgs a{};
  a.X = X; a.ldx = ldx; a.w = w; a.b = b; a.lin = lin; a.lin_bias = lin_bias; a.fm = fm; a.labels = labels;
  a.M = M; a.K = K; a.scale = loss_scale;
  a.mbits = mask_bits; a.mbld = mask_ld; a.keep_div = mask_bits ? keep_prob : 1.f; a.keep_rcp = 1.0f / a.keep_div;
  a.dnn = dnn; a.logits = logits; a.d_logit = d_logit;
  a.out = static_cast<char*>(dXp->data); a.ldo_b = dXp->blk_stride; a.row_exp = dXp->row_exp;
  a.dX = dX; a.lddx = lddx; a.amax_out = amax_out;
  a.part = static_cast<float*>(workspace);
  a.rpb = tail_rows_per_block(K);
  const int64_t nb = mi::ceil_div(M, a.rpb);
  MI_REQUIRE(nb <= INT32_MAX, "logits_head_fused: grid too large");
  hipStream_t st = mi::as_stream(stream);
  const dim3 g(static_cast<unsigned>(nb)), blk(kBlock);
  switch (K) {
    case 64: logits_head_tail_k<1, 4><<<g, blk, 0, st>>>(a); break;
    case 128: logits_head_tail_k<2, 2><<<g, blk, 0, st>>>(a); break;
    default: logits_head_tail_k<4, 1><<<g, blk, 0, st>>>(a); break;
  }
  MI_CHECK_LAUNCH("logits_head_fused");
  tail_fold_k<<<dim3(static_cast<unsigned>(mi::ceil_div(K + 2, 16))), blk, 0, st>>>(a.part, static_cast<int>(nb), K, dW, db, d_logit_sum, loss_out);
  MI_CHECK_LAUNCH("logits_head_fused(fold)");
  return MI_OK;
}

}  // extern "C"
