// Optimizer applies: dense (flat parameter buffer), sparse (unique embedding / linear rows with
// fused duplicate-summing) and the lazy catch-up that makes sparse Adam equal to TF's
// whole-table sweep.
//
// Replaces get_optimizer + optimizer.minimize (trainers/model_utils.py:57-72, invoked through
// head.create_estimator_spec at trainers/deep_fm.py:117-125) for tf.train.{Adam, Adagrad, Ftrl,
// RMSProp, GradientDescent}Optimizer and the canned estimators' Ftrl / Adagrad
// (trainers/linear.py:30, deep.py:32, linear_deep.py:32).  Update rules: SURVEY Appendix A.6/A.7.
//
// This file is compiled with -ffp-contract=off: each expression below is written in the order of
// TF's Eigen expressions, one rounding per operation, so fp32 results equal the numpy oracle's
// (oracle/optimizers.py) bit for bit given equal gradients.  HBM-bound; per unique row the
// algorithmic traffic is 4E (grad) + 24E (w, slot0, slot1 read+write) + 8 bytes.
#include "common.h"

namespace {

constexpr int kBlock = 256;

struct Hp {
  int kind;
  float lr, beta1, beta2, eps, lr_t, decay, momentum, lr_power, l1, l2;
};

Hp make_hp(const mi_opt_hparams* h) {
  return Hp{h->kind, h->lr, h->beta1, h->beta2, h->epsilon, h->lr_t, h->decay, h->momentum,
            h->lr_power, h->l1, h->l2};
}

// one element of a dense variable (training_ops.cc Apply* functors)
__device__ __forceinline__ void dense_rule(const Hp& h, float& w, float& s0, float& s1, float g) {
  switch (h.kind) {
    case MI_OPT_ADAM: {
      s0 = s0 + (g - s0) * (1.f - h.beta1);
      s1 = s1 + (g * g - s1) * (1.f - h.beta2);
      w = w - (s0 * h.lr_t) / (sqrtf(s1) + h.eps);
    } break;
    case MI_OPT_ADAGRAD: {
      s0 = s0 + g * g;
      w = w - (g * h.lr) * (1.f / sqrtf(s0));
    } break;
    case MI_OPT_FTRL: {
      const float na = s0 + g * g;
      s1 = s1 + (g - ((sqrtf(na) - sqrtf(s0)) / h.lr) * w);
      const float adj = fminf(fmaxf(s1, -h.l1), h.l1);
      w = (adj - s1) / (sqrtf(na) / h.lr + 2.f * h.l2);
      s0 = na;
    } break;
    case MI_OPT_RMSPROP: {
      s0 = s0 + (g * g - s0) * (1.f - h.decay);
      s1 = s1 * h.momentum + (g * h.lr) / sqrtf(s0 + h.eps);
      w = w - s1;
    } break;
    default:
      w = w - g * h.lr;
  }
}

// one element of a TOUCHED row of a sparse variable.  Adam: adam.py _apply_sparse_shared
// (m*beta1 then scatter_add); the others act on touched rows exactly like the dense rule.
__device__ __forceinline__ void sparse_rule(const Hp& h, float& w, float& s0, float& s1, float g) {
  if (h.kind == MI_OPT_ADAM) {
    s0 = s0 * h.beta1 + g * (1.f - h.beta1);
    s1 = s1 * h.beta2 + (g * g) * (1.f - h.beta2);
    w = w - (h.lr_t * s0) / (sqrtf(s1) + h.eps);
  } else {
    dense_rule(h, w, s0, s1, g);
  }
}

__global__ __launch_bounds__(kBlock) void dense_apply_k(float* __restrict__ w, float* __restrict__ s0,
                                                        float* __restrict__ s1,
                                                        const float* __restrict__ g, int64_t n,
                                                        const Hp h) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    float wv = w[i], a = s0 ? s0[i] : 0.f, b = s1 ? s1[i] : 0.f;
    dense_rule(h, wv, a, b, g[i]);
    w[i] = wv;
    if (s0) s0[i] = a;
    if (s1) s1[i] = b;
  }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// One group of LPR lanes per unique row; lane l owns elements 4l..4l+3; lane 0 also owns the
// row's linear weight.  Duplicates are summed in ascending entry order.
// FUSED: the per-entry gradients are not read from d_rows / d_lin but rebuilt in place from what
// the backward left behind (single-GPU path): entry e = (b, f) contributes
//   d_concat[b, f*E:(f+1)*E] + dlf[b] * (sumv[b,:] - w)      (w = this row, still un-updated)
// to the row and dll[b] to its linear weight — mi_embed_fm_linear_bwd folded into the apply, so the
// [B*F, E] gradient matrix is never written or re-read.
struct FusedGrad {
  const float* d_concat; int64_t ldd; const float* sumv; const float* dlf; const float* dll; int F;
};

template <int LPR, bool FUSED>
__global__ __launch_bounds__(kBlock) void sparse_apply_k(
    float* __restrict__ table, float* __restrict__ t0, float* __restrict__ t1,
    float* __restrict__ lin_w, float* __restrict__ l0, float* __restrict__ l1,
    int32_t* __restrict__ last_step, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ sorted_entry,
    const int32_t* __restrict__ num_uniq, const float* __restrict__ d_rows,
    const float* __restrict__ d_lin, int E, int step, const Hp h, const FusedGrad fg) {
  const int64_t u = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  if (u >= *num_uniq) return;
  const int64_t r = uniq_rows[u];
  const int s_beg = seg_start[u], s_end = seg_start[u + 1];
  const bool lane_on = 4 * l < E;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  float gl = 0.f;
  float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
  if (table && lane_on) w = ld4(table + r * E + 4 * l);
  for (int k = s_beg; k < s_end; ++k) {
    const int64_t e = sorted_entry[k];
    if constexpr (FUSED) {
      const int64_t b = e / fg.F;
      const int f = static_cast<int>(e - b * fg.F);
      if (table && lane_on) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fg.d_concat) v = ld4(fg.d_concat + b * fg.ldd + static_cast<int64_t>(f) * E + 4 * l);
        if (fg.dlf) {
          const float gf = fg.dlf[b];
          const float4 sv = ld4(fg.sumv + b * E + 4 * l);
          v.x += gf * (sv.x - w.x); v.y += gf * (sv.y - w.y);
          v.z += gf * (sv.z - w.z); v.w += gf * (sv.w - w.w);
        }
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
      }
      if (lin_w && l == 0) gl += fg.dll[b];
    } else {
      if (table && lane_on) {
        const float4 v = ld4(d_rows + e * E + 4 * l);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
      }
      if (lin_w && l == 0) gl += d_lin[e];
    }
  }
  if (table && lane_on) {
    const int64_t o = r * E + 4 * l;
    float4 a = t0 ? ld4(t0 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 b = t1 ? ld4(t1 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    sparse_rule(h, w.x, a.x, b.x, g.x);
    sparse_rule(h, w.y, a.y, b.y, g.y);
    sparse_rule(h, w.z, a.z, b.z, g.z);
    sparse_rule(h, w.w, a.w, b.w, g.w);
    st4(table + o, w);
    if (t0) st4(t0 + o, a);
    if (t1) st4(t1 + o, b);
  }
  if (l == 0) {
    if (lin_w) {
      float w = lin_w[r], a = l0 ? l0[r] : 0.f, b = l1 ? l1[r] : 0.f;
      sparse_rule(h, w, a, b, gl);
      lin_w[r] = w;
      if (l0) l0[r] = a;
      if (l1) l1[r] = b;
    }
    if (last_step) last_step[r] = step;
  }
}

// Lazy replay of TF Adam's whole-table decay for the steps a row sat out.
__device__ __forceinline__ void replay(float& w, float& m, float& v, int s_from, int s_to,
                                       const float* __restrict__ lr_table, float b1, float b2,
                                       float eps) {
  for (int s = s_from; s <= s_to; ++s) {
    m = m * b1;
    v = v * b2;
    w = w - (lr_table[s] * m) / (sqrtf(v) + eps);
  }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void sparse_catchup_k(
    float* __restrict__ table, float* __restrict__ tm, float* __restrict__ tv,
    float* __restrict__ lin_w, float* __restrict__ lm, float* __restrict__ lv,
    int32_t* __restrict__ last_step, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, int64_t n_max, int E, int step_to,
    const float* __restrict__ lr_table, float b1, float b2, float eps) {
  const int64_t u = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  const int64_t count = uniq_rows ? static_cast<int64_t>(*num_uniq) : n_max;
  if (u >= count) return;
  const int64_t r = uniq_rows ? static_cast<int64_t>(uniq_rows[u]) : u;
  const int ls = last_step[r];
  if (ls >= step_to) return;
  // a row that was never applied has m = v = 0: every replayed step subtracts exactly 0
  if (ls > 0) {
    if (table && 4 * l < E) {
      const int64_t o = r * E + 4 * l;
      float4 w = ld4(table + o), m = ld4(tm + o), v = ld4(tv + o);
      // one loop for the four elements: one lr_t load and one loop counter per step instead of four,
      // four independent sqrt/divide chains in flight (the arithmetic per element is unchanged)
      for (int s = ls + 1; s <= step_to; ++s) {
        const float lr = lr_table[s];
        m.x = m.x * b1; m.y = m.y * b1; m.z = m.z * b1; m.w = m.w * b1;
        v.x = v.x * b2; v.y = v.y * b2; v.z = v.z * b2; v.w = v.w * b2;
        w.x = w.x - (lr * m.x) / (sqrtf(v.x) + eps);
        w.y = w.y - (lr * m.y) / (sqrtf(v.y) + eps);
        w.z = w.z - (lr * m.z) / (sqrtf(v.z) + eps);
        w.w = w.w - (lr * m.w) / (sqrtf(v.w) + eps);
      }
      st4(table + o, w); st4(tm + o, m); st4(tv + o, v);
    }
    if (lin_w && l == 0) {
      float w = lin_w[r], m = lm[r], v = lv[r];
      replay(w, m, v, ls + 1, step_to, lr_table, b1, b2, eps);
      lin_w[r] = w; lm[r] = m; lv[r] = v;
    }
  }
  // every lane of the group has read last_step[r] above (same wave, program order) before lane 0 writes
  if (l == 0) last_step[r] = step_to;
}

int lanes_per_row(int E) {
  int q = (E + 3) / 4, l = 1;
  while (l < q) l <<= 1;
  return l;
}

}  // namespace

#define MI_DISPATCH_LPR(lpr, CALL)                  \
  switch (lpr) {                                    \
    case 1: { constexpr int L = 1; CALL; } break;   \
    case 2: { constexpr int L = 2; CALL; } break;   \
    case 4: { constexpr int L = 4; CALL; } break;   \
    case 8: { constexpr int L = 8; CALL; } break;   \
    case 16: { constexpr int L = 16; CALL; } break; \
    case 32: { constexpr int L = 32; CALL; } break; \
    default: { constexpr int L = 64; CALL; } break; \
  }

static int32_t check_hp(const char* who, const mi_opt_hparams* hp) {
  if (!hp || hp->kind < MI_OPT_ADAM || hp->kind > MI_OPT_SGD) {
    mi::set_error("%s: bad optimizer kind", who);
    return MI_ERR_INVALID;
  }
  if (hp->kind == MI_OPT_FTRL && hp->lr_power != -0.5f) {
    mi::set_error("%s: Ftrl learning_rate_power %f unsupported (TF default -0.5 only)", who, hp->lr_power);
    return MI_ERR_UNSUPPORTED;
  }
  return MI_OK;
}

// y += alpha * x (gradient accumulation over the chunks of a pipelined multi-GPU step)
__global__ __launch_bounds__(kBlock) void axpy_k(float* __restrict__ y, const float* __restrict__ x, int64_t n, float alpha) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) y[i] = y[i] + alpha * x[i];
}

extern "C" {

int32_t mi_axpy(float* y, const float* x, int64_t n, float alpha, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && (n == 0 || (x && y)), "axpy: n=%lld", (long long)n);
  if (n == 0) return MI_OK;
  axpy_k<<<dim3((unsigned)mi::ceil_div(n, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(y, x, n, alpha);
  MI_CHECK_LAUNCH("axpy");
  return MI_OK;
}


int32_t mi_dense_apply(float* param, float* slot0, float* slot1, const float* grad, int64_t n,
                       const mi_opt_hparams* hp, mi_stream_t stream) {
  if (int32_t rc = check_hp("dense_apply", hp)) return rc;
  MI_REQUIRE(n >= 0, "dense_apply: n=%lld", (long long)n);
  if (n == 0) return MI_OK;
  MI_REQUIRE(param && grad, "dense_apply: null buffer");
  MI_REQUIRE(hp->kind == MI_OPT_SGD || slot0, "dense_apply: optimizer needs slot0");
  MI_REQUIRE((hp->kind != MI_OPT_ADAM && hp->kind != MI_OPT_FTRL && hp->kind != MI_OPT_RMSPROP) || slot1,
             "dense_apply: optimizer needs slot1");
  int64_t nb = mi::ceil_div(n, kBlock);
  if (nb > 4096) nb = 4096;
  dense_apply_k<<<dim3((unsigned)nb), dim3(kBlock), 0, mi::as_stream(stream)>>>(param, slot0, slot1, grad, n, make_hp(hp));
  MI_CHECK_LAUNCH("dense_apply");
  return MI_OK;
}

int32_t mi_sparse_apply(float* table, float* t_slot0, float* t_slot1, float* lin_w, float* l_slot0,
                        float* l_slot1, int32_t* last_step, const int32_t* uniq_rows,
                        const int32_t* seg_start, const int32_t* sorted_entry,
                        const int32_t* num_uniq, int64_t n_max, const float* d_rows,
                        const float* d_lin, int32_t E, int32_t step, const mi_opt_hparams* hp,
                        mi_stream_t stream) {
  if (int32_t rc = check_hp("sparse_apply", hp)) return rc;
  MI_REQUIRE(n_max >= 0, "sparse_apply: n_max=%lld", (long long)n_max);
  if (n_max == 0) return MI_OK;
  MI_REQUIRE(table || lin_w, "sparse_apply: nothing to update");
  MI_REQUIRE(uniq_rows && seg_start && sorted_entry && num_uniq, "sparse_apply: null index buffer");
  MI_REQUIRE(!table || (d_rows && E >= 4 && E <= 256 && (E & 3) == 0 && mi::aligned16(table) && mi::aligned16(d_rows)),
             "sparse_apply: table needs d_rows, E multiple of 4 in [4,256], 16-byte alignment");
  MI_REQUIRE(!lin_w || d_lin, "sparse_apply: lin_w needs d_lin");
  const bool need0 = hp->kind != MI_OPT_SGD;
  const bool need1 = hp->kind == MI_OPT_ADAM || hp->kind == MI_OPT_FTRL || hp->kind == MI_OPT_RMSPROP;
  MI_REQUIRE(!table || ((!need0 || t_slot0) && (!need1 || t_slot1)), "sparse_apply: table slots missing");
  MI_REQUIRE(!lin_w || ((!need0 || l_slot0) && (!need1 || l_slot1)), "sparse_apply: linear slots missing");
  const int lpr = table ? lanes_per_row(E) : 1;
  const int64_t blocks = mi::ceil_div(n_max * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "sparse_apply: grid too large");
  const Hp h = make_hp(hp);
  MI_DISPATCH_LPR(lpr, (sparse_apply_k<L, false><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           table, t_slot0, t_slot1, lin_w, l_slot0, l_slot1, last_step, uniq_rows, seg_start,
                           sorted_entry, num_uniq, d_rows, d_lin, E, step, h, FusedGrad{})));
  MI_CHECK_LAUNCH("sparse_apply");
  return MI_OK;
}

int32_t mi_sparse_apply_fused(float* table, float* t_slot0, float* t_slot1, float* lin_w, float* l_slot0,
                              float* l_slot1, int32_t* last_step, const int32_t* uniq_rows,
                              const int32_t* seg_start, const int32_t* sorted_entry, const int32_t* num_uniq,
                              int64_t n_max, const float* d_concat, int64_t ld_dconcat, const float* sumv,
                              const float* d_logit_fm, const float* d_logit_lin, int32_t F, int32_t E,
                              int32_t step, const mi_opt_hparams* hp, mi_stream_t stream) {
  if (int32_t rc = check_hp("sparse_apply_fused", hp)) return rc;
  MI_REQUIRE(n_max >= 0 && F > 0, "sparse_apply_fused: n_max=%lld F=%d", (long long)n_max, F);
  if (n_max == 0) return MI_OK;
  MI_REQUIRE(table || lin_w, "sparse_apply_fused: nothing to update");
  MI_REQUIRE(uniq_rows && seg_start && sorted_entry && num_uniq, "sparse_apply_fused: null index buffer");
  MI_REQUIRE(!table || (E >= 4 && E <= 256 && (E & 3) == 0 && mi::aligned16(table)),
             "sparse_apply_fused: E multiple of 4 in [4,256], 16-byte alignment");
  MI_REQUIRE(!table || d_concat || d_logit_fm, "sparse_apply_fused: table update needs d_concat and/or d_logit_fm");
  MI_REQUIRE(!d_concat || (ld_dconcat >= (int64_t)F * E && (ld_dconcat & 3) == 0 && mi::aligned16(d_concat)),
             "sparse_apply_fused: d_concat leading dimension / alignment");
  MI_REQUIRE(!d_logit_fm || (sumv && mi::aligned16(sumv)), "sparse_apply_fused: FM gradient needs sumv");
  MI_REQUIRE(!lin_w || d_logit_lin, "sparse_apply_fused: lin_w needs d_logit_lin");
  const bool need0 = hp->kind != MI_OPT_SGD;
  const bool need1 = hp->kind == MI_OPT_ADAM || hp->kind == MI_OPT_FTRL || hp->kind == MI_OPT_RMSPROP;
  MI_REQUIRE(!table || ((!need0 || t_slot0) && (!need1 || t_slot1)), "sparse_apply_fused: table slots missing");
  MI_REQUIRE(!lin_w || ((!need0 || l_slot0) && (!need1 || l_slot1)), "sparse_apply_fused: linear slots missing");
  const int lpr = table ? lanes_per_row(E) : 1;
  const int64_t blocks = mi::ceil_div(n_max * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "sparse_apply_fused: grid too large");
  const Hp h = make_hp(hp);
  const FusedGrad fg{d_concat, ld_dconcat, sumv, d_logit_fm, d_logit_lin, F};
  MI_DISPATCH_LPR(lpr, (sparse_apply_k<L, true><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           table, t_slot0, t_slot1, lin_w, l_slot0, l_slot1, last_step, uniq_rows, seg_start,
                           sorted_entry, num_uniq, nullptr, nullptr, E, step, h, fg)));
  MI_CHECK_LAUNCH("sparse_apply_fused");
  return MI_OK;
}

int32_t mi_sparse_catchup(float* table, float* t_m, float* t_v, float* lin_w, float* l_m, float* l_v,
                          int32_t* last_step, const int32_t* uniq_rows, const int32_t* num_uniq,
                          int64_t n_max, int32_t E, int32_t step_to, const float* lr_table,
                          float beta1, float beta2, float epsilon, mi_stream_t stream) {
  MI_REQUIRE(n_max >= 0 && step_to >= 0, "sparse_catchup: n_max=%lld step_to=%d", (long long)n_max, step_to);
  if (n_max == 0 || step_to == 0) return MI_OK;
  MI_REQUIRE(last_step && lr_table, "sparse_catchup: null buffer");
  MI_REQUIRE(table || lin_w, "sparse_catchup: nothing to update");
  MI_REQUIRE(!table || (t_m && t_v && E >= 4 && E <= 256 && (E & 3) == 0 && mi::aligned16(table)),
             "sparse_catchup: table needs m, v, E multiple of 4 in [4,256]");
  MI_REQUIRE(!lin_w || (l_m && l_v), "sparse_catchup: lin_w needs m, v");
  MI_REQUIRE(!uniq_rows || num_uniq, "sparse_catchup: uniq_rows without num_uniq");
  const int lpr = table ? lanes_per_row(E) : 1;
  const int64_t blocks = mi::ceil_div(n_max * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "sparse_catchup: grid too large");
  MI_DISPATCH_LPR(lpr, (sparse_catchup_k<L><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           table, t_m, t_v, lin_w, l_m, l_v, last_step, uniq_rows, num_uniq, n_max, E, step_to,
                           lr_table, beta1, beta2, epsilon)));
  MI_CHECK_LAUNCH("sparse_catchup");
  return MI_OK;
}

}  // extern "C"
