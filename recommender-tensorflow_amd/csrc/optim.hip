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
#include <algorithm>

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
                                                        Hp h, const mi_step_state_t* __restrict__ st) {
  if (st) h.lr_t = st->lr_t;
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
// streaming forms for optimizer slots: read once and written once per step, never re-read before the
// next step — keep them from evicting the rows the gather is about to read
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  const f32x4_t x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<f32x4_t*>(p));
}

// One group of LPR lanes per unique row; lane l owns elements 4l..4l+3; lane 0 also owns the
// row's linear weight.  Duplicates are summed in ascending entry order.
// FUSED: the per-entry gradients are not read from d_rows / d_lin but rebuilt in place from what
// the backward left behind (single-GPU path): entry e = (b, f) contributes
//   d_concat[b, f*E:(f+1)*E] + dlf[b] * (sumv[b,:] - w)      (w = this row, still un-updated)
// to the row and dll[b] to its linear weight — mi_embed_fm_linear_bwd folded into the apply, so the
// [B*F, E] gradient matrix is never written or re-read.
struct FusedGrad {
  const float* d_concat; int64_t ldd; const float* sumv; const float* dlf; const float* dll; int F;
  int64_t b0;        // d_concat / sumv / dlf / dll belong to examples b0.. of the batch (a chunk of a pipelined step)
};

// Segments longer than this are left to sparse_apply_long_k (a workgroup per row instead of a lane
// group): with skewed ids one row can own thousands of a batch's entries.
constexpr int kLongSeg = 48;
// consecutive rows a workgroup of sparse_apply_long_k owns together.  1: a workgroup's rows are a grid apart.  (Measured with 8 —
// 32 bytes of seg_start per lane group instead of a cache line per lane: the scan that finds nothing, uniform ids, 17 -> 8 us,
// but Zipf(1.05) ids 2.60 -> 3.25 ms per step: a field's hot rows are its first ids, and eight of them then queue up in one
// workgroup.  The spread is worth more than the scan.)
constexpr int kRun = 1;

struct ApplyArgs {
  float* table; float* t0; float* t1;
  float* lin_w; float* l0; float* l1;
  int32_t* last_step;
  const int32_t* uniq_rows; const int32_t* seg_start; const int32_t* sorted_entry; const int32_t* num_uniq;
  const float* d_rows; const float* d_lin;
  int E, step;
  // STORE form (mi_entry_grads_segsum): the summed gradients of distinct requests u_begin .. u_begin + u_count
  // are written to out_rows / out_lin instead of being applied; "row" u is slot u of the exchange buffer
  float* out_rows; float* out_lin; int u_begin, u_count;
  // element stride of lin_w / l0 / l1 / last_step: 1 = four separate arrays, 4 = one 16-byte record per row
  // {w, slot0, slot1, stamp} (a row's wide-part state then costs one memory sector instead of four)
  int ls;
  const mi_step_state_t* st;   // device-resident step / lr_t of a replayable (captured) step, or nullptr
  // floats between consecutive rows of table / t0 / t1 (>= E): E = three separate [R, E] arrays; 3 E with t0 = table + E,
  // t1 = table + 2 E = ONE [w | slot0 | slot1] record per row — a row's whole state is then one contiguous run of
  // 12 E bytes (one DRAM page visit per row and direction instead of three)
  int64_t ts;
  // floats between consecutive entries of d_rows / of d_lin (E and 1: two arrays; one value for both: the gradients arrive
  // as ONE record [row gradient | weight gradient | pad] per request, d_lin = d_rows + E — the packed exchange of the
  // row-sharded step), and the same for out_rows / out_lin of the STORE form
  int64_t gs, gls, os, ols;
};

// sum of the gradients of entries sorted_entry[k_beg..k_end) of one row, in that order
template <bool FUSED>
__device__ __forceinline__ void seg_accumulate(const ApplyArgs& a, const FusedGrad& fg, int k_beg, int k_end, int l,
                                               bool lane_on, const float4& w, float4& g, float& gl) {
  const int E = a.E;
  for (int k = k_beg; k < k_end; ++k) {
    const int64_t e = a.sorted_entry[k];
    if constexpr (FUSED) {
      const int64_t bg = e / fg.F;
      const int f = static_cast<int>(e - bg * fg.F);
      const int64_t b = bg - fg.b0;
      if (a.table && lane_on) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fg.d_concat) v = ld4(fg.d_concat + b * fg.ldd + static_cast<int64_t>(f) * E + 4 * l);
        if (fg.dlf) {
          const float gf = fg.dlf[b];
          if (fg.sumv) {
            const float4 sv = ld4(fg.sumv + b * E + 4 * l);
            v.x += gf * (sv.x - w.x); v.y += gf * (sv.y - w.y);
            v.z += gf * (sv.z - w.z); v.w += gf * (sv.w - w.w);
          } else {      // d_concat already carries gf * sumv (the data gradient's epilogue added it once per example)
            v.x -= gf * w.x; v.y -= gf * w.y; v.z -= gf * w.z; v.w -= gf * w.w;
          }
        }
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
      }
      if (a.lin_w && l == 0) gl += fg.dll[b];
    } else {
      if (a.table && lane_on) {
        const float4 v = ld4(a.d_rows + e * a.gs + 4 * l);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
      }
      if (a.lin_w && l == 0) gl += a.d_lin[e * a.gls];
    }
  }
}

// optimizer rule on row r with the summed gradient (g, gl); stamps the row.  Adam rows whose stamp
// is older than step - 1 (mi_sparse_catchup ran with defer_slots: it moved w only) first get the
// decay of m and v for the steps they sat out — the same multiply chain the catch-up ran, hence
// the same bits; a fully caught-up row (stamp == step - 1) and a never-applied one (m = v = 0) skip it.
// A row's optimizer state, loaded BEFORE its gradient is summed: the loads depend on the row id only, and issued next
// to the load of w they are in flight while the segment's entries are fetched (sorted_entry -> d_concat, two dependent
// latencies) instead of starting after them — the kernel is bound by bytes in flight per wave (its time scales with
// occupancy), not by issue.
struct RowState { float4 s0, s1; float lw, ls0, ls1; int stamp; };
__device__ __forceinline__ RowState load_row_state(const ApplyArgs& a, const Hp& h, int64_t r, int l, bool lane_on) {
  RowState q;
  q.s0 = q.s1 = make_float4(0.f, 0.f, 0.f, 0.f);
  q.lw = q.ls0 = q.ls1 = 0.f;
  q.stamp = 0;
  if (a.table && lane_on) {
    const int64_t o = r * a.ts + 4 * l;
    if (a.t0) q.s0 = ld4_nt(a.t0 + o);
    if (a.t1) q.s1 = ld4_nt(a.t1 + o);
  }
  if (h.kind == MI_OPT_ADAM && a.last_step) q.stamp = a.last_step[r * a.ls];
  if (l == 0 && a.lin_w) {
    const int64_t o = r * a.ls;
    q.lw = a.lin_w[o];
    if (a.l0) q.ls0 = a.l0[o];
    if (a.l1) q.ls1 = a.l1[o];
  }
  return q;
}

__device__ __forceinline__ void apply_row(const ApplyArgs& a, const Hp& h, int64_t r, int l, bool lane_on, float4 w,
                                          const float4& g, float gl, const RowState& q) {
  int missed = 0;
  if (h.kind == MI_OPT_ADAM && a.last_step) missed = q.stamp > 0 ? max(0, a.step - 1 - q.stamp) : 0;
  if (a.table && lane_on) {
    const int64_t o = r * a.ts + 4 * l;
    float4 s0 = q.s0, s1 = q.s1;
    for (int j = 0; j < missed; ++j) {
      s0.x = s0.x * h.beta1; s0.y = s0.y * h.beta1; s0.z = s0.z * h.beta1; s0.w = s0.w * h.beta1;
      s1.x = s1.x * h.beta2; s1.y = s1.y * h.beta2; s1.z = s1.z * h.beta2; s1.w = s1.w * h.beta2;
    }
    sparse_rule(h, w.x, s0.x, s1.x, g.x);
    sparse_rule(h, w.y, s0.y, s1.y, g.y);
    sparse_rule(h, w.z, s0.z, s1.z, g.z);
    sparse_rule(h, w.w, s0.w, s1.w, g.w);
    st4(a.table + o, w);
    if (a.t0) st4_nt(a.t0 + o, s0);
    if (a.t1) st4_nt(a.t1 + o, s1);
  }
  if (l == 0) {
    if (a.lin_w) {
      const int64_t o = r * a.ls;
      float lw = q.lw, s0 = q.ls0, s1 = q.ls1;
      for (int j = 0; j < missed; ++j) { s0 = s0 * h.beta1; s1 = s1 * h.beta2; }
      sparse_rule(h, lw, s0, s1, gl);
      a.lin_w[o] = lw;
      if (a.l0) a.l0[o] = s0;
      if (a.l1) a.l1[o] = s1;
    }
    if (a.last_step) a.last_step[r * a.ls] = a.step;
  }
}

// one lane group per unique row: duplicates summed in ascending entry order (TF's CPU order)
__device__ __forceinline__ void store_row(const ApplyArgs& a, int64_t u, int l, bool lane_on, const float4& g, float gl) {
  if (a.out_rows && lane_on) st4(a.out_rows + u * a.os + 4 * l, g);
  if (a.out_lin && l == 0) a.out_lin[u * a.ols] = gl;
}

template <int LPR, bool FUSED, bool STORE = false>
__global__ __launch_bounds__(kBlock) void sparse_apply_k(ApplyArgs a, Hp h, const FusedGrad fg) {
  if (a.st) { a.step = a.st->step; h.lr_t = a.st->lr_t; }
  int64_t u = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  if (u >= (STORE ? a.u_count : *a.num_uniq)) return;
  if (STORE) u += a.u_begin;
  const int s_beg = a.seg_start[u], s_end = a.seg_start[u + 1];
  if (s_end - s_beg > kLongSeg) return;            // sparse_apply_long_k's
  const int64_t r = STORE ? u : static_cast<int64_t>(a.uniq_rows[u]);
  const bool lane_on = 4 * l < a.E;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  float gl = 0.f;
  float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.table && lane_on) w = ld4(a.table + r * a.ts + 4 * l);
  RowState q;
  if constexpr (!STORE) q = load_row_state(a, h, r, l, lane_on);
  seg_accumulate<FUSED>(a, fg, s_beg, s_end, l, lane_on, w, g, gl);
  if constexpr (STORE) store_row(a, u, l, lane_on, g, gl);
  else apply_row(a, h, r, l, lane_on, w, g, gl, q);
}

// Rows with more than kLongSeg entries: a workgroup per row.  Workgroup j looks at the rows of the runs
// j, j + grid, j + 2 grid, ... of kRun consecutive rows (hot rows have neighbouring ids: the stride spreads them), collects
// the long ones, and for each splits the segment into kBlock/LPR contiguous slices, one per lane
// group, summed in order; the slice sums are then added in slice order.  A fixed order, so results
// are reproducible; it differs from the one-pass order only in fp32 association.
template <int LPR, bool FUSED, bool STORE = false>
__global__ __launch_bounds__(kBlock) void sparse_apply_long_k(ApplyArgs a, Hp h, const FusedGrad fg) {
  if (a.st) { a.step = a.st->step; h.lr_t = a.st->lr_t; }
  constexpr int G = kBlock / LPR;
  __shared__ int list[kBlock];
  __shared__ int n_list;
  __shared__ float4 part[kBlock];                  // [G][LPR]
  __shared__ float part_l[G];
  const int U = STORE ? a.u_count : *a.num_uniq;
  const int ub = STORE ? a.u_begin : 0;
  const int t = threadIdx.x, l = t & (LPR - 1), grp = t / LPR;
  const bool lane_on = 4 * l < a.E;
  const int64_t per_round = static_cast<int64_t>(kBlock) * gridDim.x;
  // The common case first (uniform ids: no segment of the batch is long): one look at every segment length this workgroup owns,
  // ONE barrier, out — the round-by-round collection below costs two barriers and a shared atomic per round (16 us per
  // step at config 3 for finding nothing).
  {
    int any = 0;
    for (int64_t base = 0; base < U; base += per_round) {
      const int64_t u = base + (static_cast<int64_t>(t / kRun) * gridDim.x + blockIdx.x) * kRun + t % kRun;
      if (u < U) any |= (a.seg_start[ub + u + 1] - a.seg_start[ub + u] > kLongSeg) ? 1 : 0;
    }
    if (!__syncthreads_or(any)) return;
  }
  for (int64_t base = 0; base < U; base += per_round) {
    if (t == 0) n_list = 0;
    __syncthreads();
    const int64_t u = base + (static_cast<int64_t>(t / kRun) * gridDim.x + blockIdx.x) * kRun + t % kRun;
    if (u < U && a.seg_start[ub + u + 1] - a.seg_start[ub + u] > kLongSeg) list[atomicAdd(&n_list, 1)] = static_cast<int>(ub + u);
    __syncthreads();
    const int n = n_list;
    for (int j = 0; j < n; ++j) {
      const int uu = list[j];
      const int64_t r = STORE ? static_cast<int64_t>(uu) : static_cast<int64_t>(a.uniq_rows[uu]);
      const int s_beg = a.seg_start[uu], s_end = a.seg_start[uu + 1];
      const int per = (s_end - s_beg + G - 1) / G;
      const int k0 = min(s_end, s_beg + grp * per), k1 = min(s_end, k0 + per);
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      float gl = 0.f;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.table && lane_on) w = ld4(a.table + r * a.ts + 4 * l);
      seg_accumulate<FUSED>(a, fg, k0, k1, l, lane_on, w, g, gl);
      part[t] = g;
      if (l == 0) part_l[grp] = gl;
      __syncthreads();
      if (grp == 0) {
        g = part[l];
        gl = part_l[0];
#pragma unroll 4
        for (int q = 1; q < G; ++q) {
          const float4 v = part[q * LPR + l];
          g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
          gl += part_l[q];
        }
        if constexpr (STORE) store_row(a, uu, l, lane_on, g, gl);
        else apply_row(a, h, r, l, lane_on, w, g, gl, load_row_state(a, h, r, l, lane_on));
      }
      __syncthreads();
    }
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

// With the rows sorted by staleness the replay is bound by VALU issue.  hipcc's correctly rounded sqrtf
// and '/' carry range scaling (v_div_scale x2, v_div_fmas, v_div_fixup; the 2^32 pre-scale, un-scale
// and class test of sqrt) that only matters at the ends of the exponent range.  These are the same
// algorithms without it — for sqrt LLVM's rsq-based expansion (below), for '/' v_rcp_f32 + Newton + two
// residual corrections — hence the same bits wherever no intermediate leaves the normal range;
// catchup_in_range() is the (generous) condition under which a WAVE takes them (all its lanes in
// range: a wave never runs both loops); otherwise it runs sqrtf and '/'.
// sqrt, first form: v_sqrt_f32 + the one-ulp residual test (what hipcc emits for sqrtf, minus the range scaling): 2 integer
// adds, 2 FMAs, 2 compares and 2 selects per element after the transcendental instruction — only the FMAs pack.
__device__ __forceinline__ float sqrt_rn_fixup(float x) {              // x == 0 or x >= 2^-96
  float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
  s = (rm <= 0.f) ? sm : s;
  s = (rp > 0.f) ? sp : s;
  return s;
}
// sqrt, the form the replay loops use (round 2): v_rsq_f32, one coupled Newton step on (s ~ sqrt x, h ~ 1/(2 sqrt x)) and a
// final residual correction — LLVM's other correctly rounded expansion (the one it picks when f32 denormals are flushed),
// again without the range scaling.  Two multiplies and five FMAs, ALL of which hipcc packs two elements to an
// instruction: per 4 elements 14 packed instructions instead of 4 packed + 24 single ones (38 packed + 8 transcendental (half rate on gfx950: tools/probe/valu_cost_probe.hip)
// + 3 single per replayed step instead of 28 + 8 + 27).  That it returns the correctly rounded root — the bits of
// sqrtf — is not taken on trust: mi_selftest_sqrt compares the two on the device for EVERY fp32 value in
// [2^-100, 2^24] (tests/test_hip_kernels.py::test_fast_sqrt_equals_sqrtf_on_every_value_in_range).  x == 0 gives NaN
// (0 * inf): the in-range condition below keeps v == 0 out.
__device__ __forceinline__ float sqrt_rn_inrange(float x) {            // 2^-96 <= x <= 2^24
  const float r = __builtin_amdgcn_rsqf(x);
  float s = x * r, h = 0.5f * r;
  const float e = fmaf(-h, s, 0.5f);
  h = fmaf(h, e, h);
  s = fmaf(s, e, s);
  const float d = fmaf(-s, s, x);
  return fmaf(d, h, s);
}
__device__ __forceinline__ float div_rn_inrange(float n, float d) {     // n == 0 or |n| >= 2^-100; d, n/d normal
  float r = __builtin_amdgcn_rcpf(d);
  const float e = fmaf(-d, r, 1.0f);
  r = fmaf(e, r, r);
  float q = n * r;
  const float e2 = fmaf(-d, q, n);
  q = fmaf(e2, r, q);
  const float e3 = fmaf(-d, q, n);
  return fmaf(e3, r, q);
}
// m and v only shrink during the replay (by b1^k, b2^k <= 1): in range at the start and after k steps
// means in range throughout.  |m| >= 2^-50 and k <= 200 keep lr_t*m above 2^-101 (lr_t >= 2^-20) and
// its residuals (2^-24 below) normal; v in [2^-93, 2^20] keeps sqrt(v)+eps within [eps, 2^10] and
// v*b2^k above 2^-96.
__device__ __forceinline__ bool catchup_in_range(float m, float v) {
  const float am = fabsf(m);
  return (am == 0.f || (am >= 0x1p-50f && am <= 0x1p60f)) && v >= 0x1p-93f && v <= 0x1p20f;   // (v == 0: generic loop)
}
// (the bounds above assume the decays of 200 steps stay above 2^-31 and 2^-3: beta1 >= 0.9, beta2 >= 0.99)
__device__ __forceinline__ bool catchup_params_in_range(int steps, float lr_last, float eps, float b1, float b2) {
  return steps <= 200 && lr_last >= 0x1p-17f && eps >= 0x1p-40f && eps <= 1.f &&   // lr_t >= 0.14 lr for every t
         b1 >= 0.9f && b1 <= 1.f && b2 >= 0.99f && b2 <= 1.f;
}

// ---- the BOUNDED-ERROR replay (mi_sparse_catchup flag MI_CATCHUP_BOUNDED) --------------------------------------------
// The exact replay above spends ~19 dependent operations per element and step on a correctly rounded sqrt and divide
// whose only purpose is to land on TF's bits.  The contract (north star) is 1e-5 on the logits, not bit equality, so
// this form keeps what is cheap to keep exact and approximates the rest with a stated bound:
//   m_j = m_{j-1} * b1                 exactly the reference's chain (same bits)
//   u_j = lr_t[s] * m_j                 exactly the reference's numerator (same bits)
//   sqrt(v_j)  ~  s0 * rho_j            s0 = v_sqrt_f32(v_0) (1 ulp, once per element), rho_j ~ beta2^(j/2): a per-ROW
//                                       scalar chain rho_j = rho_{j-1} * (rho_hi + rho_lo), rho_hi + rho_lo = sqrt(beta2) to
//                                       2^-48 (no systematic drift; rounding noise <= sqrt(j) * 2^-24 rms)
//   w_j = fma(-u_j, r_j, w_{j-1}), r_j ~ 1 / fma(s0, rho_j, eps)    a reciprocal to <= 1 ulp (the wide part: v_rcp_f32; the rows: carried
//                                       from step to step and corrected against each d_j, see the row kernel); ONE rounding of w per step, like the reference
// Per element and step: 4 packed-able VALU operations + 1 transcendental instead of 16 + 2; no range conditions at all
// (v = 0, denormal m, any gap: the same loop), so no wave ever falls back to a slow generic loop.
// Error against the reference's literal fp32 sweep, per replayed step j of a row: the update t_j = u_j / (sqrt(v_j) + eps)
// is reproduced to |t~_j / t_j - 1| <= (2 [the reciprocal: <= 1 ulp] + 2 [s0] + 1 [fma] + 2j [rho chain, worst case; ~sqrt(j)/2
// rms] + 3 [the reference's own sqrt, add and divide roundings] + j/2 [its v chain]) * 2^-24; the updates decay like
// 0.9^j, so the sum over a replay is off by <~ 1e-6 of its FIRST update in the worst case and ~1e-7 of it typically —
// of an update that is itself ~1e-3 |w|.  On top of that comes the rare step in which the difference moves RN(w - t)
// across a rounding boundary (1 ulp of w each, about one step in 300):
// tests/test_hip_kernels.py::test_bounded_catchup_stays_within_its_bound_of_the_sweep holds every variable to
// 3 ulp(w) + 2e-6 * sum_j |t_j| (and >= 98 % of them to 1e-7 relative) after 150-200 replayed steps with (m, v) from the
// smallest to the largest magnitudes Adam can produce.
// m and v themselves (written back only without defer_slots) are the exact chains in both modes.
struct RhoSplit { float hi, lo; };
__device__ __forceinline__ RhoSplit rho_split(float b2) {
  const double r = sqrt(static_cast<double>(b2));
  RhoSplit q;
  q.hi = static_cast<float>(r);
  q.lo = static_cast<float>(r - static_cast<double>(q.hi));
  return q;
}
__device__ __forceinline__ float bounded_step(float w, float u, float s0, float rj, float eps) {
  return fmaf(-u, __builtin_amdgcn_rcpf(fmaf(s0, rj, eps)), w);
}

// The wide part: one thread per row (1/E of the work).  Its own kernel, run BEFORE the row kernel (it
// reads the stamps the row kernel writes), so that lane 0 of a row's lane group does not drag a fifth
// chain through a second loop of the same length.
__global__ __launch_bounds__(kBlock) void catchup_lin_k(
    float* __restrict__ lin_w, float* __restrict__ lm, float* __restrict__ lv, const int32_t* __restrict__ last_step,
    const int32_t* __restrict__ uniq_rows, const int32_t* __restrict__ num_uniq, int64_t n_max, int step_to,
    const float* __restrict__ lr_table, float b1, float b2, float eps, bool defer_slots, int st,
    const mi_step_state_t* __restrict__ ss, bool bounded) {
  if (ss) step_to = ss->step - 1;
  const int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t count = uniq_rows ? static_cast<int64_t>(*num_uniq) : n_max;
  const bool on = u < count;
  const int64_t r = (on ? (uniq_rows ? static_cast<int64_t>(uniq_rows[u]) : u) : 0) * st;   // (element offset of the row's state)
  const int ls = on ? last_step[r] : step_to;
  const bool work = on && ls > 0 && ls < step_to;
  float w = 0.f, m = 0.f, v = 0.f;
  if (work) { w = lin_w[r]; m = lm[r]; v = lv[r]; }
  if (bounded) {
    if (work) {
      const RhoSplit rho = rho_split(b2);
      const float s0 = sqrtf(v);
      float rj = 1.f;
      for (int s = ls + 1; s <= step_to; ++s) {
        rj = fmaf(rj, rho.lo, rj * rho.hi);
        m = m * b1;
        w = bounded_step(w, lr_table[s] * m, s0, rj, eps);
      }
      if (!defer_slots)
        for (int s = ls + 1; s <= step_to; ++s) v = v * b2;
      lin_w[r] = w;
      if (!defer_slots) { lm[r] = m; lv[r] = v; }
    }
    return;
  }
  const bool ok = !work || (catchup_params_in_range(step_to - ls, lr_table[step_to], eps, b1, b2) && catchup_in_range(m, v));
  if (__ballot(!ok) == 0) {
    if (work)
      for (int s = ls + 1; s <= step_to; ++s) {
        m = m * b1; v = v * b2;
        w = w - div_rn_inrange(lr_table[s] * m, sqrt_rn_inrange(v) + eps);
      }
  } else if (work) {
    replay(w, m, v, ls + 1, step_to, lr_table, b1, b2, eps);
  }
  if (work) {
    lin_w[r] = w;
    if (!defer_slots) { lm[r] = m; lv[r] = v; }
  }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void sparse_catchup_k(
    float* __restrict__ table, float* __restrict__ tm, float* __restrict__ tv,
    int32_t* __restrict__ last_step, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, int64_t n_max, int E, int step_to,
    const float* __restrict__ lr_table, float b1, float b2, float eps, bool defer_slots, int st,
    const mi_step_state_t* __restrict__ ss, bool keep_stamps, int64_t ts) {
  if (ss) step_to = ss->step - 1;
  const int64_t u = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  const int64_t count = uniq_rows ? static_cast<int64_t>(*num_uniq) : n_max;
  if (u >= count) return;
  const int64_t r = uniq_rows ? static_cast<int64_t>(uniq_rows[u]) : u;
  const int ls = last_step[r * st];
  if (ls >= step_to) return;
  // a row that was never applied has m = v = 0: every replayed step subtracts exactly 0
  if (ls > 0) {
    if (table && 4 * l < E) {
      const int64_t o = r * ts + 4 * l;
      float4 w = ld4(table + o), m = ld4_nt(tm + o), v = ld4_nt(tv + o);
      // one loop for the four elements: one lr_t load and one loop counter per step instead of four,
      // four independent sqrt/divide chains in flight (the arithmetic per element is unchanged)
      const bool ok = catchup_params_in_range(step_to - ls, lr_table[step_to], eps, b1, b2) && catchup_in_range(m.x, v.x) &&
                      catchup_in_range(m.y, v.y) && catchup_in_range(m.z, v.z) && catchup_in_range(m.w, v.w);
      if (__ballot(!ok) == 0) {                    // over the wave's active lanes
        for (int s = ls + 1; s <= step_to; ++s) {
          const float lr = lr_table[s];
          m.x = m.x * b1; m.y = m.y * b1; m.z = m.z * b1; m.w = m.w * b1;
          v.x = v.x * b2; v.y = v.y * b2; v.z = v.z * b2; v.w = v.w * b2;
          w.x = w.x - div_rn_inrange(lr * m.x, sqrt_rn_inrange(v.x) + eps);
          w.y = w.y - div_rn_inrange(lr * m.y, sqrt_rn_inrange(v.y) + eps);
          w.z = w.z - div_rn_inrange(lr * m.z, sqrt_rn_inrange(v.z) + eps);
          w.w = w.w - div_rn_inrange(lr * m.w, sqrt_rn_inrange(v.w) + eps);
        }
      } else
      for (int s = ls + 1; s <= step_to; ++s) {
        const float lr = lr_table[s];
        m.x = m.x * b1; m.y = m.y * b1; m.z = m.z * b1; m.w = m.w * b1;
        v.x = v.x * b2; v.y = v.y * b2; v.z = v.z * b2; v.w = v.w * b2;
        w.x = w.x - (lr * m.x) / (sqrtf(v.x) + eps);
        w.y = w.y - (lr * m.y) / (sqrtf(v.y) + eps);
        w.z = w.z - (lr * m.z) / (sqrtf(v.z) + eps);
        w.w = w.w - (lr * m.w) / (sqrtf(v.w) + eps);
      }
      st4(table + o, w);
      if (!defer_slots) { st4_nt(tm + o, m); st4_nt(tv + o, v); }
    }
  }
  // (the wide part's scalar per row: catchup_lin_k)
  // defer_slots: the sparse apply that follows in the same step decays m and v itself (it reads and
  // writes them anyway) from the old stamp, so neither they nor the stamp are written here — a third
  // of this kernel's HBM traffic.
  // every lane of the group has read last_step[r] above (same wave, program order) before lane 0 writes
  if (l == 0 && !defer_slots && !keep_stamps) last_step[r * st] = step_to;
}

// The bounded-error replay of the rows (MI_CATCHUP_BOUNDED).  With 4 + 1 instructions per element and step the replay is
// no longer bound by instruction issue (0.25 ms of it at config 3) but by how a wave's life is spent: row id -> stamp ->
// w, m, v are three DEPENDENT memory latencies before a dozen replayed steps, and a kernel shaped like the exact one
// (one row per lane group, then exit) moved its 1.7 GB at 2.9 TB/s.  So this kernel is a software pipeline:
//   * a lane group walks rows u, u + G, u + 2G, ... (G lane groups in the grid, a few workgroups per CU) and loads row
//     u + G's stamp, w, m and v — all four at once, they depend on the row id only — BEFORE it replays row u: the next
//     row's latencies pass under this row's arithmetic;
//   * nothing inside the replay loop touches vector memory: lr_t[s] comes from an LDS copy of the table's last
//     kLrWindow entries (s_waitcnt vmcnt counts in order: one global load in the loop would wait for the whole prefetch);
//     older steps — a row that sat out more than kLrWindow steps — are replayed from the global table first;
//   * the rows arrive sorted by staleness (mi_catchup_rows_by_gap), so the lane groups of a wave run loops of about the
//     same length in every round.
constexpr int kLrWindow = 1024;

struct RowIn { int64_t r; int ls; float4 w, m, v; };

// a row's stamp, w, m and v: four loads that depend on the row id only, issued together
__device__ __forceinline__ RowIn load_row_in(const float* __restrict__ table, const float* __restrict__ tm, const float* __restrict__ tv,
                                             const int32_t* __restrict__ last_step, int64_t r, int64_t ts, int l, bool lane_on, int st) {
  RowIn q;
  q.r = r;
  q.ls = last_step[r * st];
  q.w = q.m = q.v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane_on) {
    const int64_t o = r * ts + 4 * l;
    q.w = ld4(table + o); q.m = ld4_nt(tm + o); q.v = ld4_nt(tv + o);
  }
  return q;
}

template <int LPR, bool DEEP>
__global__ __launch_bounds__(kBlock) void sparse_catchup_bounded_k(
    float* __restrict__ table, float* __restrict__ tm, float* __restrict__ tv,
    int32_t* __restrict__ last_step, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, int64_t n_max, int E, int step_to,
    const float* __restrict__ lr_table, float b1, float b2, float eps, bool defer_slots, int st,
    const mi_step_state_t* __restrict__ ss, bool keep_stamps, int64_t ts) {
  __shared__ float lr_s[kLrWindow];
  if (ss) step_to = ss->step - 1;
  const int base = step_to - (kLrWindow - 1);                 // lr_s[i] = lr_t[base + i]
  for (int i = threadIdx.x; i < kLrWindow; i += kBlock) lr_s[i] = base + i >= 1 ? lr_table[base + i] : 0.f;
  __syncthreads();
  const int l = threadIdx.x & (LPR - 1);
  const bool lane_on = 4 * l < E;
  const int64_t count = uniq_rows ? static_cast<int64_t>(*num_uniq) : n_max;
  // Which row when.  The rows come sorted by staleness, and a wave's lane groups should see equal staleness — but the
  // CHIP should not: a window of the sorted list is all short replays (memory bound) or all long ones (issue bound),
  // and waves that walk the list side by side make the whole chip one or the other.  So the list is cut into chunks of
  // one wave's rows (64 / LPR consecutive rows), the chunks into W segments of J chunks (W = waves in the grid, a power
  // of two), and wave k takes in round j chunk j of segment (k + j S) mod W: for fixed j a bijection over the waves, so
  // every chunk is taken once; over the rounds a wave samples every staleness (equal work per wave), and in any round
  // the waves between them hold all of them (memory and arithmetic overlap across waves).
  constexpr int RPW = 64 / LPR;
  const int64_t n_chunks = (count + RPW - 1) / RPW;
  const int W = static_cast<int>(gridDim.x) * (kBlock / 64);                  // (a power of two: the launcher's grid)
  const int64_t J = (n_chunks + W - 1) / W;
  const int S = static_cast<int>(J > 0 && W / J > 1 ? W / J : 1);
  const int k = static_cast<int>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  const int grp = (threadIdx.x & 63) / LPR;
  auto row_at = [&](int64_t j) -> int64_t {                                   // position in the sorted list, or >= count
    const int64_t seg = (k + j * S) & (W - 1);
    const int64_t c = seg * J + j;
    return c < n_chunks ? c * RPW + grp : count;
  };
  if (J == 0) return;
  const RhoSplit rho = rho_split(b2);
  auto row_id = [&](int64_t x) { return uniq_rows ? uniq_rows[x] : static_cast<int32_t>(x); };   // (rows fit int32: the engine checks)
  // The pipeline: while round j's row is replayed, round j + 1's state and round j + 2's ID are in flight.  A round starts
  // by taking over what the previous round prefetched (the one point where the wave waits for memory: everything
  // outstanding there was issued before the previous round's arithmetic), then issues the next prefetch, then computes
  // on registers.  A lane group whose chunk lies past the end of the list carries ls = step_to: nothing to do.
  RowIn none;
  none.r = 0; none.ls = INT32_MAX; none.w = none.m = none.v = make_float4(0.f, 0.f, 0.f, 0.f);
  // the replay of one row from registers (the state a round took over)
  auto replay = [&](const RowIn& cur) {
    // ---- replay row u (a row that was never applied has m = v = 0: every step subtracts exactly 0)
    const int ls = cur.ls;
    if (ls > 0 && ls < step_to) {
      if (lane_on) {
        float4 w = cur.w, m = cur.m;
        // (v_sqrt_f32: 1 ulp; a v too small for it is also far too small to matter next to eps.  An overflowed v = inf keeps a
        // FINITE root: the sweep's update is u / inf = 0, and so is u * r with r ~ 1 / FLT_MAX — but the carried reciprocal's
        // residual 1 - inf * 0 would be NaN)
        const float4 s0 = make_float4(fminf(__builtin_amdgcn_sqrtf(cur.v.x), 3.4028234e38f), fminf(__builtin_amdgcn_sqrtf(cur.v.y), 3.4028234e38f),
                                      fminf(__builtin_amdgcn_sqrtf(cur.v.z), 3.4028234e38f), fminf(__builtin_amdgcn_sqrtf(cur.v.w), 3.4028234e38f));
        float rj = 1.f;
        int s = ls + 1;
        for (; s < base && s <= step_to; ++s) {                 // steps older than the LDS window (rare)
          const float lr = lr_table[s];
          rj = fmaf(rj, rho.lo, rj * rho.hi);
          m.x = m.x * b1; m.y = m.y * b1; m.z = m.z * b1; m.w = m.w * b1;
          w.x = bounded_step(w.x, lr * m.x, s0.x, rj, eps); w.y = bounded_step(w.y, lr * m.y, s0.y, rj, eps);
          w.z = bounded_step(w.z, lr * m.z, s0.z, rj, eps); w.w = bounded_step(w.w, lr * m.w, s0.w, rj, eps);
        }
        // (two-element vectors: hipcc then packs the fma of the denominators too; two steps per trip of the loop)
        // The reciprocal is CARRIED from step to step instead of taken anew (round 5): v_rcp_f32 is a transcendental
        // instruction (half rate on gfx950, one element each: tools/probe/valu_cost_probe.hip) — four of them per lane and
        // step next to ~13 packed / scalar ones.  Consecutive denominators differ by at most 1 - sqrt(beta2) (5e-4) relatively, d_j / d_{j-1} in [sqrt(beta2), 1], so with
        // e = 1 - d_j r_{j-1} (one fma: the exact residual, rounded once) the second-order step r_j = r_{j-1} (1 + e + e^2)
        // lands within e^3 <= 1.3e-10 of 1 / d_j plus ONE rounding (0.5 ulp — tighter than v_rcp_f32's 1 ulp), and since
        // every step corrects against its own d_j nothing accumulates.  Three packed full-rate fmas per two elements in place
        // of two v_rcp_f32: 35 instead of 29 + 8 transcendental instructions per two steps.  Measured (same box, alternating,
        // profiles/r05_catchup_reciprocal.md): rows sorted by staleness 0.441 against 0.444 ms alone and the step 2.685
        // against 2.69 ms — the kernel waits for rows, not for the VALU — rows NOT sorted 0.63 against 0.66 ms.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sa = {s0.x, s0.y}, sb = {s0.z, s0.w};
        const f32x2 ev = {eps, eps}, one = {1.f, 1.f};
        f32x2 ra, rb;                                           // 1 / (s0 rho_j + eps) of the step before the loop's first
        {
          const f32x2 rv = {rj, rj};
          const f32x2 da = __builtin_elementwise_fma(sa, rv, ev), db = __builtin_elementwise_fma(sb, rv, ev);
          ra.x = __builtin_amdgcn_rcpf(da.x); ra.y = __builtin_amdgcn_rcpf(da.y);
          rb.x = __builtin_amdgcn_rcpf(db.x); rb.y = __builtin_amdgcn_rcpf(db.y);
        }
        f32x2 ma = {m.x, m.y}, mb = {m.z, m.w}, wa = {w.x, w.y}, wb = {w.z, w.w};
#pragma unroll 2
        for (; s <= step_to; ++s) {
          const float lr = lr_s[s - base];
          rj = fmaf(rj, rho.lo, rj * rho.hi);
          ma = ma * b1; mb = mb * b1;
          const f32x2 rv = {rj, rj};
          const f32x2 da = __builtin_elementwise_fma(sa, rv, ev), db = __builtin_elementwise_fma(sb, rv, ev);
          const f32x2 ea = __builtin_elementwise_fma(-da, ra, one), eb = __builtin_elementwise_fma(-db, rb, one);
          const f32x2 pa = __builtin_elementwise_fma(ea, ea, ea), pb = __builtin_elementwise_fma(eb, eb, eb);
          ra = __builtin_elementwise_fma(ra, pa, ra); rb = __builtin_elementwise_fma(rb, pb, rb);
          const f32x2 ua = lr * ma, ub = lr * mb;
          wa = __builtin_elementwise_fma(-ua, ra, wa); wb = __builtin_elementwise_fma(-ub, rb, wb);
        }
        m = make_float4(ma.x, ma.y, mb.x, mb.y); w = make_float4(wa.x, wa.y, wb.x, wb.y);
        const int64_t o = cur.r * ts + 4 * l;
        st4(table + o, w);
        if (!defer_slots) {
          float4 v = cur.v;
          for (int q = ls + 1; q <= step_to; ++q) { v.x = v.x * b2; v.y = v.y * b2; v.z = v.z * b2; v.w = v.w * b2; }
          st4_nt(tm + o, m); st4_nt(tv + o, v);
        }
      }
    }
    // (every lane of the group read the stamp above before lane 0 overwrites it: same wave, program order)
    if (l == 0 && !defer_slots && !keep_stamps && ls < step_to) last_step[cur.r * st] = step_to;
  };
  if constexpr (!DEEP) {
  int64_t u0 = row_at(0);
  RowIn nxt = u0 < count ? load_row_in(table, tm, tv, last_step, row_id(u0), ts, l, lane_on, st) : none;
  int64_t u1 = J > 1 ? row_at(1) : count;
  int32_t id_pref = u1 < count ? row_id(u1) : 0;
  for (int64_t j = 0; j < J; ++j) {
    const RowIn cur = nxt;
    const int64_t r_next = id_pref;
    const bool more = j + 1 < J;
    if (more) {
      nxt = u1 < count ? load_row_in(table, tm, tv, last_step, r_next, ts, l, lane_on, st) : none;
      u1 = j + 2 < J ? row_at(j + 2) : count;
      if (u1 < count) id_pref = row_id(u1);
    }
    replay(cur);
  }
  } else {
  // Two rounds' state in flight: twice the bytes per wave on the wire for the same number of resident waves.
  //  * Every load is UNCONDITIONAL — a lane group past the end of the list, or a lane past the row's end, loads a valid
  //    address it does not use (position count - 1 / the row's first floats; without a row list the stamps stand in for
  //    it): with loads under branches hipcc's wait-count bookkeeping gives up and drains the pipeline (s_waitcnt vmcnt(0))
  //    every round.
  //  * The three buffers ROTATE through an unrolled loop — a register copy of a state still in flight would wait for it.
  //  * The ID of round j + 3's row is loaded BEFORE round j + 2's state, so that waiting for it a round later does not
  //    wait for that state: vmcnt counts in order.
  const int64_t last_pos = count - 1;                        // (count >= 1 here: J > 0)
  const int col = lane_on ? 4 * l : 0;
  const int32_t* id_list = uniq_rows ? uniq_rows : last_step;
  const bool has_list = uniq_rows != nullptr;
  auto id_at = [&](int64_t jj) -> int32_t {                  // row of round jj (clamped into the list)
    const int64_t u = row_at(jj < J ? jj : J - 1);
    const int64_t pos = u < count ? u : last_pos;
    const int32_t v = id_list[pos];
    return has_list ? v : static_cast<int32_t>(pos);
  };
  auto on_at = [&](int64_t jj) -> bool { return jj < J && row_at(jj) < count; };
  auto state_of = [&](int32_t id, bool on) -> RowIn {
    RowIn q;
    q.r = id;
    const int stamp = last_step[static_cast<int64_t>(id) * st];
    const int64_t o = static_cast<int64_t>(id) * ts + col;
    q.w = ld4(table + o); q.m = ld4_nt(tm + o); q.v = ld4_nt(tv + o);
    q.ls = on ? stamp : INT32_MAX;
    return q;
  };
  RowIn sa = state_of(id_at(0), on_at(0)), sb = state_of(id_at(1), on_at(1)), sc;
  int32_t id_next = id_at(2);
  auto round = [&](const RowIn& x, RowIn& z, int64_t j) {    // consume round j's state x, fill z with round j + 2's
    const int32_t id_a = id_next;
    id_next = id_at(j + 3);
    z = state_of(id_a, on_at(j + 2));
    replay(x);
  };
  for (int64_t j = 0; j < J; j += 3) {                       // (rounds past J - 1 are off: they load and do nothing)
    round(sa, sc, j);
    round(sb, sa, j + 1);
    round(sc, sb, j + 2);
  }
  }
}

// The proof obligation of sqrt_rn_inrange: the same bits as hipcc's correctly rounded sqrtf for every value it is given.
// One thread per fp32 bit pattern; wave-reduced counts.
__global__ __launch_bounds__(kBlock) void selftest_sqrt_k(uint32_t first_bits, int64_t count, unsigned long long* __restrict__ mism) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  bool bad_fast = false, bad_fixup = false;
  if (i < count) {
    const float x = __uint_as_float(first_bits + static_cast<uint32_t>(i));
    const uint32_t ref = __float_as_uint(sqrtf(x));
    bad_fast = __float_as_uint(sqrt_rn_inrange(x)) != ref;
    bad_fixup = __float_as_uint(sqrt_rn_fixup(x)) != ref;
  }
  const unsigned long long bf = __ballot(bad_fast), bx = __ballot(bad_fixup);
  if ((threadIdx.x & 63) == 0) {
    if (bf) atomicAdd(mism, static_cast<unsigned long long>(__popcll(bf)));
    if (bx) atomicAdd(mism + 1, static_cast<unsigned long long>(__popcll(bx)));
  }
}

// key[u] = number of steps row uniq_rows[u] has to be replayed over (clamped to 62), 63 for the slots past
// num_uniq: sorting the rows by it (one 6-bit radix pass) puts rows of equal staleness into the same
// wave of sparse_catchup_k, whose lane groups otherwise all run as long as the stalest of their rows
// (4 rows per wave at E = 64: twice the average gap with geometric gaps).
__global__ __launch_bounds__(kBlock) void gap_keys_k(const int32_t* __restrict__ uniq_rows,
                                                     const int32_t* __restrict__ num_uniq,
                                                     const int32_t* __restrict__ last_step, int64_t n_max, int step_to,
                                                     int32_t* __restrict__ keys, int st,
                                                     const mi_step_state_t* __restrict__ ss) {
  if (ss) step_to = ss->step - 1;
  const int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (u >= n_max) return;
  int32_t key = 63;
  if (u < *num_uniq) {
    const int ls = last_step[static_cast<int64_t>(uniq_rows[u]) * st];
    key = (ls > 0 && ls < step_to) ? min(step_to - ls, 62) : 0;
  }
  keys[u] = key;
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

// workgroups of sparse_apply_long_k: enough to spread the hot rows, few enough that a batch without
// long segments costs one scan of seg_start
unsigned long_grid(int64_t n_max) {
  return static_cast<unsigned>(std::min<int64_t>(1024, mi::ceil_div(n_max, kBlock)));
}

__global__ void step_advance_k(mi_step_state_t* __restrict__ st, const float* __restrict__ lr_table) {
  const int s = st->step + 1;
  st->step = s;
  st->lr_t = lr_table ? lr_table[s] : 0.f;
  st->seed_term = static_cast<uint64_t>(s) * 1000003ull;
}

extern "C" {

int32_t mi_step_advance(mi_step_state_t* device_state, const float* lr_table, mi_stream_t stream) {
  MI_REQUIRE(device_state, "step_advance: null state");
  step_advance_k<<<dim3(1), dim3(1), 0, mi::as_stream(stream)>>>(device_state, lr_table);
  MI_CHECK_LAUNCH("step_advance");
  return MI_OK;
}

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
  dense_apply_k<<<dim3((unsigned)nb), dim3(kBlock), 0, mi::as_stream(stream)>>>(param, slot0, slot1, grad, n, make_hp(hp), mi::step_state());
  MI_CHECK_LAUNCH("dense_apply");
  return MI_OK;
}

int32_t mi_sparse_apply(float* table, float* t_slot0, float* t_slot1, float* lin_w, float* l_slot0,
                        float* l_slot1, int32_t* last_step, const int32_t* uniq_rows,
                        const int32_t* seg_start, const int32_t* sorted_entry,
                        const int32_t* num_uniq, int64_t n_max, const float* d_rows,
                        const float* d_lin, int32_t E, int32_t step, const mi_opt_hparams* hp,
                        int32_t lin_stride, int64_t table_stride, int64_t grad_stride, mi_stream_t stream) {
  if (int32_t rc = check_hp("sparse_apply", hp)) return rc;
  MI_REQUIRE(grad_stride == 0 || (grad_stride >= (table ? E : 1) && (!table || (grad_stride & 3) == 0)),
             "sparse_apply: grad_stride=%lld (0 = d_rows E apart and d_lin 1 apart, else one record per entry)", (long long)grad_stride);
  if (!table) table_stride = 0;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "sparse_apply", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;

  MI_REQUIRE(lin_stride >= 1, "sparse_apply: lin_stride=%d", lin_stride);
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
  ApplyArgs a{table, t_slot0, t_slot1, lin_w, l_slot0, l_slot1, last_step, uniq_rows, seg_start, sorted_entry,
              num_uniq, d_rows, d_lin, E, step};
  a.ls = lin_stride; a.ts = ts;
  a.gs = grad_stride ? grad_stride : E; a.gls = grad_stride ? grad_stride : 1;
  a.st = mi::step_state();
  MI_DISPATCH_LPR(lpr, (sparse_apply_k<L, false><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           a, h, FusedGrad{})));
  MI_CHECK_LAUNCH("sparse_apply");
  if (n_max > kLongSeg) {
    MI_DISPATCH_LPR(lpr, (sparse_apply_long_k<L, false><<<dim3(long_grid(n_max)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                             a, h, FusedGrad{})));
    MI_CHECK_LAUNCH("sparse_apply(long segments)");
  }
  return MI_OK;
}

int32_t mi_sparse_apply_fused(float* table, float* t_slot0, float* t_slot1, float* lin_w, float* l_slot0,
                              float* l_slot1, int32_t* last_step, const int32_t* uniq_rows,
                              const int32_t* seg_start, const int32_t* sorted_entry, const int32_t* num_uniq,
                              int64_t n_max, const float* d_concat, int64_t ld_dconcat, const float* sumv,
                              const float* d_logit_fm, const float* d_logit_lin, int32_t F, int32_t E,
                              int32_t step, const mi_opt_hparams* hp, int32_t lin_stride, int64_t table_stride, mi_stream_t stream) {
  if (int32_t rc = check_hp("sparse_apply_fused", hp)) return rc;
  if (!table) table_stride = 0;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "sparse_apply_fused", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;

  MI_REQUIRE(lin_stride >= 1, "sparse_apply_fused: lin_stride=%d", lin_stride);
  MI_REQUIRE(n_max >= 0 && F > 0, "sparse_apply_fused: n_max=%lld F=%d", (long long)n_max, F);
  if (n_max == 0) return MI_OK;
  MI_REQUIRE(table || lin_w, "sparse_apply_fused: nothing to update");
  MI_REQUIRE(uniq_rows && seg_start && sorted_entry && num_uniq, "sparse_apply_fused: null index buffer");
  MI_REQUIRE(!table || (E >= 4 && E <= 256 && (E & 3) == 0 && mi::aligned16(table)),
             "sparse_apply_fused: E multiple of 4 in [4,256], 16-byte alignment");
  MI_REQUIRE(!table || d_concat || d_logit_fm, "sparse_apply_fused: table update needs d_concat and/or d_logit_fm");
  MI_REQUIRE(!d_concat || (ld_dconcat >= (int64_t)F * E && (ld_dconcat & 3) == 0 && mi::aligned16(d_concat)),
             "sparse_apply_fused: d_concat leading dimension / alignment");
  MI_REQUIRE(!d_logit_fm || (sumv ? mi::aligned16(sumv) : d_concat != nullptr),
             "sparse_apply_fused: FM gradient needs sumv, or a d_concat that already carries d_logit_fm * sumv");
  MI_REQUIRE(!lin_w || d_logit_lin, "sparse_apply_fused: lin_w needs d_logit_lin");
  const bool need0 = hp->kind != MI_OPT_SGD;
  const bool need1 = hp->kind == MI_OPT_ADAM || hp->kind == MI_OPT_FTRL || hp->kind == MI_OPT_RMSPROP;
  MI_REQUIRE(!table || ((!need0 || t_slot0) && (!need1 || t_slot1)), "sparse_apply_fused: table slots missing");
  MI_REQUIRE(!lin_w || ((!need0 || l_slot0) && (!need1 || l_slot1)), "sparse_apply_fused: linear slots missing");
  const int lpr = table ? lanes_per_row(E) : 1;
  const int64_t blocks = mi::ceil_div(n_max * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "sparse_apply_fused: grid too large");
  const Hp h = make_hp(hp);
  const FusedGrad fg{d_concat, ld_dconcat, sumv, d_logit_fm, d_logit_lin, F, 0};
  ApplyArgs a{table, t_slot0, t_slot1, lin_w, l_slot0, l_slot1, last_step, uniq_rows, seg_start, sorted_entry,
              num_uniq, nullptr, nullptr, E, step};
  a.ls = lin_stride; a.ts = ts;
  a.st = mi::step_state();
  MI_DISPATCH_LPR(lpr, (sparse_apply_k<L, true><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(a, h, fg)));
  MI_CHECK_LAUNCH("sparse_apply_fused");
  if (n_max > kLongSeg) {
    MI_DISPATCH_LPR(lpr, (sparse_apply_long_k<L, true><<<dim3(long_grid(n_max)), dim3(kBlock), 0, mi::as_stream(stream)>>>(a, h, fg)));
    MI_CHECK_LAUNCH("sparse_apply_fused(long segments)");
  }
  return MI_OK;
}

int32_t mi_entry_grads_segsum(const float* rows, const int32_t* seg_start, const int32_t* sorted_entry, int64_t u_begin,
                              int64_t u_count, const float* d_concat, int64_t ld_dconcat, const float* sumv,
                              const float* d_logit_fm, const float* d_logit_lin, int64_t b0, int32_t F, int32_t E,
                              float* out_rows, float* out_lin, int64_t out_row0, int64_t rows_stride, int64_t out_stride,
                              mi_stream_t stream) {
  MI_REQUIRE(rows_stride == 0 || (rows_stride >= E && (rows_stride & 3) == 0), "entry_grads_segsum: rows_stride=%lld", (long long)rows_stride);
  MI_REQUIRE(out_stride == 0 || (out_stride >= (out_rows ? E : 1) && (!out_rows || (out_stride & 3) == 0)),
             "entry_grads_segsum: out_stride=%lld (0 = out_rows E apart and out_lin 1 apart, else one record per request)", (long long)out_stride);
  MI_REQUIRE(u_begin >= 0 && u_count >= 0 && u_begin + u_count <= INT32_MAX && F > 0 && b0 >= 0, "entry_grads_segsum: u_begin=%lld u_count=%lld",
             (long long)u_begin, (long long)u_count);
  MI_REQUIRE(out_row0 >= 0 && out_row0 <= u_begin, "entry_grads_segsum: out_row0=%lld must lie in [0, u_begin=%lld]", (long long)out_row0,
             (long long)u_begin);
  if (u_count == 0) return MI_OK;
  MI_REQUIRE(seg_start && sorted_entry && (out_rows || out_lin), "entry_grads_segsum: null buffer");
  MI_REQUIRE(!out_rows || (E >= 4 && E <= 256 && (E & 3) == 0 && mi::aligned16(out_rows) && (d_concat || d_logit_fm)),
             "entry_grads_segsum: row gradients need E multiple of 4 in [4,256] and d_concat and/or d_logit_fm");
  MI_REQUIRE(!d_concat || (ld_dconcat >= (int64_t)F * E && (ld_dconcat & 3) == 0 && mi::aligned16(d_concat)),
             "entry_grads_segsum: d_concat leading dimension / alignment");
  MI_REQUIRE(!d_logit_fm || (sumv && rows && mi::aligned16(sumv) && mi::aligned16(rows)), "entry_grads_segsum: the FM gradient needs sumv and the rows");
  MI_REQUIRE(!out_lin || d_logit_lin, "entry_grads_segsum: out_lin needs d_logit_lin");
  const int lpr = out_rows ? lanes_per_row(E) : 1;
  const int64_t blocks = mi::ceil_div(u_count * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "entry_grads_segsum: grid too large");
  const FusedGrad fg{d_concat, ld_dconcat, sumv, d_logit_fm, d_logit_lin, F, b0};
  ApplyArgs a{};
  a.table = out_rows ? const_cast<float*>(rows) : nullptr;     // read only: w of d fm / d v = sumv - w
  if (out_rows && !rows) a.table = out_rows;                    // (no FM term: w is never read; any valid pointer turns the row part on)
  a.lin_w = out_lin;                                            // (non-null turns the linear part on; never read)
  a.seg_start = seg_start; a.sorted_entry = sorted_entry;
  // request u is written at row u - out_row0 of the out buffers (the kernels index by u: the bases are moved back; only
  // u >= u_begin >= out_row0 is ever touched)
  a.E = E;
  a.os = out_stride ? out_stride : E; a.ols = out_stride ? out_stride : 1;
  a.out_rows = out_rows ? reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(out_rows) - static_cast<uintptr_t>(out_row0) * a.os * sizeof(float)) : nullptr;
  a.out_lin = out_lin ? reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(out_lin) - static_cast<uintptr_t>(out_row0) * a.ols * sizeof(float)) : nullptr;
  a.u_begin = (int)u_begin; a.u_count = (int)u_count; a.ls = 1;
  a.ts = rows_stride ? rows_stride : E;                          // (rows: the exchange's receive buffer)
  const Hp h{};
  MI_DISPATCH_LPR(lpr, (sparse_apply_k<L, true, true><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(a, h, fg)));
  MI_CHECK_LAUNCH("entry_grads_segsum");
  if (u_count > 0) {
    MI_DISPATCH_LPR(lpr, (sparse_apply_long_k<L, true, true><<<dim3(long_grid(u_count)), dim3(kBlock), 0, mi::as_stream(stream)>>>(a, h, fg)));
    MI_CHECK_LAUNCH("entry_grads_segsum(long segments)");
  }
  return MI_OK;
}

int32_t mi_selftest_sqrt(uint32_t first_bits, int64_t count, uint64_t* mismatches, mi_stream_t stream) {
  MI_REQUIRE(count >= 0 && static_cast<uint64_t>(first_bits) + static_cast<uint64_t>(count) <= (1ull << 32), "selftest_sqrt: range leaves 32 bits");
  if (count == 0) return MI_OK;
  MI_REQUIRE(mismatches, "selftest_sqrt: null buffer");
  const int64_t blocks = mi::ceil_div(count, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "selftest_sqrt: grid too large");
  selftest_sqrt_k<<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
      first_bits, count, reinterpret_cast<unsigned long long*>(mismatches));
  MI_CHECK_LAUNCH("selftest_sqrt");
  return MI_OK;
}

int32_t mi_catchup_gap_keys(const int32_t* uniq_rows, const int32_t* num_uniq, const int32_t* last_step, int64_t n_max,
                            int32_t step_to, int32_t* keys, int32_t lin_stride, mi_stream_t stream) {
  MI_REQUIRE(n_max >= 0 && step_to >= 0 && lin_stride >= 1, "catchup_gap_keys: n_max=%lld", (long long)n_max);
  if (n_max == 0) return MI_OK;
  MI_REQUIRE(uniq_rows && num_uniq && last_step && keys, "catchup_gap_keys: null buffer");
  gap_keys_k<<<dim3((unsigned)mi::ceil_div(n_max, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
      uniq_rows, num_uniq, last_step, n_max, step_to, keys, lin_stride, mi::step_state());
  MI_CHECK_LAUNCH("catchup_gap_keys");
  return MI_OK;
}

int32_t mi_sparse_catchup(float* table, float* t_m, float* t_v, float* lin_w, float* l_m, float* l_v,
                          int32_t* last_step, const int32_t* uniq_rows, const int32_t* num_uniq,
                          int64_t n_max, int32_t E, int32_t step_to, const float* lr_table,
                          float beta1, float beta2, float epsilon, int32_t flags, int32_t lin_stride,
                          int64_t table_stride, mi_stream_t stream) {
  MI_REQUIRE(n_max >= 0 && step_to >= 0 && lin_stride >= 1, "sparse_catchup: n_max=%lld step_to=%d", (long long)n_max, step_to);
  if (!table) table_stride = 0;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "sparse_catchup", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;

  MI_REQUIRE((flags & ~(MI_CATCHUP_DEFER_SLOTS | MI_CATCHUP_BOUNDED | MI_CATCHUP_KEEP_STAMPS)) == 0, "sparse_catchup: flags=%d", flags);
  const bool keep_stamps = (flags & MI_CATCHUP_KEEP_STAMPS) != 0;
  const int32_t defer_slots = flags & MI_CATCHUP_DEFER_SLOTS;
  // the bounded form divides by rcp(sqrt(v) + eps): eps must keep that sum a normal number (TF's default 1e-8 does)
  const bool bounded = (flags & MI_CATCHUP_BOUNDED) != 0 && epsilon >= 1e-30f && beta2 > 0.f && beta2 <= 1.f;
  if (n_max == 0 || (step_to == 0 && !mi::step_state())) return MI_OK;
  MI_REQUIRE(last_step && lr_table, "sparse_catchup: null buffer");
  MI_REQUIRE(table || lin_w, "sparse_catchup: nothing to update");
  MI_REQUIRE(!table || (t_m && t_v && E >= 4 && E <= 256 && (E & 3) == 0 && mi::aligned16(table)),
             "sparse_catchup: table needs m, v, E multiple of 4 in [4,256]");
  MI_REQUIRE(!lin_w || (l_m && l_v), "sparse_catchup: lin_w needs m, v");
  MI_REQUIRE(!uniq_rows || num_uniq, "sparse_catchup: uniq_rows without num_uniq");
  const bool defer = defer_slots != 0 && uniq_rows != nullptr;
  if (lin_w) {
    catchup_lin_k<<<dim3((unsigned)mi::ceil_div(n_max, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
        lin_w, l_m, l_v, last_step, uniq_rows, num_uniq, n_max, step_to, lr_table, beta1, beta2, epsilon, defer, lin_stride, mi::step_state(),
        bounded);
    MI_CHECK_LAUNCH("sparse_catchup(wide part)");
  }
  // the wide part alone (two Adams, or its replay on a stream of its own): the row kernel would only write the stamps —
  // nothing at all when those are deferred or kept
  if (!table && (defer || keep_stamps)) return MI_OK;
  const int lpr = table ? lanes_per_row(E) : 1;
  const int64_t blocks = mi::ceil_div(n_max * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "sparse_catchup: grid too large");
  if (bounded && table) {
    // a pipelined grid: a few resident workgroups per CU, every lane group walks its share of the rows
    int64_t pb = mi::env_int("MI_CATCHUP_BLOCKS", 1024);          // 4 workgroups per CU resident: half the wave slots stay free for the side streams' small kernels (A/B on one box, 3 x alternating: 2.864 vs 2.881 ms per step at 2048)
    while (pb > 1 && pb > blocks) pb >>= 1;                       // (a power of two: the kernel's wave -> chunk map)
    // MI_CATCHUP_DEPTH=2: two rounds' row state in flight per lane group instead of one (see the kernel).  Alone on the
    // GPU it is the faster kernel (tools/catchup_bench.py: 0.450 -> 0.411 ms at 1,024 workgroups, 0.398 at 2,048); inside
    // the step it is not (A/B on one box, three alternating runs: 2.82-2.96 vs 2.84-2.95 ms) — it finishes earlier against
    // the next batch's sort on the side stream, more of which then lands on the gather (189-216 -> 247-269 us).  Default 1.
    if (mi::env_int("MI_CATCHUP_DEPTH", 1) >= 2) {
      MI_DISPATCH_LPR(lpr, (sparse_catchup_bounded_k<L, true><<<dim3((unsigned)pb), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                               table, t_m, t_v, last_step, uniq_rows, num_uniq, n_max, E, step_to, lr_table, beta1, beta2,
                               epsilon, defer, lin_stride, mi::step_state(), keep_stamps, ts)));
    } else {
      MI_DISPATCH_LPR(lpr, (sparse_catchup_bounded_k<L, false><<<dim3((unsigned)pb), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                               table, t_m, t_v, last_step, uniq_rows, num_uniq, n_max, E, step_to, lr_table, beta1, beta2,
                               epsilon, defer, lin_stride, mi::step_state(), keep_stamps, ts)));
    }
  } else {
    MI_DISPATCH_LPR(lpr, (sparse_catchup_k<L><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                             table, t_m, t_v, last_step, uniq_rows, num_uniq, n_max, E, step_to, lr_table, beta1, beta2,
                             epsilon, defer, lin_stride, mi::step_state(), keep_stamps, ts)));
  }
  MI_CHECK_LAUNCH("sparse_catchup");
  return MI_OK;
}

}  // extern "C"
