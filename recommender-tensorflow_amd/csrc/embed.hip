// Embedding gather + wide linear reduce + FM second-order term, forward and per-entry backward.
//
// Replaces (reference file:line)  tf.feature_column.linear_model      trainers/deep_fm.py:39
//                                 embedding_column + input_layer      trainers/deep_fm.py:52-54
//                                 numeric embedding                   trainers/deep_fm.py:62-73
//                                 FM (sum-square minus square-sum)    trainers/deep_fm.py:79-87
//
// HBM-bound.  One *group* of LPR = pow2ceil(E/4) lanes owns one example; every lane keeps a
// float4 slice of the row, so a wave-instruction moves 64/LPR whole rows (E=64: four 256-B rows,
// 1 KiB per instruction) and the sum over fields needs no cross-lane traffic.  Eight row loads
// are in flight per lane; ids and linear weights are fetched one field per lane and handed
// round the group with ds_bpermute.  The FM reduction over e is a butterfly inside the group.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kRowsInFlight = 8;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, LPR);
  return x;
}

// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(kBlock) void embed_fm_linear_fwd_k(
    const float* __restrict__ table, const float* __restrict__ lin_w,
    const int64_t* __restrict__ field_off, const int32_t* __restrict__ ids, int64_t B, int F, int E,
    float* __restrict__ concat, int64_t ldc, float* __restrict__ sumv, float* __restrict__ fm,
    float* __restrict__ lin, float* __restrict__ amax_rows, int ls, int64_t ts) {
  constexpr int U = kRowsInFlight;
  float mx = 0.f;                   // largest |row element| seen (scale of the layer-1 GEMM operand)
  const int64_t g = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  const bool valid = g < B;
  const int64_t b = valid ? g : 0;  // clamped: loads stay legal, stores are predicated on `valid`
  const bool lane_on = 4 * l < E;
  const int eo = 4 * l;

  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  float lacc = 0.f;
  const int32_t* idrow = ids + b * F;
  float* crow = concat + b * ldc;

  if constexpr (LPR >= 8) {
    for (int fb = 0; fb < F; fb += LPR) {
      const int fl = fb + l;
      int32_t myrow = 0;
      if (fl < F) {
        myrow = static_cast<int32_t>(field_off[fl] + idrow[fl]);
        if (lin_w) lacc += lin_w[static_cast<int64_t>(myrow) * ls];
      }
      const int nf = min(LPR, F - fb);
      for (int j0 = 0; j0 < nf; j0 += U) {
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t row = __shfl(myrow, (j0 + u) & (LPR - 1), LPR);
          r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j0 + u < nf && lane_on) r[u] = ld4(table + static_cast<int64_t>(row) * ts + eo);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (j0 + u < nf) {
            s.x += r[u].x; s.y += r[u].y; s.z += r[u].z; s.w += r[u].w;
            q.x += __fmul_rn(r[u].x, r[u].x); q.y += __fmul_rn(r[u].y, r[u].y);
            q.z += __fmul_rn(r[u].z, r[u].z); q.w += __fmul_rn(r[u].w, r[u].w);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[u].x), fabsf(r[u].y))), fmaxf(fabsf(r[u].z), fabsf(r[u].w)));
            if (valid && lane_on && concat) st4(crow + static_cast<int64_t>(fb + j0 + u) * E + eo, r[u]);
          }
        }
      }
    }
  } else {
    for (int f0 = 0; f0 < F; f0 += U) {
      int32_t row[U];
      float4 r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + u;
        row[u] = (f < F) ? static_cast<int32_t>(field_off[f] + idrow[f]) : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f0 + u < F && lane_on) r[u] = ld4(table + static_cast<int64_t>(row[u]) * ts + eo);
        if (lin_w && f0 + u < F && ((f0 + u) & (LPR - 1)) == l) lacc += lin_w[static_cast<int64_t>(row[u]) * ls];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (f0 + u < F) {
          s.x += r[u].x; s.y += r[u].y; s.z += r[u].z; s.w += r[u].w;
          q.x += __fmul_rn(r[u].x, r[u].x); q.y += __fmul_rn(r[u].y, r[u].y);
          q.z += __fmul_rn(r[u].z, r[u].z); q.w += __fmul_rn(r[u].w, r[u].w);
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[u].x), fabsf(r[u].y))), fmaxf(fabsf(r[u].z), fabsf(r[u].w)));
          if (valid && lane_on && concat) st4(crow + static_cast<int64_t>(f0 + u) * E + eo, r[u]);
        }
      }
    }
  }

  // deep_fm.py:81-87: 0.5 * sum_e( (sum_d v)^2 - sum_d v^2 )
  // tf.square then subtract: products rounded on their own (no fma), so one field gives exactly 0
  float t = ((__fmul_rn(s.x, s.x) - q.x) + (__fmul_rn(s.y, s.y) - q.y)) +
            ((__fmul_rn(s.z, s.z) - q.z) + (__fmul_rn(s.w, s.w) - q.w));
  t = group_sum<LPR>(t);
  lacc = group_sum<LPR>(lacc);
  if (valid) {
    if (sumv && lane_on) st4(sumv + b * E + eo, s);
    if (l == 0) {
      if (fm) fm[b] = 0.5f * t;
      if (lin) lin[b] = lacc;
    }
  }
  if (amax_rows) mi_amax_publish(amax_rows, mx);
}

// ------------------------------------------------------------------------------------------
// The same gather with the input_layer concat written as fp16 high/low PLANES (mi_planes_t, gemm_pl.hip):
// the operand of the layer-1 GEMMs.  A lane group owns a whole example, so it knows the example's abs-max
// before it writes: all F rows of the example stay in registers (FC >= F float4 per lane, all loads in
// flight at once), then one power-of-two exponent per example, then 8-byte stores (4 k of one plane per
// lane; the LPR lanes of a group write each field's 4E bytes of planes contiguously).  E % 16 == 0.
typedef _Float16 e_h16x2 __attribute__((ext_vector_type(2)));
typedef float e_f32x2 __attribute__((ext_vector_type(2)));
template <int LPR, int FC>
__global__ __launch_bounds__(kBlock) void embed_fm_planes_fwd_k(
    const float* __restrict__ table, const int64_t* __restrict__ field_off, const int32_t* __restrict__ ids, int64_t B,
    int F, int E, float* __restrict__ sumv, float* __restrict__ fm, char* __restrict__ planes, int64_t ldp_b,
    int32_t* __restrict__ row_exp, float* __restrict__ amax_rows, const float* __restrict__ x_num, int nd, int tail_cols,
    int64_t ts) {
  static_assert(LPR >= 8, "planes gather: E >= 32");
  const int64_t g = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  const bool valid = g < B;
  const int64_t b = valid ? g : 0;
  const bool lane_on = 4 * l < E;
  const int eo = 4 * l;
  const int32_t* idrow = ids + b * F;
  float4 r[FC];
#pragma unroll
  for (int fb = 0; fb < FC; fb += LPR) {
    const int fl = fb + l;
    int32_t myrow = 0;
    if (fl < F) myrow = static_cast<int32_t>(field_off[fl] + idrow[fl]);
#pragma unroll
    for (int j = 0; j < LPR; ++j) {
      if (fb + j < FC) {
        const int32_t row = __shfl(myrow, j, LPR);
        r[fb + j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fb + j < F && lane_on) r[fb + j] = ld4(table + static_cast<int64_t>(row) * ts + eo);
      }
    }
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  float mx = 0.f;
#pragma unroll
  for (int f = 0; f < FC; ++f) {
    if (f < F) {
      s.x += r[f].x; s.y += r[f].y; s.z += r[f].z; s.w += r[f].w;
      q.x += __fmul_rn(r[f].x, r[f].x); q.y += __fmul_rn(r[f].y, r[f].y);
      q.z += __fmul_rn(r[f].z, r[f].z); q.w += __fmul_rn(r[f].w, r[f].w);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[f].x), fabsf(r[f].y))), fmaxf(fabsf(r[f].z), fabsf(r[f].w)));
    }
  }
  // The canned estimators' raw numeric columns: the values themselves follow the embedding columns in the
  // input_layer concat — columns F E .. F E + tail_cols of the planes (nd values, then zeros: whole k-tiles).  They share
  // the example's exponent, so they join its abs-max; the FM sums above do not see them (no FM term there).
  constexpr int kTailMax = 4;                      // float4 per lane: tail_cols <= 16 LPR
  float4 tv[kTailMax];
#pragma unroll
  for (int c = 0; c < kTailMax; ++c) {
    tv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int col = c * 4 * LPR + 4 * l;
    if (col < tail_cols && valid) {
      const float* xr = x_num + b * nd;
      if (col < nd) tv[c].x = xr[col];
      if (col + 1 < nd) tv[c].y = xr[col + 1];
      if (col + 2 < nd) tv[c].z = xr[col + 2];
      if (col + 3 < nd) tv[c].w = xr[col + 3];
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(tv[c].x), fabsf(tv[c].y))), fmaxf(fabsf(tv[c].z), fabsf(tv[c].w)));
    }
  }
  float t = ((__fmul_rn(s.x, s.x) - q.x) + (__fmul_rn(s.y, s.y) - q.y)) +
            ((__fmul_rn(s.z, s.z) - q.z) + (__fmul_rn(s.w, s.w) - q.w));
  t = group_sum<LPR>(t);
  float gmx = mx;                                  // the example's abs-max over all fields and lanes
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) gmx = fmaxf(gmx, __shfl_xor(gmx, off, LPR));
  const int ex = static_cast<int>((__float_as_uint(gmx) >> 23) & 0xffu);
  const int sx = max(-100, min(100, 141 - ex));
  const float sc = __uint_as_float(static_cast<uint32_t>(127 + sx) << 23);
  if (valid) {
    if (sumv && lane_on) st4(sumv + b * E + eo, s);
    if (l == 0) {
      if (fm) fm[b] = 0.5f * t;
      row_exp[b] = sx;
    }
    // 4 k of one example: 8 bytes of the high plane, 8 of the low one, in k-block (k >> 4) at planes + (k >> 4) * ldp_b
    auto store4 = [&](int k, const float4& v) {
      const e_f32x2 u01 = {v.x * sc, v.y * sc}, u23 = {v.z * sc, v.w * sc};
      const e_h16x2 h01 = __builtin_convertvector(u01, e_h16x2), h23 = __builtin_convertvector(u23, e_h16x2);
      const e_f32x2 d01 = {u01[0] - static_cast<float>(h01[0]), u01[1] - static_cast<float>(h01[1])};
      const e_f32x2 d23 = {u23[0] - static_cast<float>(h23[0]), u23[1] - static_cast<float>(h23[1])};
      const e_h16x2 l01 = __builtin_convertvector(d01, e_h16x2), l23 = __builtin_convertvector(d23, e_h16x2);
      char* d = planes + b * 64 + (k & 15) * 2 + (k >> 4) * ldp_b;
      // (tried here, no gain: non-temporal stores, 0.188 vs 0.175 ms per launch; a DPP quad exchange so that every
      // lane stores a whole 16-byte piece of the block instead of two 8-byte halves, 0.168-0.176 vs 0.166-0.18)
      *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
      *reinterpret_cast<uint2*>(d + 32) = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
    };
    if (lane_on) {
#pragma unroll
      for (int f = 0; f < FC; ++f)
        if (f < F) store4(f * E + eo, r[f]);
    }
#pragma unroll
    for (int c = 0; c < kTailMax; ++c) {
      const int col = c * 4 * LPR + 4 * l;
      if (col < tail_cols) store4(F * E + col, tv[c]);
    }
  }
  if (amax_rows) mi_amax_publish(amax_rows, mx);
}

// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(kBlock) void gather_rows_k(const float* __restrict__ table,
                                                        const float* __restrict__ lin_w,
                                                        const int32_t* __restrict__ rows, int64_t n,
                                                        int E, float* __restrict__ out_rows,
                                                        float* __restrict__ out_lin, int ls, int64_t ts,
                                                        int64_t os, int64_t ols) {
  constexpr int U = kRowsInFlight;
  const int64_t g = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  const bool lane_on = 4 * l < E;
  const int64_t i0 = g * U;
  int32_t row[U];
  float4 r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) row[u] = (i0 + u < n) ? rows[i0 + u] : 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (table && i0 + u < n && lane_on) r[u] = ld4(table + static_cast<int64_t>(row[u]) * ts + 4 * l);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (i0 + u < n) {
      if (table && lane_on) st4(out_rows + (i0 + u) * os + 4 * l, r[u]);
      if (lin_w && out_lin && (u & (LPR - 1)) == l) out_lin[(i0 + u) * ols] = lin_w[static_cast<int64_t>(row[u]) * ls];
    }
  }
}

// ------------------------------------------------------------------------------------------
// d_rows[p(b,f),:] = d_concat[b,f,:] + dlf[b] * (sumv[b,:] - v(b,f));  d_lin[p(b,f)] = dll[b]
// v(b,f) = concat[b,f,:], or rows[p(b,f),:] when the gathered rows are held in slot order instead
template <int LPR>
__global__ __launch_bounds__(kBlock) void embed_fm_linear_bwd_k(
    const float* __restrict__ d_concat, int64_t lddc, const float* __restrict__ concat, int64_t ldc,
    const float* __restrict__ rows, const float* __restrict__ sumv, const float* __restrict__ dlf,
    const float* __restrict__ dll, const int32_t* __restrict__ pos, int64_t B, int F, int E,
    float* __restrict__ d_rows, float* __restrict__ d_lin) {
  const int64_t b = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  if (b >= B) return;
  const bool lane_on = 4 * l < E;
  const int eo = 4 * l;
  const float gf = dlf ? dlf[b] : 0.f;
  float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dlf && lane_on) sv = ld4(sumv + b * E + eo);
  const float gl = dll ? dll[b] : 0.f;
#pragma unroll 4
  for (int f = 0; f < F; ++f) {
    const int64_t p = pos ? static_cast<int64_t>(pos[b * F + f]) : b * F + f;
    if (lane_on) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d_concat) g = ld4(d_concat + b * lddc + static_cast<int64_t>(f) * E + eo);
      if (dlf) {
        const float4 v = rows ? ld4(rows + p * E + eo) : ld4(concat + b * ldc + static_cast<int64_t>(f) * E + eo);
        g.x += gf * (sv.x - v.x); g.y += gf * (sv.y - v.y);
        g.z += gf * (sv.z - v.z); g.w += gf * (sv.w - v.w);
      }
      st4(d_rows + p * E + eo, g);
    }
    if (d_lin && (f & (LPR - 1)) == l) d_lin[p] = gl;
  }
}

// wide part alone (trainers/linear.py: no embedding table): lin[b] = sum_f lin_w[row(b,f)]
__global__ __launch_bounds__(kBlock) void linear_only_fwd_k(const float* __restrict__ lin_w,
                                                            const int64_t* __restrict__ field_off,
                                                            const int32_t* __restrict__ ids, int64_t B,
                                                            int F, float* __restrict__ lin, int ls) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int f = 0; f < F; ++f) acc += lin_w[(field_off[f] + ids[b * F + f]) * ls];
  lin[b] = acc;
}

__global__ __launch_bounds__(kBlock) void linear_only_bwd_k(const float* __restrict__ dll,
                                                            const int32_t* __restrict__ pos, int64_t n,
                                                            int F, float* __restrict__ d_lin) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  d_lin[pos ? static_cast<int64_t>(pos[i]) : i] = dll[i / F];
}

__global__ __launch_bounds__(kBlock) void global_rows_k(const int32_t* __restrict__ ids,
                                                        const int64_t* __restrict__ field_off,
                                                        int64_t n, int F, int32_t* __restrict__ rows) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) rows[i] = static_cast<int32_t>(field_off[i % F] + ids[i]);
}

// ------------------------------------------------------------------------------------------
// deep_fm.py:62-70 numeric embedding; one group per example, FM / sumv / lin patched in place.
template <int LPR>
__global__ __launch_bounds__(kBlock) void numeric_embed_fwd_k(
    const float* __restrict__ x, const float* __restrict__ V, const float* __restrict__ w_num,
    int64_t B, int nd, int E, float* __restrict__ concat, int64_t ldc, int64_t col0,
    float* __restrict__ sumv, float* __restrict__ fm, float* __restrict__ lin) {
  const int64_t g = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x & (LPR - 1);
  const bool valid = g < B;
  const int64_t b = valid ? g : 0;
  const bool lane_on = 4 * l < E;
  const int eo = 4 * l;
  float4 sn = make_float4(0.f, 0.f, 0.f, 0.f), qn = sn;
  float lacc = 0.f;
  for (int j = 0; j < nd; ++j) {
    const float xv = x[b * nd + j];
    if (w_num && (j & (LPR - 1)) == l) lacc += xv * w_num[j];
    if (lane_on) {
      const float4 vv = ld4(V + static_cast<int64_t>(j) * E + eo);
      const float4 r = make_float4(xv * vv.x, xv * vv.y, xv * vv.z, xv * vv.w);
      sn.x += r.x; sn.y += r.y; sn.z += r.z; sn.w += r.w;
      qn.x += r.x * r.x; qn.y += r.y * r.y; qn.z += r.z * r.z; qn.w += r.w * r.w;
      if (valid) st4(concat + b * ldc + col0 + static_cast<int64_t>(j) * E + eo, r);
    }
  }
  float t = 0.f;
  if (lane_on && sumv) {
    const float4 sc = ld4(sumv + b * E + eo);
    const float4 st = make_float4(sc.x + sn.x, sc.y + sn.y, sc.z + sn.z, sc.w + sn.w);
    t = ((st.x * st.x - sc.x * sc.x) - qn.x) + ((st.y * st.y - sc.y * sc.y) - qn.y) +
        ((st.z * st.z - sc.z * sc.z) - qn.z) + ((st.w * st.w - sc.w * sc.w) - qn.w);
    if (valid) st4(sumv + b * E + eo, st);
  }
  t = group_sum<LPR>(t);
  lacc = group_sum<LPR>(lacc);
  if (valid && l == 0) {
    if (fm) fm[b] += 0.5f * t;
    if (lin && w_num) lin[b] += lacc;
  }
}

// stage 1 of the numeric-embedding backward: each block reduces a slice of examples.
// part[blk, j, e] = sum_{b in slice} x[b,j] * g[b,j,e];  partw[blk, j] = sum_b dll[b]*x[b,j]
__global__ __launch_bounds__(kBlock) void numeric_embed_bwd_part_k(
    const float* __restrict__ x, const float* __restrict__ d_concat, int64_t lddc,
    const float* __restrict__ concat, int64_t ldc, int64_t col0, const float* __restrict__ sumv,
    const float* __restrict__ dlf, const float* __restrict__ dll, int64_t B, int nd, int E,
    int64_t rows_per_block, float* __restrict__ part, float* __restrict__ partw) {
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t b1 = min(B, b0 + rows_per_block);
  const int ne = nd * E;
  for (int c = threadIdx.x; c < ne; c += kBlock) {
    const int j = c / E, e = c - j * E;
    float acc = 0.f;
    for (int64_t b = b0; b < b1; ++b) {
      float g = d_concat ? d_concat[b * lddc + col0 + c] : 0.f;
      if (dlf) g += dlf[b] * (sumv[b * E + e] - concat[b * ldc + col0 + c]);
      acc += x[b * nd + j] * g;
    }
    part[static_cast<int64_t>(blockIdx.x) * ne + c] = acc;
  }
  if (partw) {
    for (int j = threadIdx.x; j < nd; j += kBlock) {
      float acc = 0.f;
      if (dll)
        for (int64_t b = b0; b < b1; ++b) acc += dll[b] * x[b * nd + j];
      partw[static_cast<int64_t>(blockIdx.x) * nd + j] = acc;
    }
  }
}

// Canned estimators' numeric columns (tf.feature_column.numeric_column in dnn_feature_columns /
// linear_feature_columns of trainers/linear_deep.py:32-39): the value itself is a column of the
// input_layer concat, and linear_model multiplies it by a [1,1] weight.  One thread per example;
// the linear terms are added in column order after the categorical sum (products rounded on their own).
__global__ __launch_bounds__(kBlock) void numeric_raw_fwd_k(const float* __restrict__ x, const float* __restrict__ w_num,
                                                            int64_t B, int nd, float* __restrict__ concat, int64_t ldc,
                                                            int64_t col0, int ncols, float* __restrict__ lin) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  const float* xr = x + b * nd;
  if (concat) {
    float* cr = concat + b * ldc + col0;
    for (int j = 0; j < ncols; ++j) cr[j] = j < nd ? xr[j] : 0.f;
  }
  if (lin && w_num) {
    float acc = lin[b];
    for (int j = 0; j < nd; ++j) acc = acc + xr[j] * w_num[j];
    lin[b] = acc;
  }
}

// partw[blk, j] = sum_{b in slice} dll[b] * x[b,j]  (reduce_parts_k adds the slices in order)
// (round 3: the rows of a block are spread over its threads and summed by a fixed shuffle tree — the first form had one
// thread per COLUMN walk the block's rows: 13 of 256 threads busy, 91 us for 3.4 MB at config 4)
__global__ __launch_bounds__(kBlock) void numeric_raw_bwd_part_k(const float* __restrict__ x, const float* __restrict__ dll,
                                                                 int64_t B, int nd, int64_t rows_per_block,
                                                                 float* __restrict__ partw) {
  __shared__ float red[kBlock / 64];
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t b1 = min(B, b0 + rows_per_block);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = 0; j < nd; ++j) {
    float acc = 0.f;
    for (int64_t b = b0 + threadIdx.x; b < b1; b += kBlock) acc += dll[b] * x[b * nd + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) t += red[w];
      partw[static_cast<int64_t>(blockIdx.x) * nd + j] = t;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(kBlock) void reduce_parts_k(const float* __restrict__ part, int nparts,
                                                         int width, float* __restrict__ out) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= width) return;
  float acc = 0.f;
  for (int p = 0; p < nparts; ++p) acc += part[static_cast<int64_t>(p) * width + c];
  out[c] = acc;
}

int lanes_per_row(int E) {
  int q = E / 4, l = 1;
  while (l < q) l <<= 1;
  return l;
}

constexpr int64_t kNumericRowsPerBlock = 256;

}  // namespace

#define MI_DISPATCH_LPR(lpr, CALL)          \
  switch (lpr) {                            \
    case 1: { constexpr int L = 1; CALL; } break;   \
    case 2: { constexpr int L = 2; CALL; } break;   \
    case 4: { constexpr int L = 4; CALL; } break;   \
    case 8: { constexpr int L = 8; CALL; } break;   \
    case 16: { constexpr int L = 16; CALL; } break; \
    case 32: { constexpr int L = 32; CALL; } break; \
    default: { constexpr int L = 64; CALL; } break; \
  }

static int32_t check_E(const char* who, int32_t E) {
  if (E < 4 || E > 256 || (E & 3)) {
    mi::set_error("%s: embedding size %d unsupported (multiple of 4 in [4,256])", who, E);
    return MI_ERR_UNSUPPORTED;
  }
  return MI_OK;
}

extern "C" {

int32_t mi_embed_fm_linear_fwd(const float* table, const float* lin_w, const int64_t* field_off,
                               const int32_t* ids, int64_t B, int32_t F, int32_t E, float* concat,
                               int64_t ld_concat, float* sumv, float* fm, float* lin, float* amax_rows,
                               int32_t lin_stride, int64_t table_stride, mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && F > 0 && lin_stride >= 1, "embed_fm_linear_fwd: B=%lld F=%d lin_stride=%d", (long long)B, F, lin_stride);
  if (!table) {  // wide part only
    MI_REQUIRE(!concat && !sumv && !fm && lin && lin_w && field_off && ids,
               "embed_fm_linear_fwd: without a table only lin can be produced");
    if (B == 0) return MI_OK;
    linear_only_fwd_k<<<dim3((unsigned)mi::ceil_div(B, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
        lin_w, field_off, ids, B, F, lin, lin_stride);
    MI_CHECK_LAUNCH("embed_fm_linear_fwd(linear only)");
    return MI_OK;
  }
  if (int32_t rc = check_E("embed_fm_linear_fwd", E)) return rc;
  MI_REQUIRE(table && field_off && ids, "embed_fm_linear_fwd: null buffer");
  MI_REQUIRE(concat || sumv || fm || lin || amax_rows, "embed_fm_linear_fwd: no output requested");
  MI_REQUIRE(!concat || (ld_concat >= (int64_t)F * E && (ld_concat & 3) == 0),
             "embed_fm_linear_fwd: ld_concat=%lld must be >= F*E and a multiple of 4", (long long)ld_concat);
  MI_REQUIRE(mi::aligned16(table) && (!concat || mi::aligned16(concat)) && (!sumv || mi::aligned16(sumv)),
             "embed_fm_linear_fwd: table/concat/sumv must be 16-byte aligned");
  MI_REQUIRE(!lin || lin_w, "embed_fm_linear_fwd: lin requested without lin_w");
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "embed_fm_linear_fwd", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;
  if (B == 0) return MI_OK;
  const int lpr = lanes_per_row(E);
  const int64_t blocks = mi::ceil_div(B * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "embed_fm_linear_fwd: grid too large");
  MI_DISPATCH_LPR(lpr, (embed_fm_linear_fwd_k<L><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           table, lin ? lin_w : nullptr, field_off, ids, B, F, E, concat, ld_concat, sumv, fm, lin,
                           amax_rows, lin_stride, ts)));
  MI_CHECK_LAUNCH("embed_fm_linear_fwd");
  return MI_OK;
}

int32_t mi_embed_fm_planes_fwd(const float* table, const int64_t* field_off, const int32_t* ids, int64_t B, int32_t F,
                               int32_t E, float* sumv, float* fm, const mi_planes_t* concat, float* amax_rows,
                               const float* x_num, int32_t n_numeric, int32_t tail_cols, int64_t table_stride, mi_stream_t stream) {
  if (int32_t rc = check_E("embed_fm_planes_fwd", E)) return rc;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "embed_fm_planes_fwd", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;

  MI_REQUIRE(B >= 0 && F > 0, "embed_fm_planes_fwd: B=%lld F=%d", (long long)B, F);
  if ((E & 15) || E < 32 || F > 48) {
    mi::set_error("embed_fm_planes_fwd: needs E a multiple of 16 >= 32 and F <= 48 (E=%d F=%d): use mi_embed_fm_linear_fwd + mi_split_rows", E, F);
    return MI_ERR_UNSUPPORTED;
  }
  MI_REQUIRE(table && field_off && ids && concat && concat->data && concat->row_exp, "embed_fm_planes_fwd: null buffer");
  MI_REQUIRE(concat->blk_stride >= B * 64 && (concat->blk_stride & 63) == 0 && mi::aligned16(concat->data) && mi::aligned16(table) &&
                 (!sumv || mi::aligned16(sumv)), "embed_fm_planes_fwd: planes block stride / alignment");
  MI_REQUIRE(!fm || sumv, "embed_fm_planes_fwd: fm needs sumv");
  MI_REQUIRE(tail_cols >= 0 && (tail_cols & 15) == 0 && n_numeric >= 0 && n_numeric <= tail_cols && (tail_cols == 0 || x_num),
             "embed_fm_planes_fwd: %d numeric columns in a tail of %d (whole 16-k blocks, x_num given)", n_numeric, tail_cols);
  if (B == 0) return MI_OK;
  const int lpr = lanes_per_row(E);
  if (tail_cols > 4 * E) {                       // (the kernel holds 4 float4 of tail per lane: 16 lpr >= 4 E columns)
    mi::set_error("embed_fm_planes_fwd: a tail of %d columns needs E >= %d (E=%d)", tail_cols, tail_cols / 4, E);
    return MI_ERR_UNSUPPORTED;
  }
  const int64_t blocks = mi::ceil_div(B * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "embed_fm_planes_fwd: grid too large");
  char* pd = static_cast<char*>(concat->data);
  const dim3 g((unsigned)blocks), blk(kBlock);
  hipStream_t st = mi::as_stream(stream);
#define MI_PL_GATHER(L, FCAP) embed_fm_planes_fwd_k<L, FCAP><<<g, blk, 0, st>>>(table, field_off, ids, B, F, E, sumv, fm, pd, concat->blk_stride, concat->row_exp, amax_rows, x_num, n_numeric, tail_cols, ts)
  if (F <= 32) {
    switch (lpr) {
      case 8: MI_PL_GATHER(8, 32); break;
      case 16: MI_PL_GATHER(16, 32); break;
      case 32: MI_PL_GATHER(32, 32); break;
      default: MI_PL_GATHER(64, 32); break;
    }
  } else {
    switch (lpr) {
      case 8: MI_PL_GATHER(8, 48); break;
      case 16: MI_PL_GATHER(16, 48); break;
      case 32: MI_PL_GATHER(32, 48); break;
      default: MI_PL_GATHER(64, 48); break;
    }
  }
#undef MI_PL_GATHER
  MI_CHECK_LAUNCH("embed_fm_planes_fwd");
  return MI_OK;
}

int32_t mi_gather_rows(const float* table, const float* lin_w, const int32_t* rows, int64_t n,
                       int32_t E, float* out_rows, float* out_lin, int32_t lin_stride, int64_t table_stride, int64_t out_stride,
                       mi_stream_t stream) {
  MI_REQUIRE(out_stride == 0 || (out_stride >= (table ? E : 1) && (!table || (out_stride & 3) == 0)),
             "gather_rows: out_stride=%lld (0 = rows E apart and weights 1 apart, else one record of out_stride floats per request)", (long long)out_stride);
  const int64_t os = out_stride ? out_stride : E, ols = out_stride ? out_stride : 1;
  if (!table) { E = 4; table_stride = 0; }   // wide part only: the row half of the kernel is off
  if (int32_t rc = check_E("gather_rows", E)) return rc;
  MI_REQUIRE(table_stride == 0 || (table_stride >= E && (table_stride & 3) == 0), "%s: table_stride=%lld (0 = E, else >= E and a multiple of 4)", "gather_rows", (long long)table_stride);
  const int64_t ts = table_stride ? table_stride : E;

  MI_REQUIRE(n >= 0 && lin_stride >= 1, "gather_rows: n=%lld lin_stride=%d", (long long)n, lin_stride);
  if (n == 0) return MI_OK;
  MI_REQUIRE(rows && ((table && out_rows) || (lin_w && out_lin)), "gather_rows: null buffer");
  MI_REQUIRE(!table || (mi::aligned16(table) && mi::aligned16(out_rows)), "gather_rows: 16-byte alignment");
  const int lpr = lanes_per_row(E);
  const int64_t groups = mi::ceil_div(n, kRowsInFlight);
  const int64_t blocks = mi::ceil_div(groups * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "gather_rows: grid too large");
  MI_DISPATCH_LPR(lpr, (gather_rows_k<L><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           table, lin_w, rows, n, E, out_rows, out_lin, lin_stride, ts, os, ols)));
  MI_CHECK_LAUNCH("gather_rows");
  return MI_OK;
}

int32_t mi_embed_fm_linear_bwd(const float* d_concat, int64_t ld_dconcat, const float* concat,
                               int64_t ld_concat, const float* rows, const float* sumv, const float* d_logit_fm,
                               const float* d_logit_lin, const int32_t* pos, int64_t B, int32_t F,
                               int32_t E, float* d_rows, float* d_lin, mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && F > 0, "embed_fm_linear_bwd: B=%lld F=%d", (long long)B, F);
  if (B == 0) return MI_OK;
  if (!d_rows) {  // wide part only
    MI_REQUIRE(d_lin && d_logit_lin && !d_concat && !d_logit_fm, "embed_fm_linear_bwd: without d_rows only d_lin can be produced");
    linear_only_bwd_k<<<dim3((unsigned)mi::ceil_div(B * F, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
        d_logit_lin, pos, B * F, F, d_lin);
    MI_CHECK_LAUNCH("embed_fm_linear_bwd(linear only)");
    return MI_OK;
  }
  if (int32_t rc = check_E("embed_fm_linear_bwd", E)) return rc;
  MI_REQUIRE(d_rows, "embed_fm_linear_bwd: d_rows is null");
  MI_REQUIRE(!d_logit_fm || ((concat || rows) && sumv), "embed_fm_linear_bwd: FM gradient needs concat (or rows) and sumv");
  MI_REQUIRE(!rows || (!concat && mi::aligned16(rows)), "embed_fm_linear_bwd: give concat or rows, not both");
  MI_REQUIRE(!d_concat || (ld_dconcat >= (int64_t)F * E && (ld_dconcat & 3) == 0 && mi::aligned16(d_concat)),
             "embed_fm_linear_bwd: d_concat leading dimension/alignment");
  MI_REQUIRE(!concat || (ld_concat >= (int64_t)F * E && (ld_concat & 3) == 0 && mi::aligned16(concat)),
             "embed_fm_linear_bwd: concat leading dimension/alignment");
  MI_REQUIRE(mi::aligned16(d_rows) && (!sumv || mi::aligned16(sumv)), "embed_fm_linear_bwd: alignment");
  MI_REQUIRE(!d_lin || d_logit_lin, "embed_fm_linear_bwd: d_lin requested without d_logit_lin");
  const int lpr = lanes_per_row(E);
  const int64_t blocks = mi::ceil_div(B * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "embed_fm_linear_bwd: grid too large");
  MI_DISPATCH_LPR(lpr, (embed_fm_linear_bwd_k<L><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           d_concat, ld_dconcat, concat, ld_concat, rows, sumv, d_logit_fm,
                           d_lin ? d_logit_lin : nullptr, pos, B, F, E, d_rows, d_lin)));
  MI_CHECK_LAUNCH("embed_fm_linear_bwd");
  return MI_OK;
}

int32_t mi_global_rows(const int32_t* ids, const int64_t* field_off, int64_t B, int32_t F,
                       int32_t* rows, mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && F > 0, "global_rows: B=%lld F=%d", (long long)B, F);
  if (B == 0) return MI_OK;
  MI_REQUIRE(ids && field_off && rows, "global_rows: null buffer");
  const int64_t n = B * F;
  const int64_t blocks = mi::ceil_div(n, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "global_rows: grid too large");
  global_rows_k<<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(ids, field_off, n, F, rows);
  MI_CHECK_LAUNCH("global_rows");
  return MI_OK;
}

int32_t mi_numeric_embed_fwd(const float* x, const float* V, const float* w_num, int64_t B,
                             int32_t n_d, int32_t E, float* concat, int64_t ld_concat, int64_t col0,
                             float* sumv, float* fm, float* lin, mi_stream_t stream) {
  if (int32_t rc = check_E("numeric_embed_fwd", E)) return rc;
  MI_REQUIRE(B >= 0 && n_d > 0, "numeric_embed_fwd: B=%lld n_d=%d", (long long)B, n_d);
  if (B == 0) return MI_OK;
  MI_REQUIRE(x && V && concat, "numeric_embed_fwd: null buffer");
  MI_REQUIRE(col0 >= 0 && (col0 & 3) == 0 && (ld_concat & 3) == 0 && ld_concat >= col0 + (int64_t)n_d * E,
             "numeric_embed_fwd: col0=%lld ld=%lld", (long long)col0, (long long)ld_concat);
  MI_REQUIRE(mi::aligned16(V) && mi::aligned16(concat) && (!sumv || mi::aligned16(sumv)),
             "numeric_embed_fwd: alignment");
  MI_REQUIRE(!fm || sumv, "numeric_embed_fwd: fm needs sumv");
  const int lpr = lanes_per_row(E);
  const int64_t blocks = mi::ceil_div(B * lpr, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "numeric_embed_fwd: grid too large");
  MI_DISPATCH_LPR(lpr, (numeric_embed_fwd_k<L><<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(
                           x, V, w_num, B, n_d, E, concat, ld_concat, col0, sumv, fm, lin)));
  MI_CHECK_LAUNCH("numeric_embed_fwd");
  return MI_OK;
}

size_t mi_numeric_embed_bwd_workspace_bytes(int64_t B, int32_t n_d, int32_t E) {
  const int64_t nb = mi::ceil_div(B > 0 ? B : 1, kNumericRowsPerBlock);
  return static_cast<size_t>(nb) * (static_cast<size_t>(n_d) * E + n_d) * sizeof(float);
}

int32_t mi_numeric_embed_bwd(const float* x, const float* d_concat, int64_t ld_dconcat,
                             const float* concat, int64_t ld_concat, int64_t col0, const float* sumv,
                             const float* d_logit_fm, const float* d_logit_lin, int64_t B,
                             int32_t n_d, int32_t E, float* dV, float* dw_num, void* workspace,
                             size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(B > 0 && n_d > 0 && E > 0, "numeric_embed_bwd: B=%lld n_d=%d E=%d", (long long)B, n_d, E);
  MI_REQUIRE(x && dV && workspace, "numeric_embed_bwd: null buffer");
  MI_REQUIRE(!d_logit_fm || (concat && sumv), "numeric_embed_bwd: FM gradient needs concat and sumv");
  if (workspace_bytes < mi_numeric_embed_bwd_workspace_bytes(B, n_d, E)) {
    mi::set_error("numeric_embed_bwd: workspace %zu < %zu", workspace_bytes,
                  mi_numeric_embed_bwd_workspace_bytes(B, n_d, E));
    return MI_ERR_WORKSPACE;
  }
  const int64_t nb = mi::ceil_div(B, kNumericRowsPerBlock);
  float* part = static_cast<float*>(workspace);
  float* partw = part + nb * n_d * E;
  hipStream_t st = mi::as_stream(stream);
  numeric_embed_bwd_part_k<<<dim3((unsigned)nb), dim3(kBlock), 0, st>>>(
      x, d_concat, ld_dconcat, concat, ld_concat, col0, sumv, d_logit_fm, d_logit_lin, B, n_d, E,
      kNumericRowsPerBlock, part, dw_num ? partw : nullptr);
  MI_CHECK_LAUNCH("numeric_embed_bwd(part)");
  reduce_parts_k<<<dim3((unsigned)mi::ceil_div(n_d * E, kBlock)), dim3(kBlock), 0, st>>>(part, (int)nb, n_d * E, dV);
  MI_CHECK_LAUNCH("numeric_embed_bwd(reduce)");
  if (dw_num) {
    reduce_parts_k<<<dim3((unsigned)mi::ceil_div(n_d, kBlock)), dim3(kBlock), 0, st>>>(partw, (int)nb, n_d, dw_num);
    MI_CHECK_LAUNCH("numeric_embed_bwd(reduce w)");
  }
  return MI_OK;
}

int32_t mi_numeric_raw_fwd(const float* x, const float* w_num, int64_t B, int32_t n_d, float* concat,
                           int64_t ld_concat, int64_t col0, int32_t n_cols, float* lin, mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && n_d > 0, "numeric_raw_fwd: B=%lld n_d=%d", (long long)B, n_d);
  if (B == 0) return MI_OK;
  MI_REQUIRE(x && (concat || lin), "numeric_raw_fwd: null buffer");
  MI_REQUIRE(!concat || (col0 >= 0 && n_cols >= n_d && ld_concat >= col0 + n_cols),
             "numeric_raw_fwd: col0=%lld n_cols=%d ld=%lld", (long long)col0, n_cols, (long long)ld_concat);
  MI_REQUIRE(!lin || w_num, "numeric_raw_fwd: lin needs w_num");
  const int64_t blocks = mi::ceil_div(B, kBlock);
  MI_REQUIRE(blocks <= INT32_MAX, "numeric_raw_fwd: grid too large");
  numeric_raw_fwd_k<<<dim3((unsigned)blocks), dim3(kBlock), 0, mi::as_stream(stream)>>>(x, lin ? w_num : nullptr, B, n_d, concat,
                                                                                        ld_concat, col0, n_cols, lin);
  MI_CHECK_LAUNCH("numeric_raw_fwd");
  return MI_OK;
}

size_t mi_numeric_raw_bwd_workspace_bytes(int64_t B, int32_t n_d) {
  return static_cast<size_t>(mi::ceil_div(B > 0 ? B : 1, kNumericRowsPerBlock)) * static_cast<size_t>(n_d) * sizeof(float);
}

int32_t mi_numeric_raw_bwd(const float* x, const float* d_logit_lin, int64_t B, int32_t n_d, float* dw_num,
                           void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(B > 0 && n_d > 0, "numeric_raw_bwd: B=%lld n_d=%d", (long long)B, n_d);
  MI_REQUIRE(x && d_logit_lin && dw_num && workspace, "numeric_raw_bwd: null buffer");
  if (workspace_bytes < mi_numeric_raw_bwd_workspace_bytes(B, n_d)) {
    mi::set_error("numeric_raw_bwd: workspace %zu < %zu", workspace_bytes, mi_numeric_raw_bwd_workspace_bytes(B, n_d));
    return MI_ERR_WORKSPACE;
  }
  const int64_t nb = mi::ceil_div(B, kNumericRowsPerBlock);
  float* partw = static_cast<float*>(workspace);
  hipStream_t st = mi::as_stream(stream);
  numeric_raw_bwd_part_k<<<dim3((unsigned)nb), dim3(kBlock), 0, st>>>(x, d_logit_lin, B, n_d, kNumericRowsPerBlock, partw);
  MI_CHECK_LAUNCH("numeric_raw_bwd(part)");
  reduce_parts_k<<<dim3((unsigned)mi::ceil_div(n_d, kBlock)), dim3(kBlock), 0, st>>>(partw, (int)nb, n_d, dw_num);
  MI_CHECK_LAUNCH("numeric_raw_bwd(reduce)");
  return MI_OK;
}

}  // extern "C"
