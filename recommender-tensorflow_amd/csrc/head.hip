// Logits sum + sigmoid cross-entropy head, layer_summary statistics, streaming eval counters.
//
// Replaces `logits += ...` (trainers/deep_fm.py:36,44,90,111), the estimator head
// tf.contrib.estimator.binary_classification_head (deep_fm.py:118-125; the prediction / loss /
// metric contract is spelled out by trainers/model_utils.py:9-54) and layer_summary
// (model_utils.py:4-6).  Reductions are two-stage with a fixed order: bitwise reproducible.
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kAucThresholds = 200;  // tf.metrics.auc default num_thresholds

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

__device__ __forceinline__ float block_sum(float x, float* red /*[4]*/) {
  x = wave_sum(x);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  const float r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

__device__ __forceinline__ float sigmoid_stable(float x) {
  const float e = expf(-fabsf(x));
  return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

__global__ __launch_bounds__(kBlock) void head_k(const float* __restrict__ lin,
                                                 const float* __restrict__ lin_bias,
                                                 const float* __restrict__ fm,
                                                 const float* __restrict__ dnn,
                                                 const uint8_t* __restrict__ labels, int64_t B,
                                                 float scale, float* __restrict__ logits,
                                                 float* __restrict__ d_logit,
                                                 float* __restrict__ partial,
                                                 float* __restrict__ partial_d) {
  __shared__ float red[4];
  const int64_t per = (B + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = blockIdx.x * per, b1 = min(B, b0 + per);
  const float lb = (lin && lin_bias) ? lin_bias[0] : 0.f;
  float acc = 0.f, dacc = 0.f;
  for (int64_t b = b0 + threadIdx.x; b < b1; b += kBlock) {
    float x = 0.f;                        // deep_fm.py:36
    if (lin) x += lin[b] + lb;            // :44   (linear_model adds its bias last)
    if (fm) x += fm[b];                   // :90
    if (dnn) x += dnn[b];                 // :111
    if (logits) logits[b] = x;
    if (labels) {
      const float y = labels[b] ? 1.f : 0.f;
      const float l = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      acc += l * scale;
      if (d_logit) {
        const float d = (sigmoid_stable(x) - y) * scale;
        d_logit[b] = d;
        dacc += d;
      }
    }
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0 && partial) partial[blockIdx.x] = tot;
  if (partial_d) {
    const float dtot = block_sum(dacc, red);
    if (threadIdx.x == 0) partial_d[blockIdx.x] = dtot;
  }
}

// (block 1, if launched, folds a second vector: the loss and the gradient sum in one launch)
__global__ __launch_bounds__(kBlock) void sum_partials_k(const float* __restrict__ partial, int n,
                                                         float* __restrict__ out, const float* __restrict__ partial2 = nullptr,
                                                         float* __restrict__ out2 = nullptr) {
  __shared__ float red[4];
  if (blockIdx.x == 1) { partial = partial2; out = out2; }
  if (!partial) return;                         // (block-uniform)
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) acc += partial[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = tot;
}

// ---- layer_summary ---------------------------------------------------------------------
struct Stats { float zeros, mn, mx, sum; };

__global__ __launch_bounds__(kBlock) void stats_part_k(const float* __restrict__ x, int64_t n,
                                                       Stats* __restrict__ part) {
  __shared__ float red[4][4];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = blockIdx.x * per, i1 = min(n, i0 + per);
  float z = 0.f, mn = INFINITY, mx = -INFINITY, s = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += kBlock) {
    const float v = x[i];
    z += (v == 0.f) ? 1.f : 0.f;
    mn = fminf(mn, v); mx = fmaxf(mx, v); s += v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    z += __shfl_xor(z, off, 64); s += __shfl_xor(s, off, 64);
    mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = z; red[w][1] = mn; red[w][2] = mx; red[w][3] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    Stats o{0.f, INFINITY, -INFINITY, 0.f};
    for (int k = 0; k < 4; ++k) {
      o.zeros += red[k][0]; o.mn = fminf(o.mn, red[k][1]); o.mx = fmaxf(o.mx, red[k][2]); o.sum += red[k][3];
    }
    part[blockIdx.x] = o;
  }
}

__global__ void stats_final_k(const Stats* __restrict__ part, int nparts, int64_t n, float* __restrict__ out4) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double z = 0, s = 0;
  float mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < nparts; ++k) {
    z += part[k].zeros; s += part[k].sum; mn = fminf(mn, part[k].mn); mx = fmaxf(mx, part[k].mx);
  }
  out4[0] = static_cast<float>(z / static_cast<double>(n));
  out4[1] = mn; out4[2] = mx;
  out4[3] = static_cast<float>(s / static_cast<double>(n));
}

// ---- eval counters -----------------------------------------------------------------------
// hist[y][k]: number of examples with label y whose sigmoid exceeds exactly k of the 200
// tf.metrics.auc thresholds (thresholds ascending => "p > th[j]" <=> j < k).
__global__ __launch_bounds__(kBlock) void eval_accumulate_k(const float* __restrict__ logits,
                                                            const uint8_t* __restrict__ labels,
                                                            int64_t B,
                                                            unsigned long long* __restrict__ hist,
                                                            unsigned long long* __restrict__ counts,
                                                            double* __restrict__ sums) {
  __shared__ float th[kAucThresholds];
  __shared__ unsigned int lh[2][kAucThresholds + 1];
  __shared__ unsigned int lc[8];
  __shared__ double ls[4][4];
  for (int j = threadIdx.x; j < kAucThresholds; j += kBlock) {
    float v;
    if (j == 0) v = static_cast<float>(0.0 - 1e-7);
    else if (j == kAucThresholds - 1) v = static_cast<float>(1.0 + 1e-7);
    else v = static_cast<float>(static_cast<double>(j) * 1.0 / static_cast<double>(kAucThresholds - 1));
    th[j] = v;
  }
  for (int j = threadIdx.x; j < 2 * (kAucThresholds + 1); j += kBlock) (&lh[0][0])[j] = 0;
  if (threadIdx.x < 8) lc[threadIdx.x] = 0;
  __syncthreads();
  double sl = 0, sp = 0, sy = 0;
  const int64_t per = (B + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = blockIdx.x * per, b1 = min(B, b0 + per);
  for (int64_t b = b0 + threadIdx.x; b < b1; b += kBlock) {
    const float x = logits[b];
    const int y = labels[b] ? 1 : 0;
    const float p = sigmoid_stable(x);
    int lo = 0, hi = kAucThresholds;        // k = #{j : th[j] < p}
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (th[mid] < p) lo = mid + 1; else hi = mid; }
    atomicAdd(&lh[y][lo], 1u);
    const int cls = p > 0.5f ? 1 : 0;       // model_utils.py:12
    atomicAdd(&lc[0], 1u);
    if (y) atomicAdd(&lc[1], 1u);
    if (cls) atomicAdd(&lc[2], 1u);
    if (cls == y) atomicAdd(&lc[3], 1u);
    if (cls && y) atomicAdd(&lc[4], 1u);
    if (cls && !y) atomicAdd(&lc[5], 1u);
    if (!cls && y) atomicAdd(&lc[6], 1u);
    const double xd = x;
    sl += fmax(xd, 0.0) - xd * y + log1p(exp(-fabs(xd)));
    sp += p; sy += y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sl += __shfl_xor(sl, off, 64); sp += __shfl_xor(sp, off, 64); sy += __shfl_xor(sy, off, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { ls[w][0] = sl; ls[w][1] = sp; ls[w][2] = sy; }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * (kAucThresholds + 1); j += kBlock) {
    const unsigned int v = (&lh[0][0])[j];
    if (v) atomicAdd(&hist[j], static_cast<unsigned long long>(v));
  }
  if (threadIdx.x < 7 && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], static_cast<unsigned long long>(lc[threadIdx.x]));
  if (threadIdx.x < 3) {
    const double v = (ls[0][threadIdx.x] + ls[1][threadIdx.x]) + (ls[2][threadIdx.x] + ls[3][threadIdx.x]);
    atomicAdd(&sums[threadIdx.x], v);
  }
}

int blocks_for(int64_t n) {
  int64_t b = mi::ceil_div(n, kBlock * 4);
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return static_cast<int>(b);
}

}  // namespace

extern "C" {

size_t mi_head_workspace_bytes(int64_t B) { (void)B; return 2 * kMaxBlocks * sizeof(float); }

int32_t mi_sigmoid_ce_head(const float* lin, const float* lin_bias, const float* fm,
                           const float* dnn, const uint8_t* labels, int64_t B, float loss_scale,
                           float* logits, float* loss_out, float* d_logit, float* d_logit_sum,
                           void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(B > 0, "sigmoid_ce_head: B=%lld", (long long)B);
  MI_REQUIRE(lin || fm || dnn, "sigmoid_ce_head: no logit component (deep_fm.py:33-34)");
  MI_REQUIRE(labels || (!loss_out && !d_logit), "sigmoid_ce_head: loss / gradient need labels");
  MI_REQUIRE(!(loss_out || d_logit_sum) || workspace, "sigmoid_ce_head: loss / gradient sum need a workspace");
  MI_REQUIRE(!d_logit_sum || d_logit, "sigmoid_ce_head: d_logit_sum needs d_logit");
  if ((loss_out || d_logit_sum) && workspace_bytes < mi_head_workspace_bytes(B)) {
    mi::set_error("sigmoid_ce_head: workspace %zu < %zu", workspace_bytes, mi_head_workspace_bytes(B));
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  const int nb = blocks_for(B);
  // one block: its totals ARE the results (what the fold of a single partial would give, bit for bit): one launch
  const bool single = nb == 1;
  float* partial = loss_out ? (single ? loss_out : static_cast<float*>(workspace)) : nullptr;
  // d_logit_sum = d loss / d linear bias (and d / d logits-layer bias when the DNN has no hidden layer)
  float* partial_d = d_logit_sum ? (single ? d_logit_sum : static_cast<float*>(workspace) + kMaxBlocks) : nullptr;
  head_k<<<dim3(nb), dim3(kBlock), 0, st>>>(lin, lin_bias, fm, dnn, labels, B, loss_scale, logits, d_logit, partial,
                                            partial_d);
  MI_CHECK_LAUNCH("sigmoid_ce_head");
  if (!single && (loss_out || d_logit_sum)) {
    sum_partials_k<<<dim3(2), dim3(kBlock), 0, st>>>(partial, nb, loss_out, partial_d, d_logit_sum);
    MI_CHECK_LAUNCH("sigmoid_ce_head(reduce)");
  }
  return MI_OK;
}

size_t mi_layer_stats_workspace_bytes(int64_t n) { (void)n; return kMaxBlocks * sizeof(Stats); }

int32_t mi_layer_stats(const float* x, int64_t n, float* out4, void* workspace, size_t workspace_bytes,
                       mi_stream_t stream) {
  MI_REQUIRE(n > 0 && x && out4 && workspace, "layer_stats: n=%lld", (long long)n);
  if (workspace_bytes < mi_layer_stats_workspace_bytes(n)) {
    mi::set_error("layer_stats: workspace %zu < %zu", workspace_bytes, mi_layer_stats_workspace_bytes(n));
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  const int nb = blocks_for(n);
  stats_part_k<<<dim3(nb), dim3(kBlock), 0, st>>>(x, n, static_cast<Stats*>(workspace));
  MI_CHECK_LAUNCH("layer_stats(part)");
  stats_final_k<<<dim3(1), dim3(64), 0, st>>>(static_cast<const Stats*>(workspace), nb, n, out4);
  MI_CHECK_LAUNCH("layer_stats(final)");
  return MI_OK;
}

int32_t mi_eval_accumulate(const float* logits, const uint8_t* labels, int64_t B, int64_t* hist,
                           int64_t* counts, double* sums, mi_stream_t stream) {
  MI_REQUIRE(B > 0 && logits && labels && hist && counts && sums, "eval_accumulate: B=%lld", (long long)B);
  const int nb = blocks_for(B);
  eval_accumulate_k<<<dim3(nb), dim3(kBlock), 0, mi::as_stream(stream)>>>(
      logits, labels, B, reinterpret_cast<unsigned long long*>(hist),
      reinterpret_cast<unsigned long long*>(counts), sums);
  MI_CHECK_LAUNCH("eval_accumulate");
  return MI_OK;
}

}  // extern "C"

// ---- (a10) the histogram half of layer_summary (tf.summary.histogram, model_utils.py:6) -------------------
namespace {
constexpr int kHistMaxLimits = 2048;
// counts[b] += 1 for b = number of limits <= x (std::upper_bound, as tensorflow::histogram::Histogram::Add);
// limits ascending, fp64; sums += (sum x, sum x^2) in fp64.  LDS histogram per block, integer atomics.
__global__ __launch_bounds__(256) void layer_histogram_k(const float* __restrict__ x, int64_t n, const double* __restrict__ limits,
                                                         int nl, unsigned long long* __restrict__ counts,
                                                         double* __restrict__ sums) {
  __shared__ double lim[kHistMaxLimits];
  __shared__ unsigned int cnt[kHistMaxLimits + 1];
  __shared__ double red[2][4];
  for (int i = threadIdx.x; i < nl; i += 256) lim[i] = limits[i];
  for (int i = threadIdx.x; i <= nl; i += 256) cnt[i] = 0u;
  __syncthreads();
  double s = 0.0, q = 0.0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    const double v = static_cast<double>(x[i]);
    int lo = 0, hi = nl;                       // first index with lim[idx] > v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (lim[mid] > v) hi = mid; else lo = mid + 1;
    }
    atomicAdd(&cnt[lo], 1u);
    s += v; q += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  for (int i = threadIdx.x; i <= nl; i += 256)
    if (cnt[i]) atomicAdd(counts + i, static_cast<unsigned long long>(cnt[i]));
  if (threadIdx.x == 0) {
    atomicAdd(sums, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(sums + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}
// get_binary_predictions / get_binary_losses (model_utils.py:9-36): per-example outputs of the head
__global__ __launch_bounds__(kBlock) void binary_predictions_k(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                                               int64_t B, float* __restrict__ logistic,
                                                               float* __restrict__ probabilities, int64_t* __restrict__ class_ids,
                                                               float* __restrict__ unreduced_loss) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  const float x = logits[b];
  const float p = sigmoid_stable(x);
  if (logistic) logistic[b] = p;
  if (probabilities) { probabilities[2 * b] = 1.f - p; probabilities[2 * b + 1] = p; }
  if (class_ids) class_ids[b] = p > 0.5f ? 1 : 0;
  if (unreduced_loss) {
    const float y = labels[b] ? 1.f : 0.f;
    unreduced_loss[b] = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
  }
}
}  // namespace

extern "C" int32_t mi_binary_predictions(const float* logits, const uint8_t* labels, int64_t B, float* logistic,
                                         float* probabilities, int64_t* class_ids, float* unreduced_loss, mi_stream_t stream) {
  MI_REQUIRE(B >= 0, "binary_predictions: B=%lld", (long long)B);
  if (B == 0) return MI_OK;
  MI_REQUIRE(logits && (logistic || probabilities || class_ids || unreduced_loss), "binary_predictions: null buffer");
  MI_REQUIRE(!unreduced_loss || labels, "binary_predictions: the per-example loss needs labels");
  binary_predictions_k<<<dim3((unsigned)mi::ceil_div(B, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
      logits, labels, B, logistic, probabilities, class_ids, unreduced_loss);
  MI_CHECK_LAUNCH("binary_predictions");
  return MI_OK;
}

extern "C" int32_t mi_layer_histogram(const float* x, int64_t n, const double* limits, int32_t n_limits, int64_t* counts,
                                      double* sums, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && n_limits > 0 && n_limits <= kHistMaxLimits, "layer_histogram: n=%lld n_limits=%d (<= %d)", (long long)n, n_limits,
             kHistMaxLimits);
  if (n == 0) return MI_OK;
  MI_REQUIRE(x && limits && counts && sums, "layer_histogram: null buffer");
  const int64_t blocks = std::min<int64_t>(mi::ceil_div(n, 256 * 8), 1024);
  layer_histogram_k<<<dim3((unsigned)blocks), dim3(256), 0, mi::as_stream(stream)>>>(
      x, n, limits, n_limits, reinterpret_cast<unsigned long long*>(counts), sums);
  MI_CHECK_LAUNCH("layer_histogram");
  return MI_OK;
}
