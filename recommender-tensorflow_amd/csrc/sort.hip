// Stable LSD radix sort of (row, entry index) pairs + duplicate compaction.
//
// Device analogue of TF's `unique` + `unsorted_segment_sum` (_deduplicate_indexed_slices) that
// runs before every sparse optimizer apply, reached from optimizer.minimize
// (trainers/model_utils.py:69-72 / head at trainers/deep_fm.py:119-125).
// Integer work, bit exact: equal rows keep their entry order, so the later segment sum adds
// duplicates in ascending example order like TF's CPU kernel.
//
// 8-bit digits; per pass: (1) per-tile digit histogram, (2) one-block exclusive scan over the
// bin-major [256][tiles] counts, (3) stable scatter: lanes find equal-digit peers in their wave
// with ballots, waves are chained through LDS counters, rounds through a running count.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 16;                 // keys per thread per tile
constexpr int kTile = kBlock * kItems;     // 4096
constexpr int kBins = 256;

__global__ __launch_bounds__(kBlock) void hist_k(const int32_t* __restrict__ keys, int64_t n, int shift,
                                                 int ntiles, int32_t* __restrict__ hist) {
  __shared__ unsigned int h[kBins];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(static_cast<uint32_t>(keys[i]) >> shift) & (kBins - 1)], 1u);
  }
  __syncthreads();
  hist[static_cast<int64_t>(threadIdx.x) * ntiles + blockIdx.x] = static_cast<int32_t>(h[threadIdx.x]);
}

// exclusive scan of `count` int32 values by ONE block of 1024 threads (count <= a few million)
__global__ __launch_bounds__(1024) void scan_k(int32_t* __restrict__ data, int64_t count,
                                               int32_t* __restrict__ total_out) {
  __shared__ int32_t part[1024];
  const int64_t per = (count + 1023) / 1024;
  const int64_t i0 = threadIdx.x * per, i1 = min(count, i0 + per);
  int32_t s = 0;
  for (int64_t i = i0; i < i1; ++i) s += data[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    int32_t v = (static_cast<int>(threadIdx.x) >= off) ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int32_t run = part[threadIdx.x] - s;
  for (int64_t i = i0; i < i1; ++i) {
    const int32_t v = data[i];
    data[i] = run;
    run += v;
  }
  if (total_out && threadIdx.x == 1023) total_out[0] = part[1023];
}

__global__ __launch_bounds__(kBlock) void scatter_k(const int32_t* __restrict__ keys_in,
                                                    const int32_t* __restrict__ vals_in, int64_t n,
                                                    int shift, int ntiles,
                                                    const int32_t* __restrict__ offs,
                                                    int32_t* __restrict__ keys_out,
                                                    int32_t* __restrict__ vals_out) {
  __shared__ int32_t running[kBins];       // global base + keys of earlier rounds, per digit
  __shared__ int32_t wcount[4][kBins];     // this round's per-wave digit counts
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  running[t] = offs[static_cast<int64_t>(t) * ntiles + blockIdx.x];
  wcount[0][t] = 0; wcount[1][t] = 0; wcount[2][t] = 0; wcount[3][t] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + t;
    const bool valid = i < n;
    const int32_t key = valid ? keys_in[i] : 0;
    const int32_t val = valid ? (vals_in ? vals_in[i] : static_cast<int32_t>(i)) : 0;
    const unsigned int d = (static_cast<uint32_t>(key) >> shift) & (kBins - 1);
    // peers = lanes of this wave holding a valid key with the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    if (valid && rank == 0) wcount[w][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int32_t dst = running[d] + rank;
      for (int ww = 0; ww < w; ++ww) dst += wcount[ww][d];
      keys_out[dst] = key;
      vals_out[dst] = val;
    }
    __syncthreads();
    running[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
    wcount[0][t] = 0; wcount[1][t] = 0; wcount[2][t] = 0; wcount[3][t] = 0;
    __syncthreads();
  }
}

// head flags per tile: heads[tile] = number of i in the tile with key[i] != key[i-1]
__global__ __launch_bounds__(kBlock) void head_count_k(const int32_t* __restrict__ keys, int64_t n,
                                                       int32_t* __restrict__ tile_heads) {
  __shared__ int32_t red[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + static_cast<int64_t>(threadIdx.x) * kItems;
  int c = 0;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r;
    if (i < n && (i == 0 || keys[i] != keys[i - 1])) ++c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_heads[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// each thread owns kItems consecutive sorted positions; writes uniq_rows / seg_start / sorted_entry
__global__ __launch_bounds__(kBlock) void compact_k(const int32_t* __restrict__ keys,
                                                    const int32_t* __restrict__ vals, int64_t n,
                                                    const int32_t* __restrict__ tile_base,
                                                    const int32_t* __restrict__ total,
                                                    int32_t* __restrict__ sorted_entry,
                                                    int32_t* __restrict__ uniq_rows,
                                                    int32_t* __restrict__ seg_start,
                                                    int32_t* __restrict__ num_uniq) {
  __shared__ int32_t wsum[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + static_cast<int64_t>(t) * kItems;
  int c = 0;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r;
    if (i < n && (i == 0 || keys[i] != keys[i - 1])) ++c;
  }
  int incl = c;                               // inclusive scan across the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int idx = tile_base[blockIdx.x] + incl - c;
  for (int ww = 0; ww < w; ++ww) idx += wsum[ww];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r;
    if (i < n) {
      sorted_entry[i] = vals[i];
      if (i == 0 || keys[i] != keys[i - 1]) {
        uniq_rows[idx] = keys[i];
        seg_start[idx] = static_cast<int32_t>(i);
        ++idx;
      }
    }
  }
  if (blockIdx.x == 0 && t == 0) {
    const int32_t U = total[0];
    num_uniq[0] = U;
    seg_start[U] = static_cast<int32_t>(n);
  }
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct Layout {
  int64_t ntiles, keysA, keysB, valsA, valsB, hist, heads, total, bytes;
};

Layout layout_for(int64_t n) {
  Layout L;
  L.ntiles = mi::ceil_div(n > 0 ? n : 1, kTile);
  int64_t o = 0;
  const int64_t nb = align_up(n * 4, 256);
  L.keysA = o; o += nb;
  L.keysB = o; o += nb;
  L.valsA = o; o += nb;
  L.valsB = o; o += nb;
  L.hist = o; o += align_up(L.ntiles * kBins * 4, 256);
  L.heads = o; o += align_up(L.ntiles * 4, 256);
  L.total = o; o += 256;
  L.bytes = o;
  return L;
}

}  // namespace

extern "C" {

size_t mi_sort_unique_workspace_bytes(int64_t n) { return static_cast<size_t>(layout_for(n).bytes); }

int32_t mi_sort_unique_rows(const int32_t* rows, int64_t n, int64_t num_rows_total,
                            int32_t* sorted_entry, int32_t* uniq_rows, int32_t* seg_start,
                            int32_t* num_uniq, void* workspace, size_t workspace_bytes,
                            mi_stream_t stream) {
  MI_REQUIRE(n > 0 && n < (int64_t)INT32_MAX - kTile, "sort_unique_rows: n=%lld", (long long)n);
  MI_REQUIRE(num_rows_total > 0 && num_rows_total <= (int64_t)INT32_MAX, "sort_unique_rows: num_rows_total=%lld",
             (long long)num_rows_total);
  MI_REQUIRE(rows && sorted_entry && uniq_rows && seg_start && num_uniq && workspace, "sort_unique_rows: null buffer");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "sort_unique_rows: workspace must be 256-byte aligned");
  const Layout L = layout_for(n);
  if (workspace_bytes < static_cast<size_t>(L.bytes)) {
    mi::set_error("sort_unique_rows: workspace %zu < %lld", workspace_bytes, (long long)L.bytes);
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int32_t* kbuf[2] = {reinterpret_cast<int32_t*>(ws + L.keysA), reinterpret_cast<int32_t*>(ws + L.keysB)};
  int32_t* vbuf[2] = {reinterpret_cast<int32_t*>(ws + L.valsA), reinterpret_cast<int32_t*>(ws + L.valsB)};
  int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
  int32_t* heads = reinterpret_cast<int32_t*>(ws + L.heads);
  int32_t* total = reinterpret_cast<int32_t*>(ws + L.total);
  const int ntiles = static_cast<int>(L.ntiles);

  int bits = 1;
  while (bits < 32 && (static_cast<int64_t>(1) << bits) < num_rows_total) ++bits;
  const int passes = (bits + 7) / 8;

  const int32_t* kin = rows;
  const int32_t* vin = nullptr;  // pass 0: value = position
  for (int p = 0; p < passes; ++p) {
    int32_t* kout = kbuf[p & 1];
    int32_t* vout = vbuf[p & 1];
    hist_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, n, 8 * p, ntiles, hist);
    MI_CHECK_LAUNCH("sort_unique_rows(hist)");
    scan_k<<<dim3(1), dim3(1024), 0, st>>>(hist, static_cast<int64_t>(ntiles) * kBins, nullptr);
    MI_CHECK_LAUNCH("sort_unique_rows(scan)");
    scatter_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, 8 * p, ntiles, hist, kout, vout);
    MI_CHECK_LAUNCH("sort_unique_rows(scatter)");
    kin = kout;
    vin = vout;
  }
  head_count_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, n, heads);
  MI_CHECK_LAUNCH("sort_unique_rows(heads)");
  scan_k<<<dim3(1), dim3(1024), 0, st>>>(heads, ntiles, total);
  MI_CHECK_LAUNCH("sort_unique_rows(scan heads)");
  compact_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, heads, total, sorted_entry, uniq_rows, seg_start, num_uniq);
  MI_CHECK_LAUNCH("sort_unique_rows(compact)");
  return MI_OK;
}

}  // extern "C"
