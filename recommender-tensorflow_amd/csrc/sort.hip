// Stable LSD radix sort of (row, entry index) pairs + duplicate compaction.
//
// Device analogue of TF's `unique` + `unsorted_segment_sum` (_deduplicate_indexed_slices) that
// runs before every sparse optimizer apply, reached from optimizer.minimize
// (trainers/model_utils.py:69-72 / head at trainers/deep_fm.py:119-125).
// Integer work, bit exact: equal rows keep their entry order, so the later segment sum adds
// duplicates in ascending example order like TF's CPU kernel.
//
// Digits of up to 9 bits (26 M rows = 25 bits -> 3 passes).  Per pass:
//   hist_k      per-tile digit histogram (LDS atomics) + per-digit totals (global integer atomics)
//   bin_scan_k  one workgroup per digit: base = sum of the lower digits' totals, then an
//               exclusive scan of that digit's per-tile counts (wave shuffles, carried in chunks)
//   scatter_k   stable scatter: lanes find their equal-digit peers in the wave with ballots; a wave owns
//               a contiguous quarter of the tile and keeps its own running slot per digit in LDS
// then head_count_k / compact_k mark the first entry of every distinct row (coalesced reads,
// ballot ranks) and emit uniq_rows / seg_start / sorted_entry.
// mi_sort_unique_fields: the same result for ids [B][F] of F fields whose rows are disjoint ranges — every field's
// B local ids sorted on their own (segments of the same kernels), 20 bits instead of 25 at config 3: 3 passes of 7
// bits, whose scattered writes come in 128-byte runs instead of 32-byte ones (193 -> 150 us).
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 16;                 // keys per thread per tile
constexpr int kTile = kBlock * kItems;     // 4096
constexpr int kMaxBits = 9;
constexpr int kMaxBins = 1 << kMaxBits;    // 512
constexpr int kMaxPasses = 4;
constexpr int kFusedMaxBits = 10;          // digit width limit of the fused per-field passes (pass_fused_k)
constexpr int kFusedMaxBins = 1 << kFusedMaxBits;
constexpr int kFusedMaxTps = 256;          // tiles per field up to which a field's tiles wait for each other inside one launch

// (our own kernel instead of hipMemsetAsync: a plain kernel node when the step is captured into a hipGraph — a
// LINEAR captured step with memset nodes faulted at replay on this ROCm, the same step with a forked side stream
// did not — and no runtime fill-kernel in between ours)
__global__ __launch_bounds__(256) void zero_i32_k(int32_t* __restrict__ p, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) p[i] = 0;
}
inline void zero_i32(int32_t* p, int64_t n, hipStream_t st) {
  // (MI_SORT_MEMSET=1: the round-2 form again, for tools/graph_memset_nodes.py only — it captures a step WITHOUT replaying it
  // and prints the memset nodes of the graph next to the live allocations)
  if (mi::env_int("MI_SORT_MEMSET", 0) != 0) {          // (the tools' build only: the shipped library cannot reach hipMemsetAsync)
    (void)hipMemsetAsync(p, 0, static_cast<size_t>(n) * 4, st);
    return;
  }
  const int64_t b = (n + 255) / 256;
  zero_i32_k<<<dim3(static_cast<unsigned>(b < 64 ? b : 64)), dim3(256), 0, st>>>(p, n);
}

// Segments (mi_sort_unique_fields: one per field, tps tiles each, sorted independently): tile = seg * tps + tis,
// hist[(seg * nbins + digit) * tps + tis], bin_total[seg * nbins + digit].  One segment: tps = ntiles.
__global__ __launch_bounds__(kBlock) void hist_k(const int32_t* __restrict__ keys, int64_t n, int shift,
                                                 int nbins, int tps, int32_t* __restrict__ hist,
                                                 int32_t* __restrict__ bin_total) {
  __shared__ unsigned int h[kMaxBins];
  for (int b = threadIdx.x; b < nbins; b += kBlock) h[b] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  const unsigned int mask = nbins - 1;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(static_cast<uint32_t>(keys[i]) >> shift) & mask], 1u);
  }
  __syncthreads();
  const int seg = blockIdx.x / tps, tis = blockIdx.x - seg * tps;
  for (int b = threadIdx.x; b < nbins; b += kBlock) {
    const unsigned int c = h[b];
    hist[(static_cast<int64_t>(seg) * nbins + b) * tps + tis] = static_cast<int32_t>(c);
    if (c) atomicAdd(&bin_total[seg * nbins + b], static_cast<int32_t>(c));
  }
}

// hist_k of the catch-up's staleness key, computed on the way (the key of mi_catchup_gap_keys: steps to replay,
// clamped to 62; 63 for the slots past num_uniq) and kept for the scatter
__global__ __launch_bounds__(kBlock) void gap_hist_k(const int32_t* __restrict__ uniq_rows, const int32_t* __restrict__ num_uniq,
                                                     const int32_t* __restrict__ last_step, int64_t n, int step_to, int st,
                                                     const mi_step_state_t* __restrict__ ss, int32_t* __restrict__ keys,
                                                     int ntiles, int32_t* __restrict__ hist, int32_t* __restrict__ bin_total) {
  __shared__ unsigned int h[64];
  if (ss) step_to = ss->step - 1;
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  const int64_t U = *num_uniq;
  int32_t key[kItems];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + threadIdx.x;
    key[r] = 63;
    if (i < U) {
      const int ls = last_step[static_cast<int64_t>(uniq_rows[i]) * st];
      key[r] = (ls > 0 && ls < step_to) ? min(step_to - ls, 62) : 0;
    }
  }
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + threadIdx.x;
    if (i < n) { keys[i] = key[r]; atomicAdd(&h[key[r]], 1u); }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned int c = h[threadIdx.x];
    hist[static_cast<int64_t>(threadIdx.x) * ntiles + blockIdx.x] = static_cast<int32_t>(c);
    if (c) atomicAdd(&bin_total[threadIdx.x], static_cast<int32_t>(c));
  }
}

// a wave's private LDS word, read / written where other lanes of the SAME wave write / read it in between
__device__ __forceinline__ int32_t lds_ld(int32_t* p) {
  return __hip_atomic_load((__attribute__((address_space(3))) int32_t*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void lds_st(int32_t* p, int32_t v) {
  __hip_atomic_store((__attribute__((address_space(3))) int32_t*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  return v;
}

// block (seg, b): hist[seg, b, 0..tps) -> global offsets (exclusive scan + base of digit b inside segment seg,
// which starts at seg * seg_len)
__global__ __launch_bounds__(kBlock) void bin_scan_k(int32_t* __restrict__ hist, int ntiles, int nbins,
                                                     const int32_t* __restrict__ bin_total, int64_t seg_len) {
  __shared__ int32_t red[4];
  __shared__ int32_t wsum[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int seg = blockIdx.x / nbins, b = blockIdx.x - seg * nbins;
  bin_total += seg * nbins;
  int part = 0;
  for (int j = t; j < b; j += kBlock) part += bin_total[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane == 0) red[w] = part;
  __syncthreads();
  int carry = red[0] + red[1] + red[2] + red[3] + static_cast<int>(seg * seg_len);
  int32_t* row = hist + (static_cast<int64_t>(seg) * nbins + b) * ntiles;
  for (int c0 = 0; c0 < ntiles; c0 += kBlock) {
    const int i = c0 + t;
    const int v = (i < ntiles) ? row[i] : 0;
    const int incl = wave_incl_scan(v, lane);
    __syncthreads();                     // previous chunk's wsum fully consumed
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int pre = carry;
    for (int ww = 0; ww < w; ++ww) pre += wsum[ww];
    if (i < ntiles) row[i] = pre + incl - v;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

// Wave w of a tile owns its keys [w * 1024, (w + 1) * 1024) (round r: 64 consecutive keys), so the tile order
// is wave order, then round order, then lane order.  Two block barriers in total: the waves first count
// their digits (private LDS histograms), every wave then derives its own running offsets
// (tile offset of the digit + the counts of the waves before it) and scatters its 16 rounds alone.
// (vals_in == NULL: the value of position i is i — or, with fm_B > 0, the entry b * fm_F + f of the field-major
// position i = f * fm_B + b.)
__global__ __launch_bounds__(kBlock) void scatter_k(const int32_t* __restrict__ keys_in,
                                                    const int32_t* __restrict__ vals_in, int64_t n,
                                                    int shift, int nbits, int tps,
                                                    const int32_t* __restrict__ offs,
                                                    int32_t* __restrict__ keys_out,
                                                    int32_t* __restrict__ vals_out, int64_t fm_B = 0, int fm_F = 0) {
  __shared__ int32_t running[4][kMaxBins];    // per wave: next output slot of each digit
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int nbins = 1 << nbits;
  const int seg = blockIdx.x / tps, tis = blockIdx.x - seg * tps;
  for (int b = t; b < 4 * kMaxBins; b += kBlock) (&running[0][0])[b] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + w * (64 * kItems);
  int32_t key[kItems], val[kItems];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0;
    val[r] = valid ? (vals_in ? vals_in[i] : (fm_B ? static_cast<int32_t>((i % fm_B) * fm_F + i / fm_B) : static_cast<int32_t>(i))) : 0;
    if (valid) atomicAdd(&running[w][(static_cast<uint32_t>(key[r]) >> shift) & (nbins - 1)], 1);
  }
  __syncthreads();
  // counts -> this wave's first slot per digit (each lane: digits lane, lane + 64, ...)
  int32_t first[kMaxBins / 64];
#pragma unroll
  for (int q = 0; q < kMaxBins / 64; ++q) {
    const int b = q * 64 + lane;
    int32_t f = 0;
    if (b < nbins) {
      f = offs[(static_cast<int64_t>(seg) * nbins + b) * tps + tis];
      for (int ww = 0; ww < w; ++ww) f += running[ww][b];
    }
    first[q] = f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kMaxBins / 64; ++q)
    if (q * 64 + lane < nbins) running[w][q * 64 + lane] = first[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // (the running slots through wavefront-scope atomics on the LDS array itself: a volatile POINTER to it loses the address
  // space and compiles to flat loads / stores with sc0 sc1)
  int32_t* run = running[w];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const bool valid = base + r * 64 + lane < n;
    const unsigned int d = (static_cast<uint32_t>(key[r]) >> shift) & (nbins - 1);
    // peers = lanes of this wave holding a valid key with the same digit
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    int32_t dst = 0;
    if (valid) dst = lds_ld(run + d) + rank;
    __builtin_amdgcn_wave_barrier();            // every peer has read the slot before its leader moves it
    if (valid) {
      if (rank == 0) lds_st(run + d, dst + __popcll(peers));
      keys_out[dst] = key[r];
      vals_out[dst] = val[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// (seg_len > 0: positions that start a segment — a field — start a row whatever the keys say)
__device__ __forceinline__ bool is_head(const int32_t* __restrict__ keys, int64_t i, int64_t seg_len) {
  return i == 0 || keys[i] != keys[i - 1] || (seg_len > 0 && i % seg_len == 0);
}

// heads[tile] = number of positions in the tile that start a new row
__global__ __launch_bounds__(kBlock) void head_count_k(const int32_t* __restrict__ keys, int64_t n,
                                                       int32_t* __restrict__ tile_heads, int64_t seg_len = 0) {
  __shared__ int32_t red[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  int c = 0;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + threadIdx.x;
    if (i < n && is_head(keys, i, seg_len)) ++c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_heads[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of `count` int32 values by ONE block (count = number of tiles: small).  256 threads, not 1,024: a
// 16-wave workgroup needs four free wave slots on every SIMD of ONE CU at the same moment, and beside the sparse apply or
// the resident catch-up workgroups it waited for that for 0.17 ms on average (0.5 ms at worst: rocprofv3 trace of the
// row-sharded step, profiles/r05_sharded_one_rank.md) — for 5 us of work.
constexpr int kScanSmallBlock = 256;
__global__ __launch_bounds__(kScanSmallBlock) void scan_small_k(int32_t* __restrict__ data, int64_t count,
                                                                int32_t* __restrict__ total_out) {
  constexpr int NW = kScanSmallBlock / 64;
  __shared__ int32_t wsum[NW];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  int carry = 0;
  for (int64_t c0 = 0; c0 < count; c0 += kScanSmallBlock) {
    const int64_t i = c0 + t;
    const int v = (i < count) ? data[i] : 0;
    const int incl = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int pre = carry, tot = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      if (ww < w) pre += wsum[ww];
      tot += wsum[ww];
    }
    if (i < count) data[i] = pre + incl - v;
    carry += tot;
  }
  if (total_out && t == 0) total_out[0] = carry;
}

__global__ __launch_bounds__(kBlock) void compact_k(const int32_t* __restrict__ keys,
                                                    const int32_t* __restrict__ vals, int64_t n,
                                                    const int32_t* __restrict__ tile_base,
                                                    const int32_t* __restrict__ total,
                                                    int32_t* __restrict__ sorted_entry,
                                                    int32_t* __restrict__ uniq_rows,
                                                    int32_t* __restrict__ seg_start,
                                                    int32_t* __restrict__ num_uniq, int64_t seg_len = 0,
                                                    const int64_t* __restrict__ field_off = nullptr,
                                                    int32_t* __restrict__ slot_of_entry = nullptr) {
  __shared__ int32_t wc[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  int run = tile_base[blockIdx.x];
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + t;
    const bool valid = i < n;
    int32_t key = 0;
    bool head = false;
    if (valid) {
      key = keys[i];
      head = is_head(keys, i, seg_len);
      sorted_entry[i] = vals[i];
    }
    const unsigned long long hb = __ballot(head);
    __syncthreads();                       // previous round's wc consumed
    if (lane == 0) wc[w] = __popcll(hb);
    __syncthreads();
    int idx = run + __popcll(hb & lt_mask);
    for (int ww = 0; ww < w; ++ww) idx += wc[ww];
    if (head) {
      uniq_rows[idx] = seg_len > 0 ? static_cast<int32_t>(field_off[i / seg_len] + key) : key;   // (local id -> global row)
      seg_start[idx] = static_cast<int32_t>(i);
    }
    // (the segment a position belongs to: the heads up to and including its own, minus one — what mi_segment_slots finds by
    // binary search, 112 us at 1.7 M entries, for the price of one scattered 4-byte store here)
    if (slot_of_entry && valid) slot_of_entry[vals[i]] = head ? idx : idx - 1;
    run += wc[0] + wc[1] + wc[2] + wc[3];
  }
  if (blockIdx.x == 0 && t == 0) {
    const int32_t U = total[0];
    num_uniq[0] = U;
    seg_start[U] = static_cast<int32_t>(n);
  }
}

// ids [B][F] -> out [F][B] (a field's ids contiguous): 64 examples per block through LDS, both sides coalesced
constexpr int kFieldsMax = 64;
__global__ __launch_bounds__(kBlock) void ids_field_major_k(const int32_t* __restrict__ ids, int64_t B, int F,
                                                            int32_t* __restrict__ out) {
  __shared__ int32_t tile[64][kFieldsMax + 1];
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int nb = static_cast<int>(min(static_cast<int64_t>(64), B - b0));
  for (int i = threadIdx.x; i < nb * F; i += kBlock) tile[i / F][i % F] = ids[b0 * F + i];
  __syncthreads();
  for (int i = threadIdx.x; i < F * 64; i += kBlock) {
    const int f = i >> 6, bl = i & 63;
    if (bl < nb) out[static_cast<int64_t>(f) * B + b0 + bl] = tile[bl][f];
  }
}


// ---- One launch per radix pass, one for the compaction (mi_sort_unique_fields) --------------------------
// hist_k + bin_scan_k + scatter_k of a pass as ONE kernel: a tile counts its digits, PUBLISHES the counts, waits until the
// other tiles of its field have published theirs (tps <= kFusedMaxTps workgroups, consecutive block ids: dispatched together),
// derives its own offsets from them and scatters.  14 dependent launches of ~12 us each (most of them a few workgroups or
// ONE) become 4-5.  Cross-workgroup traffic goes through agent-scope atomics (the L2s of the 8 XCDs are not coherent for
// plain loads inside a launch): relaxed stores of the counts, a release increment of the field's arrival counter, an
// acquire spin on it, relaxed loads of the counts.  The spin is bounded (every wave reaches its exit: a tile that never
// sees its field complete raises *err and goes on with what it has — garbage that tests catch — instead of hanging the GPU).
__device__ __forceinline__ void st_agent(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr int kSpinMax = 1 << 22;          // x s_sleep 8 (~0.25 us): about a second

// pass 0 reads ids [B][F] itself (key = the local id of (b, seg), value = the entry b * F + seg): no field-major copy
template <int MAXBINS>
__global__ __launch_bounds__(kBlock) void pass_fused_k(const int32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                       const int32_t* __restrict__ ids, int64_t B, int F, int shift, int nbits,
                                                       int tps, int32_t* __restrict__ thist, int32_t* __restrict__ ready,
                                                       int32_t* __restrict__ err, int32_t* __restrict__ keys_out,
                                                       int32_t* __restrict__ vals_out, int nap) {
  __shared__ int32_t running[4][MAXBINS];     // per wave: digit counts, then the next output slot of each digit
  __shared__ int32_t offsT[MAXBINS];          // this tile's first slot per digit
  __shared__ int32_t totS[MAXBINS];           // the field's total per digit
  __shared__ int32_t wsum[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int nbins = 1 << nbits;
  const int seg = blockIdx.x / tps, tis = blockIdx.x - seg * tps;
  for (int b = t; b < 4 * MAXBINS; b += kBlock) (&running[0][0])[b] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + w * (64 * kItems);
  int32_t key[kItems], val[kItems];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * 64 + lane;         // (B is a multiple of kTile: every position exists)
    if (ids) {
      const int64_t e = (i - static_cast<int64_t>(seg) * B) * F + seg;
      key[r] = ids[e];
      val[r] = static_cast<int32_t>(e);
    } else {
      key[r] = keys_in[i];
      val[r] = vals_in[i];
    }
    atomicAdd(&running[w][(static_cast<uint32_t>(key[r]) >> shift) & (nbins - 1)], 1);
  }
  __syncthreads();
  for (int b = t; b < nbins; b += kBlock)
    st_agent(thist + static_cast<int64_t>(blockIdx.x) * nbins + b, running[0][b] + running[1][b] + running[2][b] + running[3][b]);
  // (the counts went out as write-through stores: once they are acknowledged — vmcnt 0 in every wave, then the barrier —
  // a RELAXED increment publishes them.  A release increment would write back this XCD's whole L2, an acquire load in the
  // spin invalidate it on every turn: 160-190 us per sort instead of 150 for the 14 launches this replaces.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    __hip_atomic_fetch_add(ready + seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (ld_agent(ready + seg) < tps) {
      if (nap >= 2) __builtin_amdgcn_s_sleep(100); else if (nap == 1) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(4);
      if (++spins > kSpinMax) { atomicExch(err, 1); break; }
    }
  }
  __syncthreads();
  // per digit: the field's total and the part of it that lies in the tiles before this one (L2-bypassing loads, kept in
  // flight eight at a time: one after the other they cost a memory latency each)
  for (int b = t; b < nbins; b += kBlock) {
    const int32_t* col = thist + static_cast<int64_t>(seg) * tps * nbins + b;
    int total = 0, before = 0;
    for (int t0 = 0; t0 < tps; t0 += 8) {
      int v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (t0 + j < tps) ? ld_agent(col + static_cast<int64_t>(t0 + j) * nbins) : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        total += v[j];
        before += (t0 + j < tis) ? v[j] : 0;
      }
    }
    totS[b] = total;
    offsT[b] = before;
  }
  __syncthreads();
  // exclusive scan of the totals over the digits: thread t owns `per` consecutive digits
  {
    const int per = nbins > kBlock ? nbins / kBlock : 1;
    int own = 0;
    if (t * per < nbins)
      for (int j = 0; j < per; ++j) own += totS[t * per + j];
    const int incl = wave_incl_scan(own, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int pre = incl - own + static_cast<int>(static_cast<int64_t>(seg) * B);
    for (int ww = 0; ww < w; ++ww) pre += wsum[ww];
    if (t * per < nbins)
      for (int j = 0; j < per; ++j) {
        const int b = t * per + j;
        const int c = totS[b];
        offsT[b] += pre;
        pre += c;
      }
  }
  __syncthreads();
  int32_t first[MAXBINS / 64];
#pragma unroll
  for (int q = 0; q < MAXBINS / 64; ++q) {
    const int b = q * 64 + lane;
    int32_t f = 0;
    if (b < nbins) {
      f = offsT[b];
      for (int ww = 0; ww < w; ++ww) f += running[ww][b];
    }
    first[q] = f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXBINS / 64; ++q)
    if (q * 64 + lane < nbins) running[w][q * 64 + lane] = first[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int32_t* run = running[w];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const unsigned int d = (static_cast<uint32_t>(key[r]) >> shift) & (nbins - 1);
    unsigned long long peers = ~0ull;
    for (int b = 0; b < nbits; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const int32_t dst = lds_ld(run + d) + rank;
    __builtin_amdgcn_wave_barrier();            // every peer has read the slot before its leader moves it
    if (rank == 0) lds_st(run + d, dst + __popcll(peers));
    keys_out[dst] = key[r];
    vals_out[dst] = val[r];
    __builtin_amdgcn_wave_barrier();
  }
}

// head_count_k + scan_small_k + compact_k as one launch: a tile publishes its number of row heads (+ 1: 0 = not there yet)
// and waits for the tiles BEFORE it only — lower block ids, dispatched earlier — to know where its rows start.
__global__ __launch_bounds__(kBlock) void compact_fused_k(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t n,
                                                          int32_t* __restrict__ hcf, int32_t* __restrict__ err,
                                                          int32_t* __restrict__ sorted_entry, int32_t* __restrict__ uniq_rows,
                                                          int32_t* __restrict__ seg_start, int32_t* __restrict__ num_uniq,
                                                          int64_t seg_len, const int64_t* __restrict__ field_off) {
  __shared__ int32_t red[4];
  __shared__ int32_t wc[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  int c = 0;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + t;
    if (i < n && is_head(keys, i, seg_len)) ++c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if (lane == 0) red[w] = c;
  __syncthreads();
  const int mine = red[0] + red[1] + red[2] + red[3];
  if (t == 0) st_agent(hcf + blockIdx.x, mine + 1);
  int part = 0;
  for (int j = t; j < static_cast<int>(blockIdx.x); j += kBlock) {
    int v, spins = 0;
    while ((v = ld_agent(hcf + j)) == 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > kSpinMax) { atomicExch(err, 1); v = 1; break; }
    }
    part += v - 1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  __syncthreads();                         // red[] read by everyone
  if (lane == 0) red[w] = part;
  __syncthreads();
  int run = red[0] + red[1] + red[2] + red[3];
  for (int r = 0; r < kItems; ++r) {
    const int64_t i = base + r * kBlock + t;
    const bool valid = i < n;
    int32_t key = 0;
    bool head = false;
    if (valid) {
      key = keys[i];
      head = is_head(keys, i, seg_len);
      sorted_entry[i] = vals[i];
    }
    const unsigned long long hb = __ballot(head);
    __syncthreads();                       // previous round's wc consumed
    if (lane == 0) wc[w] = __popcll(hb);
    __syncthreads();
    if (head) {
      int idx = run + __popcll(hb & lt_mask);
      for (int ww = 0; ww < w; ++ww) idx += wc[ww];
      uniq_rows[idx] = seg_len > 0 ? static_cast<int32_t>(field_off[i / seg_len] + key) : key;   // (local id -> global row)
      seg_start[idx] = static_cast<int32_t>(i);
    }
    run += wc[0] + wc[1] + wc[2] + wc[3];
  }
  if (blockIdx.x == gridDim.x - 1 && t == 0) {
    num_uniq[0] = run;
    seg_start[run] = static_cast<int32_t>(n);
  }
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct Layout {
  int64_t ntiles, keysA, keysB, valsA, valsB, hist, heads, bin_total, total, sync, sync_words, bytes;
};

Layout layout_for(int64_t n, int nseg = 1) {
  Layout L;
  L.ntiles = mi::ceil_div(n > 0 ? n : 1, kTile);
  int64_t o = 0;
  const int64_t nb = align_up(n * 4, 256);
  L.keysA = o; o += nb;
  L.keysB = o; o += nb;
  L.valsA = o; o += nb;
  L.valsB = o; o += nb;
  L.hist = o; o += align_up(L.ntiles * (nseg > 1 ? kFusedMaxBins : kMaxBins) * 4, 256);
  L.heads = o; o += align_up(L.ntiles * 4, 256);
  L.bin_total = o; o += static_cast<int64_t>(kMaxPasses) * (nseg > 1 ? nseg : 1) * kMaxBins * 4;
  L.total = o; o += 256;
  // the fused passes' sync words (mi_sort_unique_fields): [pass][field] arrival counters, one published head count per
  // tile, one error word — zeroed by the entry's first launch
  L.sync_words = static_cast<int64_t>(kMaxPasses) * (nseg > 1 ? nseg : 1) + L.ntiles + 1;
  L.sync = o; o += align_up(L.sync_words * 4, 256);
  L.bytes = o;
  return L;
}

}  // namespace

// ---- n <= kSmallN: the whole sort + unique in ONE workgroup (the launch-bound small-batch step: B = 32 x 26 fields
// = 832 keys would otherwise take a dozen launches).
namespace {
constexpr int kSmallN = 1024;
__global__ __launch_bounds__(kBlock) void sort_small_k(const int32_t* __restrict__ keys, int n, int32_t* __restrict__ sorted_entry,
                                                       int32_t* __restrict__ uniq_rows, int32_t* __restrict__ seg_start,
                                                       int32_t* __restrict__ num_uniq, int32_t* __restrict__ slot_of_entry = nullptr) {
  // bitonic sort of (key << 10 | index) in LDS: the index makes equal keys keep their order (stable), padding sorts
  // last.  (The first version ranked every key against every other: 87 us of the 280-us B = 32 step.)
  __shared__ unsigned long long a[kSmallN];
  __shared__ int32_t sk[kSmallN], head[kSmallN];
  __shared__ int32_t wsum[4];
  const int t = threadIdx.x;
  int np2 = 64;
  while (np2 < n) np2 <<= 1;
  for (int i = t; i < np2; i += kBlock)
    a[i] = i < n ? (static_cast<unsigned long long>(static_cast<uint32_t>(keys[i])) << 10) | static_cast<unsigned>(i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < np2; i += kBlock) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long x = a[i], y = a[p];
          if ((x > y) == ((i & k) == 0)) { a[i] = y; a[p] = x; }
        }
      }
      __syncthreads();
    }
  for (int i = t; i < n; i += kBlock) {
    sk[i] = static_cast<int32_t>(a[i] >> 10);
    sorted_entry[i] = static_cast<int32_t>(a[i] & 1023u);
  }
  if (!uniq_rows) return;
  __syncthreads();
  // head flags -> exclusive scan (4 passes of 256) -> compaction
  int carry = 0;
  for (int c0 = 0; c0 < n; c0 += kBlock) {
    const int i = c0 + t;
    const int v = (i < n && (i == 0 || sk[i] != sk[i - 1])) ? 1 : 0;
    const int incl = wave_incl_scan(v, t & 63);
    __syncthreads();
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < (t >> 6); ++w) pre += wsum[w];
    if (i < n) head[i] = v ? pre + incl - 1 : -1;
    if (slot_of_entry && i < n) slot_of_entry[a[i] & 1023u] = pre + incl - 1;      // (heads up to and including i, minus one)
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  __syncthreads();
  for (int i = t; i < n; i += kBlock)
    if (head[i] >= 0) { uniq_rows[head[i]] = sk[i]; seg_start[head[i]] = i; }
  if (t == 0) { seg_start[carry] = n; *num_uniq = carry; }
}
}  // namespace

extern "C" {

size_t mi_sort_unique_workspace_bytes(int64_t n) { return static_cast<size_t>(layout_for(n).bytes); }

static int32_t sort_unique_rows_impl(const int32_t* rows, int64_t n, int64_t num_rows_total,
                                     int32_t* sorted_entry, int32_t* uniq_rows, int32_t* seg_start,
                                     int32_t* num_uniq, int32_t* slot_of_entry, void* workspace, size_t workspace_bytes,
                                     mi_stream_t stream) {
  MI_REQUIRE(n > 0 && n < (int64_t)INT32_MAX - kTile, "sort_unique_rows: n=%lld", (long long)n);
  MI_REQUIRE(num_rows_total > 0 && num_rows_total <= (int64_t)INT32_MAX, "sort_unique_rows: num_rows_total=%lld",
             (long long)num_rows_total);
  MI_REQUIRE(rows && sorted_entry && workspace, "sort_unique_rows: null buffer");
  const bool perm_only = uniq_rows == nullptr;       // just the stable sort permutation, no unique / segment pass
  MI_REQUIRE(perm_only || (seg_start && num_uniq), "sort_unique_rows: uniq_rows needs seg_start and num_uniq");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "sort_unique_rows: workspace must be 256-byte aligned");
  const Layout L = layout_for(n);
  if (workspace_bytes < static_cast<size_t>(L.bytes)) {
    mi::set_error("sort_unique_rows: workspace %zu < %lld", workspace_bytes, (long long)L.bytes);
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  if (n <= kSmallN) {
    sort_small_k<<<dim3(1), dim3(kBlock), 0, st>>>(rows, static_cast<int>(n), sorted_entry, uniq_rows, seg_start, num_uniq, slot_of_entry);
    MI_CHECK_LAUNCH("sort_unique_rows(small)");
    return MI_OK;
  }
  char* ws = static_cast<char*>(workspace);
  int32_t* kbuf[2] = {reinterpret_cast<int32_t*>(ws + L.keysA), reinterpret_cast<int32_t*>(ws + L.keysB)};
  int32_t* vbuf[2] = {reinterpret_cast<int32_t*>(ws + L.valsA), reinterpret_cast<int32_t*>(ws + L.valsB)};
  int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
  int32_t* heads = reinterpret_cast<int32_t*>(ws + L.heads);
  int32_t* bin_total = reinterpret_cast<int32_t*>(ws + L.bin_total);
  int32_t* total = reinterpret_cast<int32_t*>(ws + L.total);
  const int ntiles = static_cast<int>(L.ntiles);

  int bits = 1;
  while (bits < 31 && (static_cast<int64_t>(1) << bits) < num_rows_total) ++bits;
  const int passes = (bits + kMaxBits - 1) / kMaxBits;
  const int nbits = (bits + passes - 1) / passes;   // digit width, <= 9
  const int nbins = 1 << nbits;

  zero_i32(bin_total, kMaxPasses * kMaxBins, st);
  MI_CHECK_LAUNCH("sort_unique_rows(zero)");
  const int32_t* kin = rows;
  const int32_t* vin = nullptr;  // pass 0: value = position
  for (int p = 0; p < passes; ++p) {
    int32_t* kout = kbuf[p & 1];
    int32_t* vout = (perm_only && p == passes - 1) ? sorted_entry : vbuf[p & 1];
    int32_t* bt = bin_total + p * kMaxBins;
    hist_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, n, nbits * p, nbins, ntiles, hist, bt);
    MI_CHECK_LAUNCH("sort_unique_rows(hist)");
    bin_scan_k<<<dim3(nbins), dim3(kBlock), 0, st>>>(hist, ntiles, nbins, bt, 0);
    MI_CHECK_LAUNCH("sort_unique_rows(scan)");
    scatter_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, nbits * p, nbits, ntiles, hist, kout, vout);
    MI_CHECK_LAUNCH("sort_unique_rows(scatter)");
    kin = kout;
    vin = vout;
  }
  if (perm_only) return MI_OK;
  head_count_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, n, heads);
  MI_CHECK_LAUNCH("sort_unique_rows(heads)");
  scan_small_k<<<dim3(1), dim3(kScanSmallBlock), 0, st>>>(heads, ntiles, total);
  MI_CHECK_LAUNCH("sort_unique_rows(scan heads)");
  compact_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, heads, total, sorted_entry, uniq_rows, seg_start, num_uniq, 0, nullptr,
                                                   slot_of_entry);
  MI_CHECK_LAUNCH("sort_unique_rows(compact)");
  return MI_OK;
}

int32_t mi_sort_unique_rows(const int32_t* rows, int64_t n, int64_t num_rows_total, int32_t* sorted_entry, int32_t* uniq_rows,
                            int32_t* seg_start, int32_t* num_uniq, void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  return sort_unique_rows_impl(rows, n, num_rows_total, sorted_entry, uniq_rows, seg_start, num_uniq, nullptr, workspace, workspace_bytes, stream);
}

int32_t mi_sort_unique_rows_slots(const int32_t* rows, int64_t n, int64_t num_rows_total, int32_t* sorted_entry, int32_t* uniq_rows,
                                  int32_t* seg_start, int32_t* num_uniq, int32_t* slot_of_entry, void* workspace,
                                  size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(slot_of_entry && uniq_rows, "sort_unique_rows_slots: null buffer");
  return sort_unique_rows_impl(rows, n, num_rows_total, sorted_entry, uniq_rows, seg_start, num_uniq, slot_of_entry, workspace, workspace_bytes, stream);
}


int32_t mi_catchup_rows_by_gap(const int32_t* uniq_rows, const int32_t* num_uniq, const int32_t* last_step, int64_t n_max,
                               int32_t step_to, int32_t lin_stride, int32_t* rows_out, void* workspace,
                               size_t workspace_bytes, mi_stream_t stream) {
  MI_REQUIRE(n_max >= 0 && n_max < (int64_t)INT32_MAX - kTile && step_to >= 0 && lin_stride >= 1,
             "catchup_rows_by_gap: n_max=%lld", (long long)n_max);
  if (n_max == 0) return MI_OK;
  MI_REQUIRE(uniq_rows && num_uniq && last_step && rows_out && workspace, "catchup_rows_by_gap: null buffer");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "catchup_rows_by_gap: workspace must be 256-byte aligned");
  const Layout L = layout_for(n_max);
  if (workspace_bytes < static_cast<size_t>(L.bytes)) {
    mi::set_error("catchup_rows_by_gap: workspace %zu < %lld", workspace_bytes, (long long)L.bytes);
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int32_t* keys = reinterpret_cast<int32_t*>(ws + L.keysA);
  int32_t* keys_sorted = reinterpret_cast<int32_t*>(ws + L.keysB);
  int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
  int32_t* bin_total = reinterpret_cast<int32_t*>(ws + L.bin_total);
  const int ntiles = static_cast<int>(L.ntiles);
  zero_i32(bin_total, 64, st);
  MI_CHECK_LAUNCH("catchup_rows_by_gap(zero)");
  gap_hist_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(uniq_rows, num_uniq, last_step, n_max, step_to, lin_stride, mi::step_state(),
                                                    keys, ntiles, hist, bin_total);
  MI_CHECK_LAUNCH("catchup_rows_by_gap(keys + hist)");
  bin_scan_k<<<dim3(64), dim3(kBlock), 0, st>>>(hist, ntiles, 64, bin_total, 0);
  MI_CHECK_LAUNCH("catchup_rows_by_gap(scan)");
  scatter_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(keys, uniq_rows, n_max, 0, 6, ntiles, hist, keys_sorted, rows_out);
  MI_CHECK_LAUNCH("catchup_rows_by_gap(scatter)");
  return MI_OK;
}


size_t mi_sort_unique_fields_workspace_bytes(int64_t B, int32_t F) {
  return static_cast<size_t>(layout_for(B * (F > 0 ? F : 1), F).bytes);
}

int32_t mi_sort_unique_fields(const int32_t* ids, const int64_t* field_off, int64_t B, int32_t F, int64_t max_vocab,
                              int32_t* sorted_entry, int32_t* uniq_rows, int32_t* seg_start, int32_t* num_uniq,
                              void* workspace, size_t workspace_bytes, int32_t beside, mi_stream_t stream) {
  const int64_t n = B * F;
  MI_REQUIRE(B > 0 && F > 0 && F <= kFieldsMax && B % kTile == 0 && n < (int64_t)INT32_MAX - kTile,
             "sort_unique_fields: B=%lld (a multiple of %d) F=%d (<= %d)", (long long)B, kTile, F, kFieldsMax);
  MI_REQUIRE(max_vocab > 0 && max_vocab <= (int64_t)1 << 30, "sort_unique_fields: max_vocab=%lld", (long long)max_vocab);
  MI_REQUIRE(ids && field_off && sorted_entry && uniq_rows && seg_start && num_uniq && workspace, "sort_unique_fields: null buffer");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "sort_unique_fields: workspace must be 256-byte aligned");
  const Layout L = layout_for(n, F);
  if (workspace_bytes < static_cast<size_t>(L.bytes)) {
    mi::set_error("sort_unique_fields: workspace %zu < %lld", workspace_bytes, (long long)L.bytes);
    return MI_ERR_WORKSPACE;
  }
  hipStream_t st = mi::as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int32_t* kbuf[2] = {reinterpret_cast<int32_t*>(ws + L.keysA), reinterpret_cast<int32_t*>(ws + L.keysB)};
  int32_t* vbuf[2] = {reinterpret_cast<int32_t*>(ws + L.valsA), reinterpret_cast<int32_t*>(ws + L.valsB)};
  int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
  int32_t* heads = reinterpret_cast<int32_t*>(ws + L.heads);
  int32_t* bin_total = reinterpret_cast<int32_t*>(ws + L.bin_total);
  int32_t* total = reinterpret_cast<int32_t*>(ws + L.total);
  const int ntiles = static_cast<int>(L.ntiles), tps = static_cast<int>(B / kTile);

  int bits = 1;
  while (bits < 31 && (static_cast<int64_t>(1) << bits) < max_vocab) ++bits;
  const int passes = (bits + kMaxBits - 1) / kMaxBits;
  const int nbits = (bits + passes - 1) / passes;   // digit width, <= 9
  const int nbins = 1 << nbits;

  if (tps <= kFusedMaxTps && !beside && mi::env_int("MI_SORT_FUSED", 1) != 0) {
    // one launch per pass + one for the compaction (pass_fused_k / compact_fused_k above).  Digits of 7 bits: 20-bit ids = 3
    // passes, 130 us at config 3 against 149 for 2 passes of 10 bits (1,024 bins per 4,096-key tile scatter 16-byte runs)
    const int fbits = mi::env_int("MI_SORT_BITS", 7) < kFusedMaxBits ? mi::env_int("MI_SORT_BITS", 7) : kFusedMaxBits;
    const int fpasses = (bits + fbits - 1) / fbits;
    const int fnb = (bits + fpasses - 1) / fpasses;
    MI_REQUIRE(fpasses <= kMaxPasses, "sort_unique_fields: %d passes", fpasses);
    int32_t* sync = reinterpret_cast<int32_t*>(ws + L.sync);
    int32_t* ready = sync;                                   // [pass][field]
    int32_t* hcf = sync + static_cast<int64_t>(kMaxPasses) * F;  // [tile]
    int32_t* err = hcf + ntiles;
    zero_i32(sync, L.sync_words, st);
    MI_CHECK_LAUNCH("sort_unique_fields(zero)");
    const int32_t* kin = nullptr;
    const int32_t* vin = nullptr;
    const int nap = mi::env_int("MI_SORT_NAP", 1);
    for (int p = 0; p < fpasses; ++p) {
      int32_t* kout = kbuf[p & 1];
      int32_t* vout = vbuf[p & 1];
      if (fnb > 9)
        pass_fused_k<kFusedMaxBins><<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, p == 0 ? ids : nullptr, B, F, fnb * p, fnb, tps, hist,
                                                                          ready + static_cast<int64_t>(p) * F, err, kout, vout, nap);
      else
        pass_fused_k<kMaxBins><<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, p == 0 ? ids : nullptr, B, F, fnb * p, fnb, tps, hist,
                                                                      ready + static_cast<int64_t>(p) * F, err, kout, vout, nap);
      MI_CHECK_LAUNCH("sort_unique_fields(pass)");
      kin = kout;
      vin = vout;
    }
    compact_fused_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, hcf, err, sorted_entry, uniq_rows, seg_start, num_uniq, B, field_off);
    MI_CHECK_LAUNCH("sort_unique_fields(compact)");
    return MI_OK;
  }
  zero_i32(bin_total, static_cast<int64_t>(kMaxPasses) * F * kMaxBins, st);
  MI_CHECK_LAUNCH("sort_unique_fields(zero)");
  // the ids field by field (keys = local ids); pass 0 reads them from the buffer pass 1 will overwrite
  ids_field_major_k<<<dim3((unsigned)mi::ceil_div(B, 64)), dim3(kBlock), 0, st>>>(ids, B, F, kbuf[1]);
  MI_CHECK_LAUNCH("sort_unique_fields(transpose)");
  const int32_t* kin = kbuf[1];
  const int32_t* vin = nullptr;  // pass 0: value = entry of the field-major position
  for (int p = 0; p < passes; ++p) {
    int32_t* kout = kbuf[p & 1];
    int32_t* vout = vbuf[p & 1];
    int32_t* bt = bin_total + static_cast<int64_t>(p) * F * kMaxBins;
    hist_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, n, nbits * p, nbins, tps, hist, bt);
    MI_CHECK_LAUNCH("sort_unique_fields(hist)");
    bin_scan_k<<<dim3(F * nbins), dim3(kBlock), 0, st>>>(hist, tps, nbins, bt, B);
    MI_CHECK_LAUNCH("sort_unique_fields(scan)");
    scatter_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, nbits * p, nbits, tps, hist, kout, vout, p == 0 ? B : 0, F);
    MI_CHECK_LAUNCH("sort_unique_fields(scatter)");
    kin = kout;
    vin = vout;
  }
  head_count_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, n, heads, B);
  MI_CHECK_LAUNCH("sort_unique_fields(heads)");
  scan_small_k<<<dim3(1), dim3(kScanSmallBlock), 0, st>>>(heads, ntiles, total);
  MI_CHECK_LAUNCH("sort_unique_fields(scan heads)");
  compact_k<<<dim3(ntiles), dim3(kBlock), 0, st>>>(kin, vin, n, heads, total, sorted_entry, uniq_rows, seg_start, num_uniq, B, field_off);
  MI_CHECK_LAUNCH("sort_unique_fields(compact)");
  return MI_OK;
}

}  // extern "C"

// ---- routing helpers of the row-sharded (multi-GPU) path ------------------------------------
namespace {

// request key of entry i for row r: ((chunk * world + owner) * rows_per_rank + local row); sorting by it
// (mi_sort_unique_rows) orders the entries by (chunk, owner, row) and its unique keys are the DISTINCT rows
// a chunk needs from an owner: a row asked for by many entries of a batch crosses the link once.
// self_rank >= 0: the owners are numbered with the asking rank itself LAST (owner o -> o for o < self, o - 1 for
// o > self, world - 1 for self): a rank's requests to itself then sit at the end of every chunk's run — outside the
// part of the buffers the all-to-all ships, which skips the self piece (its split size is 0).
__global__ __launch_bounds__(kBlock) void shard_keys_k(const int32_t* __restrict__ rows, int64_t n, int world,
                                                       int64_t entries_per_chunk, int64_t rows_per_rank, int self_rank,
                                                       int32_t* __restrict__ keys) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rows[i];
  const int64_t chunk = entries_per_chunk > 0 ? i / entries_per_chunk : 0;
  int o = r % world;
  if (self_rank >= 0) o = o == self_rank ? world - 1 : (o > self_rank ? o - 1 : o);
  keys[i] = static_cast<int32_t>((chunk * world + o) * rows_per_rank + r / world);
}

// per distinct request u: the owner-local row to ask for, and the request counts per (chunk, owner) group.
// The keys are sorted, so a group is one contiguous run: its count is (index of the first key past it) - (index of its
// first key), and only the two threads at a run's ends touch counts[] — O(groups) atomics.  (One atomicAdd per request on
// the group's counter, as this kernel first did, serialises 1.7 M atomics on C x world addresses: 19 ms per step with
// one rank and 4 chunks, measured with bench.py --force-shard.)
__global__ __launch_bounds__(kBlock) void route_requests_k(const int32_t* __restrict__ uniq_keys,
                                                           const int32_t* __restrict__ num_uniq, int64_t rows_per_rank,
                                                           int32_t* __restrict__ send_rows, int32_t* __restrict__ counts) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t U = *num_uniq;
  if (u >= U) return;
  const int64_t key = uniq_keys[u];
  const int64_t grp = key / rows_per_rank;
  send_rows[u] = static_cast<int32_t>(key - grp * rows_per_rank);
  if (u + 1 == U) {
    atomicAdd(counts + grp, static_cast<int32_t>(U));                   // the last run ends at U
  } else {
    const int64_t gn = static_cast<int64_t>(uniq_keys[u + 1]) / rows_per_rank;
    if (gn != grp) {                                                    // run of grp ends, run of gn starts, at u + 1
      atomicAdd(counts + grp, static_cast<int32_t>(u + 1));
      atomicAdd(counts + gn, -static_cast<int32_t>(u + 1));
    }
  }
}

// slot[e] = index of the distinct request that entry e belongs to (binary search of the sorted position in
// seg_start)
__global__ __launch_bounds__(kBlock) void segment_slots_k(const int32_t* __restrict__ seg_start,
                                                          const int32_t* __restrict__ sorted_entry,
                                                          const int32_t* __restrict__ num_uniq, int64_t n,
                                                          int32_t* __restrict__ slot) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= n) return;
  int lo = 0, hi = *num_uniq;                      // largest u with seg_start[u] <= k
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_start[mid] <= k) lo = mid; else hi = mid;
  }
  slot[sorted_entry[k]] = lo;
}

__global__ __launch_bounds__(kBlock) void gather_u32_k(const uint32_t* __restrict__ src,
                                                       const int32_t* __restrict__ idx, int64_t n,
                                                       uint32_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

}  // namespace

extern "C" {

int32_t mi_shard_keys(const int32_t* rows, int64_t n, int32_t world, int64_t entries_per_chunk, int64_t rows_per_rank,
                      int32_t self_rank, int32_t* keys, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && world > 0 && entries_per_chunk >= 0 && rows_per_rank > 0 && self_rank >= -1 && self_rank < world,
             "shard_keys: n=%lld world=%d self_rank=%d", (long long)n, world, self_rank);
  if (n == 0) return MI_OK;
  MI_REQUIRE(rows && keys, "shard_keys: null buffer");
  const int64_t chunks = entries_per_chunk > 0 ? mi::ceil_div(n, entries_per_chunk) : 1;
  MI_REQUIRE(chunks * world * rows_per_rank <= INT32_MAX, "shard_keys: %lld chunks x %d ranks x %lld rows per rank does not fit an int32 key",
             (long long)chunks, world, (long long)rows_per_rank);
  shard_keys_k<<<dim3((unsigned)mi::ceil_div(n, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(rows, n, world, entries_per_chunk,
                                                                                                   rows_per_rank, self_rank, keys);
  MI_CHECK_LAUNCH("shard_keys");
  return MI_OK;
}

int32_t mi_route_requests(const int32_t* uniq_keys, const int32_t* num_uniq, int64_t n_max, int64_t rows_per_rank,
                          int32_t n_groups, int32_t* send_rows, int32_t* counts, mi_stream_t stream) {
  MI_REQUIRE(n_max >= 0 && rows_per_rank > 0 && n_groups > 0, "route_requests: n_max=%lld", (long long)n_max);
  MI_REQUIRE(counts, "route_requests: null counts");
  hipStream_t st = mi::as_stream(stream);
  zero_i32(counts, n_groups, st);
  MI_CHECK_LAUNCH("route_requests(zero)");
  if (n_max == 0) return MI_OK;
  MI_REQUIRE(uniq_keys && num_uniq && send_rows, "route_requests: null buffer");
  route_requests_k<<<dim3((unsigned)mi::ceil_div(n_max, kBlock)), dim3(kBlock), 0, st>>>(uniq_keys, num_uniq, rows_per_rank, send_rows, counts);
  MI_CHECK_LAUNCH("route_requests");
  return MI_OK;
}

int32_t mi_segment_slots(const int32_t* seg_start, const int32_t* sorted_entry, const int32_t* num_uniq, int64_t n,
                         int32_t* slot_of_entry, mi_stream_t stream) {
  MI_REQUIRE(n >= 0, "segment_slots: n=%lld", (long long)n);
  if (n == 0) return MI_OK;
  MI_REQUIRE(seg_start && sorted_entry && num_uniq && slot_of_entry, "segment_slots: null buffer");
  segment_slots_k<<<dim3((unsigned)mi::ceil_div(n, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(seg_start, sorted_entry, num_uniq, n,
                                                                                                      slot_of_entry);
  MI_CHECK_LAUNCH("segment_slots");
  return MI_OK;
}

int32_t mi_gather_u32(const void* src, const int32_t* idx, int64_t n, void* out, mi_stream_t stream) {
  MI_REQUIRE(n >= 0, "gather_u32: n=%lld", (long long)n);
  if (n == 0) return MI_OK;
  MI_REQUIRE(src && idx && out, "gather_u32: null buffer");
  gather_u32_k<<<dim3((unsigned)mi::ceil_div(n, kBlock)), dim3(kBlock), 0, mi::as_stream(stream)>>>(
      static_cast<const uint32_t*>(src), idx, n, static_cast<uint32_t*>(out));
  MI_CHECK_LAUNCH("gather_u32");
  return MI_OK;
}

}  // extern "C"
