// Shared host-side helpers for the C-ABI translation units (error text, argument checks,
// launch checks).  gfx950 only: wave = 64 lanes everywhere in this directory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/mi355x_rec.h"

namespace mi {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(mi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Tuning switches (MI_PL_TILE, MI_WGRAD_TILE, MI_WGRAD_PL, MI_CATCHUP_BLOCKS, MI_CATCHUP_DEPTH, MI_SORT_MEMSET, MI_SORT_FUSED, MI_SORT_BITS, MI_SORT_NAP): read from the
// environment ONLY in the tools' build of the library (`make tuning` -> tools/probe/libmi355x_rec_tuning.so, -DMI_TUNING; the
// tools/*_bench.py scripts load that file).  The shipped libmi355x_rec.so has no environment switch: env_int is the
// built-in value, and the measured-and-dropped variants behind the switches are dead code the compiler removes.
#ifdef MI_TUNING
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}
#else
constexpr int env_int(const char*, int dflt) { return dflt; }
#endif

// device-resident step state registered by mi_set_step_state (host_ids.cpp), or nullptr
const mi_step_state_t* step_state();

}  // namespace mi

// Publish a workgroup's abs-max into an abs-max vector (MI_AMAX_SLOTS floats, value = largest entry).
// Every thread of the workgroup must call it (it contains a barrier).  One atomic per workgroup at
// most, spread over the slots, and only when the slot does not already hold a larger value: same-
// address atomics serialise at ~0.2 us each, and a kernel's first generation of workgroups ends together.
__device__ __forceinline__ void mi_amax_publish(float* __restrict__ vec, float mx) {
  __shared__ float part[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) mx = fmaxf(mx, part[w]);
    unsigned int* slot = reinterpret_cast<unsigned int*>(vec) + (blockIdx.x & (MI_AMAX_SLOTS - 1));
    const unsigned int bits = __float_as_uint(mx);
    if (bits > *reinterpret_cast<volatile unsigned int*>(slot)) atomicMax(slot, bits);
  }
}

// ---- dropout mask (tf.layers.dropout, deep_fm.py:102-103): counter-based keep decisions, TWO per 32-bit hash --------
// Element (row, col) of a layer's output is kept iff its 16 bits of hash(row, col >> 1) — the low half for an even
// column, the high half for an odd one — are below keep_prob * 2^16.  The per-ROW key goes through a multiply-xorshift
// mixer (two v_mul_lo_u32, quarter rate on CDNA: paid once per row and lane); the per-PAIR mixer is a
// shift-add-xor chain in the manner of Thomas Wang's 32-bit integer hash, shifts, adds and xors only (round 2's mask paid 3.5 quarter-rate multiplies per
// ELEMENT: ~25 % of a GEMM epilogue).  tests/util.py replays the mask on the host, every kernel here uses these.
__device__ __forceinline__ uint32_t mi_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
constexpr uint32_t MI_DROP_PAIR_MUL = 0x85EBCA77U;
__device__ __forceinline__ uint32_t mi_drop_rowkey(uint64_t seed, uint32_t row) {
  const uint32_t s = static_cast<uint32_t>(seed) ^ (static_cast<uint32_t>(seed >> 32) * 0xC2B2AE35U);
  return mi_mix32((row * 0x9E3779B1U) ^ s);
}
// pair_term = (col >> 1) * MI_DROP_PAIR_MUL (callers with compile-time column offsets fold the multiply into constants)
__device__ __forceinline__ uint32_t mi_drop_pairhash(uint32_t rowkey, uint32_t pair_term) {
  // (every line below is one or two full-rate instructions — v_lshl_add_u32 / v_lshrrev + v_xor; no step is a sum of two
  // shifted copies, which hipcc would turn back into a quarter-rate v_mul_lo_u32)
  uint32_t x = rowkey + pair_term;
  x = ~x + (x << 15);
  x ^= x >> 12;
  x += x << 2;
  x ^= x >> 4;
  x += x << 3;
  x ^= x >> 11;
  x += x << 11;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t mi_drop_thresh16(float keep_prob) { return static_cast<uint32_t>(keep_prob * 65536.0f); }
__device__ __forceinline__ bool mi_drop_keep(uint32_t pairhash, uint32_t col, uint32_t thresh16) {
  return ((col & 1u) ? (pairhash >> 16) : (pairhash & 0xffffu)) < thresh16;
}
__device__ __forceinline__ bool mi_drop_keep_at(uint64_t seed, uint32_t row, uint32_t col, uint32_t thresh16) {
  return mi_drop_keep(mi_drop_pairhash(mi_drop_rowkey(seed, row), (col >> 1) * MI_DROP_PAIR_MUL), col, thresh16);
}

// ---- x / d for a divisor d known on the host (tf.nn.dropout: div(x, keep_prob)) ----------------------------------------
// Markstein's short division: with r = RN(1 / d) (computed on the host with IEEE division), q = RN(x r) is within an ulp of
// the quotient, the residual e = x - d q is exact in one fma, and RN(q + e r) is the correctly rounded x / d — the bits of
// the '/' operator — in 3 instructions instead of hipcc's ~12 (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup).
// Not taken on trust: mi_selftest_div compares the two on the device for EVERY fp32 x (tests: 0 mismatches for
// |x| >= 2^-100, where the residual of a quotient is still exactly representable; below that — values of 1e-30 — the last
// bit of the quotient can differ).  d = 1 gives x back exactly.
__device__ __forceinline__ float mi_div_const(float x, float d, float r) {
  const float q = x * r;
  const float e = fmaf(-d, q, x);
  return fmaf(e, r, q);
}

#define MI_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      mi::set_error(__VA_ARGS__);        \
      return MI_ERR_INVALID;             \
    }                                    \
  } while (0)

#define MI_CHECK_LAUNCH(what)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      mi::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));     \
      return MI_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)
