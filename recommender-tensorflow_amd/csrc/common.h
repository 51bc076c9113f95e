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

// tuning experiments only (tools/*_bench.py): an integer from the environment, else the built-in value
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

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
