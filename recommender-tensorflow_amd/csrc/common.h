// Shared host-side helpers for the C-ABI translation units (error text, argument checks,
// launch checks).  gfx950 only: wave = 64 lanes everywhere in this directory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mi355x_rec.h"

namespace mi {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(mi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mi

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
