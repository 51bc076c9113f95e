// Host-side entry points: library info, error text, and the categorical id transforms of
// trainers/ml_100k.py:19-35 (hash_bucket / bucketized columns).  Integer work only.
//
// Fingerprint64 is FarmHash's farmhashna::Hash64 (Google, MIT licence), restated here from the
// published algorithm: TensorFlow 1.12 (un-vendored dependency, environment.yml:10) calls it for
// tf.feature_column.categorical_column_with_hash_bucket.  The reference holds no vector for it;
// tests/test_fingerprint.py pins the 1-3 byte branch on TF's own string_to_hash_bucket test
// values and cross-checks every branch against the independent Python restatement in oracle/.
#include <string.h>

#include <string>

#include "common.h"

static const mi_step_state_t* g_step_state = nullptr;

namespace mi {

const mi_step_state_t* step_state() { return g_step_state; }

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace farm {

constexpr uint64_t k0 = 0xc3a5c85c97cb3127ULL;
constexpr uint64_t k1 = 0xb492b66fbe98f273ULL;
constexpr uint64_t k2 = 0x9ae16a3b2f90404fULL;

static inline uint64_t fetch64(const uint8_t* p) {
  uint64_t r;
  memcpy(&r, p, 8);  // little-endian host (x86-64)
  return r;
}
static inline uint32_t fetch32(const uint8_t* p) {
  uint32_t r;
  memcpy(&r, p, 4);
  return r;
}
static inline uint64_t rot(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
static inline uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static inline uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul;
  b ^= (b >> 47);
  b *= mul;
  return b;
}

static uint64_t hash_0to16(const uint8_t* s, size_t len) {
  if (len >= 8) {
    uint64_t mul = k2 + len * 2;
    uint64_t a = fetch64(s) + k2;
    uint64_t b = fetch64(s + len - 8);
    uint64_t c = rot(b, 37) * mul + a;
    uint64_t d = (rot(a, 25) + b) * mul;
    return hash_len16(c, d, mul);
  }
  if (len >= 4) {
    uint64_t mul = k2 + len * 2;
    uint64_t a = fetch32(s);
    return hash_len16(len + (a << 3), fetch32(s + len - 4), mul);
  }
  if (len > 0) {
    uint8_t a = s[0], b = s[len >> 1], c = s[len - 1];
    uint32_t y = static_cast<uint32_t>(a) + (static_cast<uint32_t>(b) << 8);
    uint32_t z = static_cast<uint32_t>(len) + (static_cast<uint32_t>(c) << 2);
    return shift_mix(y * k2 ^ z * k0) * k2;
  }
  return k2;
}

static uint64_t hash_17to32(const uint8_t* s, size_t len) {
  uint64_t mul = k2 + len * 2;
  uint64_t a = fetch64(s) * k1;
  uint64_t b = fetch64(s + 8);
  uint64_t c = fetch64(s + len - 8) * mul;
  uint64_t d = fetch64(s + len - 16) * k2;
  return hash_len16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + k2, 18) + c, mul);
}

static uint64_t hash_33to64(const uint8_t* s, size_t len) {
  uint64_t mul = k2 + len * 2;
  uint64_t a = fetch64(s) * k2;
  uint64_t b = fetch64(s + 8);
  uint64_t c = fetch64(s + len - 8) * mul;
  uint64_t d = fetch64(s + len - 16) * k2;
  uint64_t y = rot(a + b, 43) + rot(c, 30) + d;
  uint64_t z = hash_len16(y, a + rot(b + k2, 18) + c, mul);
  uint64_t e = fetch64(s + 16) * mul;
  uint64_t f = fetch64(s + 24);
  uint64_t g = (y + fetch64(s + len - 32)) * mul;
  uint64_t h = (z + fetch64(s + len - 24)) * mul;
  return hash_len16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
}

struct u128 {
  uint64_t first, second;
};

static inline u128 weak32(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b) {
  a += w;
  b = rot(b + a + z, 21);
  uint64_t c = a;
  a += x;
  a += y;
  b += rot(a, 44);
  return {a + z, b + c};
}
static inline u128 weak32(const uint8_t* s, uint64_t a, uint64_t b) {
  return weak32(fetch64(s), fetch64(s + 8), fetch64(s + 16), fetch64(s + 24), a, b);
}

static uint64_t hash64(const uint8_t* s, size_t len) {
  if (len <= 16) return hash_0to16(s, len);
  if (len <= 32) return hash_17to32(s, len);
  if (len <= 64) return hash_33to64(s, len);
  const uint64_t seed = 81;
  uint64_t x = seed;
  uint64_t y = seed * k1 + 113;
  uint64_t z = shift_mix(y * k2 + 113) * k2;
  u128 v{0, 0}, w{0, 0};
  x = x * k2 + fetch64(s);
  const uint8_t* end = s + ((len - 1) / 64) * 64;
  const uint8_t* last64 = end + ((len - 1) & 63) - 63;
  do {
    x = rot(x + y + v.first + fetch64(s + 8), 37) * k1;
    y = rot(y + v.second + fetch64(s + 48), 42) * k1;
    x ^= w.second;
    y += v.first + fetch64(s + 40);
    z = rot(z + w.first, 33) * k1;
    v = weak32(s, v.second * k1, x + w.first);
    w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
    uint64_t t = z;
    z = x;
    x = t;
    s += 64;
  } while (s != end);
  uint64_t mul = k1 + ((z & 0xff) << 1);
  s = last64;
  w.first += ((len - 1) & 63);
  v.first += w.first;
  w.first += v.first;
  x = rot(x + y + v.first + fetch64(s + 8), 37) * mul;
  y = rot(y + v.second + fetch64(s + 48), 42) * mul;
  x ^= w.second * 9;
  y += v.first * 9 + fetch64(s + 40);
  z = rot(z + w.first, 33) * mul;
  v = weak32(s, v.second * mul, x + w.first);
  w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
  uint64_t t = z;
  z = x;
  x = t;
  return hash_len16(hash_len16(v.first, w.first, mul) + shift_mix(y) * k0 + z,
                    hash_len16(v.second, w.second, mul) + x, mul);
}

}  // namespace farm
}  // namespace mi

extern "C" {

int32_t mi_set_step_state(const mi_step_state_t* device_state) {
  g_step_state = device_state;
  return MI_OK;
}

int32_t mi_abi_version(void) { return 21; }

const char* mi_last_error(void) { return mi::g_err; }

const char* mi_build_info(void) {
  static char info[128];
  snprintf(info, sizeof(info), "libmi355x_rec gfx950 HIP %d.%d.%d", HIP_VERSION_MAJOR,
           HIP_VERSION_MINOR, HIP_VERSION_PATCH);
  return info;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) — the checksum of TensorFlow's tensor-bundle checkpoint files
// (block trailers of the .index table, one per tensor in the .data shards; conf_utils.py:6-10 configures the Estimator
// that writes them).  Slicing-by-8 tables, built on first use; `crc` is the running value (0 to start), unmasked.
uint32_t mi_crc32c(const void* data, size_t len, uint32_t crc) {
  static uint32_t T[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xffu];
    ready = true;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = ~crc;
  while (len && (reinterpret_cast<uintptr_t>(p) & 7u)) { c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu]; --len; }
  while (len >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T[7][w & 0xff] ^ T[6][(w >> 8) & 0xff] ^ T[5][(w >> 16) & 0xff] ^ T[4][(w >> 24) & 0xff] ^
        T[3][(w >> 32) & 0xff] ^ T[2][(w >> 40) & 0xff] ^ T[1][(w >> 48) & 0xff] ^ T[0][(w >> 56) & 0xff];
    p += 8; len -= 8;
  }
  while (len--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu];
  return ~c;
}

uint64_t mi_fingerprint64(const void* data, size_t len) {
  return mi::farm::hash64(static_cast<const uint8_t*>(data), len);
}

int32_t mi_hash_bucket_i64(const int64_t* values, int64_t n, int64_t num_buckets,
                           int32_t* out_ids) {
  MI_REQUIRE(n >= 0 && num_buckets > 0 && num_buckets <= INT32_MAX, "hash_bucket_i64: n=%lld buckets=%lld",
             (long long)n, (long long)num_buckets);
  MI_REQUIRE(n == 0 || (values && out_ids), "hash_bucket_i64: null buffer");
  char buf[32];
  for (int64_t i = 0; i < n; ++i) {
    // tf.as_string on an integer tensor: plain decimal, '-' for negatives
    int len = snprintf(buf, sizeof(buf), "%lld", (long long)values[i]);
    out_ids[i] = static_cast<int32_t>(mi_fingerprint64(buf, (size_t)len) % (uint64_t)num_buckets);
  }
  return MI_OK;
}

int32_t mi_hash_bucket_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n,
                             int64_t num_buckets, int32_t* out_ids) {
  MI_REQUIRE(n >= 0 && num_buckets > 0 && num_buckets <= INT32_MAX, "hash_bucket_bytes: n=%lld buckets=%lld",
             (long long)n, (long long)num_buckets);
  MI_REQUIRE(n == 0 || (offsets && out_ids), "hash_bucket_bytes: null buffer");
  for (int64_t i = 0; i < n; ++i) {
    int64_t lo = offsets[i], hi = offsets[i + 1];
    MI_REQUIRE(hi >= lo && lo >= 0, "hash_bucket_bytes: offsets not monotone at %lld", (long long)i);
    out_ids[i] = static_cast<int32_t>(mi_fingerprint64(bytes + lo, (size_t)(hi - lo)) %
                                      (uint64_t)num_buckets);
  }
  return MI_OK;
}

int32_t mi_bucketize_f32(const float* values, int64_t n, const float* boundaries,
                         int32_t num_boundaries, int32_t* out_ids) {
  MI_REQUIRE(n >= 0 && num_boundaries >= 0, "bucketize: n=%lld nb=%d", (long long)n, num_boundaries);
  MI_REQUIRE(n == 0 || (values && out_ids), "bucketize: null buffer");
  for (int32_t j = 1; j < num_boundaries; ++j)
    MI_REQUIRE(boundaries[j] > boundaries[j - 1], "bucketize: boundaries must be strictly increasing");
  for (int64_t i = 0; i < n; ++i) {
    // std::upper_bound: number of boundaries <= x
    int32_t lo = 0, hi = num_boundaries;
    const float x = values[i];
    while (lo < hi) {
      int32_t mid = (lo + hi) >> 1;
      if (boundaries[mid] <= x) lo = mid + 1; else hi = mid;
    }
    out_ids[i] = lo;
  }
  return MI_OK;
}

}  // extern "C"
