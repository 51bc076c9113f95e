// Weight gradient dW[K][N] = X^T dY of a dense layer (model_utils.py:69-72: optimizer.minimize's gradients of
// tf.layers.dense, deep_fm.py:98-108) with both operands read as planes (csrc/gemm_pl.hip) and staged by LDS-DMA
// like the forward pass — the structure of gemm_pl_k (512 threads, 16-example k-steps, 3-4 stage buffers,
// counted vmcnt, the two waves of a SIMD ping-pong between loading and MFMA) with the reduction running over
// the EXAMPLES.
//
// In the planes layout a 16-feature block of 16 consecutive examples is 1 KiB of contiguous memory
// ([example][16 hi | 16 lo]), so a stage is a handful of 1-KiB runs and the LDS image is the memory image.
// The MFMA wants, per lane, 8 consecutive k (= examples) of ONE feature: ds_read_b64_tr_b16 delivers exactly
// that from the example-major image (a 16-lane group reads 4 examples x 16 features and hands lane i feature
// i of the 4 examples), two reads per fragment.  Odd blocks are stored with their high and low halves swapped
// (a permutation of the 16-byte pieces on the LDS-DMA source address) so that the two blocks a 32-lane half
// reads fall on different banks.
//
// A plane row carries its own exponent: example m's products carry 2^(sx[m] + sy[m]).  One power of two per example,
// 2^d[m] with d[m] = (SX - sx[m]) + (SY - sy[m]) <= 0 (SX, SY: exponents of the matrices' abs-max), shared out over
// the two operands as f16 factors (2^max(d, -24) on X, the rest — rarely anything — on dY, zero below 2^-38: gemm.hip's wgrad_scale_k)
// and multiplied into the fragments (v_pk_mul_f16, exact unless the result is subnormal), brings every example to
// the matrix-wide scales, which the epilogue undoes — the products of gemm.hip's matrix-wide f16x2 split (examples
// far below the abs-max go subnormal and lose low bits: invisible in a sum over examples; the MFMA keeps fp16
// subnormals: test_weight_gradient_keeps_fp16_subnormal_operands).  The factors of a k-step are wave-uniform:
// scalar loads.
//
// MFMA A operand = dY^T (rows = output columns n: a workgroup holds ALL N = 128 TN of them), B operand = X^T
// (rows = input features, 64 TM per workgroup); split-K over the examples into slabs that gemm.hip's
// slab_reduce_k folds in a fixed order (bitwise reproducible).  The bias gradient (column sums of dY) rides in
// the workgroups of the first feature tile: v_dot2_f32_f16 of the dY fragments with 2^(SY - sy[m]).
#include "common.h"
#include "wgrad_pl.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_THREADS = 512;

struct WgArgs {
  const char* A; int64_t bsa;         // dY planes: [N / 16][examples][64 B]
  const char* B; int64_t bsb;         // X planes:  [K / 16][examples][64 B]
  const float* amax_a; const float* amax_b;   // abs-max vectors of X and dY
  int M, N, K;                        // examples; dW is [K][N]
  int k_per_split, tiles_k, tiles_n;
  float* slab;                        // [splits][K][N]
  float* cpart;                       // [splits][N] bias-gradient partials, or NULL
};

__device__ __forceinline__ float wg_pow2(int s) { return __uint_as_float(static_cast<uint32_t>(127 + s) << 23); }
__device__ __forceinline__ int wg_scale_exp(const float* __restrict__ amax) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < MI_AMAX_SLOTS; ++j) m = fmaxf(m, amax[j]);
  const int e = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);
  return max(-100, min(100, 141 - e));
}
template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Two 4-example transposed reads (examples +0..3 and +4..7 of the lane half's eight) = one MFMA fragment.
// Inline asm on purpose: for the ds_read_tr intrinsic hipcc cannot tell which LDS-DMA stores the read depends on
// and parks an s_waitcnt vmcnt(0) in front of it — draining the whole stage pipeline every k-step.  The caller
// waits (lgkmcnt(0) + sched_barrier) before it touches the results.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ f16x8 wg_tr_read(uint32_t addr) {
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(hi) : "v"(addr), "n"(OFF + 256));
  const uint4 v = make_uint4(lo[0], lo[1], hi[0], hi[1]);
  return __builtin_bit_cast(f16x8, v);
}
template <int X, int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (X < N) {
    f(std::integral_constant<int, X>{});
    wg_static_for<X + 1, N>(f);
  }
}

template <int TN, int TM, int NBUF_ = (TN == 1 ? 3 : 4)>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_pl_k(const WgArgs a, const uint4* __restrict__ sx_all,
                                                            const uint4* __restrict__ sy_all,
                                                            const uint4* __restrict__ sc_all,
                                                            const int32_t* __restrict__ kflag_all) {
  constexpr int NBUF = NBUF_;
  constexpr int NBLK = 8 * TN + 4 * TM;                  // 1-KiB blocks (16 features x 16 examples) per stage
  constexpr int STAGE = NBLK * 1024;
  constexpr int LPS = NBLK / 8;                          // LDS-DMA instructions per thread and stage
  static_assert(NBLK % 8 == 0, "stage blocks");
  static_assert(NBUF * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * STAGE];

  const int nb = gridDim.x, bid = blockIdx.x;
  const int qq = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < rr) ? xcd * (qq + 1) + idx : rr * (qq + 1) + (xcd - rr) * qq + idx;
  // (the tiles of one example range run on one XCD: they read the same rows of both operands)
  const int tile_k = lid % a.tiles_k, tile_n = (lid / a.tiles_k) % a.tiles_n, split = lid / (a.tiles_k * a.tiles_n);
  const int kx0 = tile_k * 64 * TM, n0 = tile_n * 128 * TN;
  const int kblk_last = (a.K >> 4) - 1;                  // (a ragged last feature tile re-reads the last block, stores nothing for it)
  const int k0 = split * a.k_per_split;
  const int nk = (min(a.M, k0 + a.k_per_split) - k0) / 16;

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wn = wv & 3, wm = wv >> 2;

  // ---- LDS-DMA source addresses: chunk p = j * 512 + t of a stage lands at LDS byte 16 p ----
  const char* src[LPS];
#pragma unroll
  for (int j = 0; j < LPS; ++j) {
    const int p = j * WG_THREADS + t;
    const int b = p >> 6, e = (p >> 2) & 15, piece = (p & 3) ^ ((b & 1) << 1);
    if (j < TN) src[j] = a.A + static_cast<int64_t>((n0 >> 4) + b) * a.bsa + static_cast<int64_t>(k0 + e) * 64 + piece * 16;
    else src[j] = a.B + static_cast<int64_t>(min((kx0 >> 4) + b - 8 * TN, kblk_last)) * a.bsb + static_cast<int64_t>(k0 + e) * 64 + piece * 16;
  }

  // ---- transposed fragment reads: lane 4 q + pp of a 16-lane group addresses example q, features 4 pp .. 4 pp + 3 ----
  const int l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int rowoff = (8 * h + (l16 >> 2)) * 64 + (l16 & 3) * 8;
  const int offA = (wn * 2 * TN + g1) * 1024 + rowoff;
  const int offB = (8 * TN + wm * 2 * TM + g1) * 1024 + rowoff;
  const int ch = g1 * 32, cl = (g1 ^ 1) * 32;            // (odd blocks hold [lo | hi])

  f32x16 acc[TN][TM];
#pragma unroll
  for (int x = 0; x < TN; ++x)
#pragma unroll
    for (int y = 0; y < TM; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  auto issue_share = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < LPS; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + static_cast<int64_t>(kt) * 1024),
                                       (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (j * WG_THREADS + wv * 64) * 16),
                                       16, 0, 0);
  };
  f16x8 ah[TN], al[TN], bh[TM], bl[TM];
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  auto read_frags = [&](int bufi) {
    const uint32_t buf = lds0 + bufi * STAGE;
    const uint32_t aH = buf + offA + ch, aL = buf + offA + cl, bH = buf + offB + ch, bL = buf + offB + cl;
    wg_static_for<0, TN>([&](auto xc) {
      constexpr int x = decltype(xc)::value;
      ah[x] = wg_tr_read<x * 2048>(aH);
      al[x] = wg_tr_read<x * 2048>(aL);
    });
    wg_static_for<0, TM>([&](auto yc) {
      constexpr int y = decltype(yc)::value;
      bh[y] = wg_tr_read<y * 2048>(bH);
      bl[y] = wg_tr_read<y * 2048>(bL);
    });
  };
  // C(t): the MFMAs of tile t in product-major order and, pinned between them, this wave's LDS-DMA share of stage
  // t - 1 + NBUF (gemm_pl_k: issued next to the fragment reads the pieces made L longer than C); the 128-column tile
  // (3 buffers) keeps its share at the head of L(t)
  constexpr bool DMA_IN_C = TN != 1;
  auto phase_c = [&](int t) {
    int fb = t - 1 + NBUF;
    fb -= (fb / NBUF) * NBUF;
    const int kt = min(t - 1 + NBUF, nk - 1);             // (the tail re-issues the last tile: uniform counts)
    constexpr int NT = TN * TM, NM = 3 * NT, GAP = NM / (LPS + 1);
#pragma unroll
    for (int idx = 0; idx < NM; ++idx) {
      const int pr = idx / NT, x = (idx % NT) / TM, y = idx % TM;
      acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[x] : ah[x], pr == 1 ? bl[y] : bh[y], acc[x][y], 0, 0, 0);
      if (DMA_IN_C && (idx + 1) % GAP == 0 && (idx + 1) / GAP <= LPS) {
        const int j = (idx + 1) / GAP - 1;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + static_cast<int64_t>(kt) * 1024),
                                         (__attribute__((address_space(3))) void*)(smem + fb * STAGE + (j * WG_THREADS + wv * 64) * 16),
                                         16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // per-example factors of a k-step: 16 f16 = two 16-byte scalar loads, a lane keeps the eight of its half
  // (direct __restrict__ kernel parameters: only then does hipcc emit scalar loads — as vector loads they would
  // count in vmcnt and drain the LDS-DMA pipeline every k-step)
  const uint4* __restrict__ sxp = sx_all + (k0 >> 3);
  const uint4* __restrict__ syp = sy_all + (k0 >> 3);
  const uint4* __restrict__ scp = sc_all + (k0 >> 3);
  const int32_t* __restrict__ kfp = kflag_all + (k0 >> 4);
  const bool do_cs = a.cpart != nullptr && tile_k == 0 && wm == 0;       // (wave-uniform)
  float cs[TN];
#pragma unroll
  for (int x = 0; x < TN; ++x) cs[x] = 0.f;
  // (fetched ONE k-step ahead into the other of two register sets — the loop below is unrolled by two so that the
  // sets alternate without copies: a scalar load issued in the k-step that needs it exposes an L2 round trip)
  struct Factors { uint4 xa, xb, ca, cb; int flag; };
  Factors f0, f1;
  f0.ca = f0.cb = f1.ca = f1.cb = make_uint4(0u, 0u, 0u, 0u);
  auto fetch_factors = [&](int kt, Factors& f) {
    const int u = __builtin_amdgcn_readfirstlane(kt);
    f.xa = sxp[2 * u]; f.xb = sxp[2 * u + 1];
    f.flag = kfp[u];
    if (do_cs) { f.ca = scp[2 * u]; f.cb = scp[2 * u + 1]; }
  };
  auto mul8 = [](f16x8 v, uint4 s) { return v * __builtin_bit_cast(f16x8, s); };
  auto scale_and_sum = [&](int kt, const Factors& f) {
    if (do_cs) {                      // (from the fragments as stored: the bias gradient has its own factor)
      const uint4 cv = h ? f.cb : f.ca;
      const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
      for (int x = 0; x < TN; ++x) {
        const uint4 hv = __builtin_bit_cast(uint4, ah[x]), lv = __builtin_bit_cast(uint4, al[x]);
        const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cs[x] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, hw[r]), __builtin_bit_cast(h16x2, cw[r]), cs[x], false);
          cs[x] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, lw[r]), __builtin_bit_cast(h16x2, cw[r]), cs[x], false);
        }
      }
    }
    const uint4 sv = h ? f.xb : f.xa;
#pragma unroll
    for (int y = 0; y < TM; ++y) { bh[y] = mul8(bh[y], sv); bl[y] = mul8(bl[y], sv); }
    if (f.flag) {                     // rare: an example more than 2^-24 below the abs-max — the rest of its factor
      const int u = __builtin_amdgcn_readfirstlane(kt);
      const uint4 ya = syp[2 * u], yb = syp[2 * u + 1];
      const uint4 yv = h ? yb : ya;
#pragma unroll
      for (int x = 0; x < TN; ++x) { ah[x] = mul8(ah[x], yv); al[x] = mul8(al[x], yv); }
    }
  };
  // L(t): share of stage t - 1 + NBUF into the buffer tile t - 1 has left, then the fragments of tile t
  auto phase_l = [&](int t) {
    if (!DMA_IN_C && t >= 1) {
      int fb = t - 1;
      fb -= (fb / NBUF) * NBUF;
      issue_share(min(t - 1 + NBUF, nk - 1), fb);         // (the tail re-issues the last tile: uniform counts)
    }
    int rb = t;
    rb -= (rb / NBUF) * NBUF;
    read_frags(rb);
  };
  const bool g1w = __builtin_amdgcn_readfirstlane(wm) != 0;

  // the software pipeline of gemm_pl_k (see there): waves 4-7 run one barrier behind waves 0-3
#pragma unroll
  for (int s = 0; s < NBUF; ++s) issue_share(min(s, nk - 1), s);
  wg_wait_vmcnt<LPS*(NBUF - 1)>();
  __builtin_amdgcn_s_barrier();
  if (g1w) __builtin_amdgcn_s_barrier();
  auto step = [&](int t, const Factors& cur, Factors& nxt) {
    phase_l(t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);                     // nothing that reads a fragment moves above the wait
    scale_and_sum(t, cur);
    fetch_factors(min(t + 1, nk - 1), nxt);
    wg_wait_vmcnt<LPS*(DMA_IN_C ? NBUF - 3 : NBUF - 2)>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    phase_c(t);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  fetch_factors(0, f0);
#pragma unroll 1
  for (int t = 0; t < nk; t += 2) {
    step(t, f0, f1);
    if (t + 1 < nk) step(t + 1, f1, f0);
  }
  if (!g1w) __builtin_amdgcn_s_barrier();
  wg_wait_vmcnt<0>();                 // the tail's re-issued loads

  // ---------------------------------------------------------------- epilogue: the slab of this split
  const int sx_e = wg_scale_exp(a.amax_a), sy_e = wg_scale_exp(a.amax_b);
  const float fa = wg_pow2(-sx_e), fb = wg_pow2(-sy_e);
  float* Cb = a.slab + static_cast<int64_t>(split) * a.K * a.N;
#pragma unroll
  for (int y = 0; y < TM; ++y) {
    const int kx = kx0 + wm * 32 * TM + y * 32 + i;
    float* row = Cb + static_cast<int64_t>(kx) * a.N + n0 + wn * 32 * TN + 4 * h;
    if (kx < a.K)
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(row + x * 32 + 8 * g) =
            make_float4(acc[x][y][4 * g] * fa * fb, acc[x][y][4 * g + 1] * fa * fb, acc[x][y][4 * g + 2] * fa * fb,
                        acc[x][y][4 * g + 3] * fa * fb);
  }
  if (do_cs) {
#pragma unroll
    for (int x = 0; x < TN; ++x) {
      const float c = cs[x] + __shfl_xor(cs[x], 32);    // the two halves hold different examples
      if (h == 0) a.cpart[static_cast<int64_t>(split) * a.N + n0 + wn * 32 * TN + x * 32 + i] = c * fb;
    }
  }
}

}  // namespace

namespace mi {

bool wgrad_pl_plan(int64_t M, int N, int K, WgradPlPlan* p) {
  if (M <= 0 || M % 16 != 0 || (N != 128 && N != 256 && N != 512) || K <= 0 || K % 16 != 0) return false;
  p->tn = N / 128;
  p->tiles_n = 1;
  p->tm = p->tn == 2 ? env_int("MI_WGRAD_TM_N256", 2) : 2;      // (N = 256: 256 x 128 tiles — half the slab bytes of 256 x 256: 103 -> 90 us at 65536 x 512 x 256)
  p->nbuf = p->tn == 1 ? env_int("MI_WGRAD_NBUF_N128", 3) : 4;
  // N = 512: one 512-column tile x 128 features.  MI_WGRAD_TILE=256 (A/B runs): two 256-column tiles x 256 features
  // (32 KB of operands per k-step instead of 40 KB for the same MFMAs, a half-empty last feature tile at K = 1664) —
  // measured slower, 469 vs 451 us on the layer-1 shape: the loop is not bound by operand bytes.
  if (p->tn == 4) {
    if (env_int("MI_WGRAD_TILE", 512) == 256) { p->tn = 2; p->tm = 4; p->tiles_n = 2; }      // (the tools' build only)
  }
  if (p->tn == 2 && env_int("MI_WGRAD_N256_AS_128", 0)) { p->tn = 1; p->tm = 2; p->tiles_n = 2; p->nbuf = env_int("MI_WGRAD_NBUF_N128", 3); }   // (the tools' build only)
  if (K % 128 != 0) return false;
  p->tiles_k = static_cast<int>(ceil_div(K, 64 * p->tm));      // (the last tile may be half empty)
  // one round of resident workgroups (one per CU; the 128-column tile fits two), at least 32 k-steps per split
  // (a split pays a pipeline fill and a slab of the whole tile)
  int64_t target = (p->tn == 1 ? 512 : 256) / (p->tiles_k * p->tiles_n);
  const int min_ksteps = env_int("MI_WGRAD_MIN_KSTEPS", 32);
  const int64_t max_s = M / (16 * min_ksteps) > 0 ? M / (16 * min_ksteps) : 1;
  if (target > max_s) target = max_s;
  if (target < 1) target = 1;
  p->k_per_split = static_cast<int>(ceil_div(ceil_div(M, target), 16) * 16);
  p->splits = static_cast<int>(ceil_div(M, p->k_per_split));
  return true;
}

int32_t wgrad_pl_launch(const WgradPlPlan& p, const mi_planes_t* X, const mi_planes_t* dY, const void* sx, const void* sy,
                        const void* sc, const int32_t* kflag, const float* amax_x, const float* amax_dy, float* slab, float* cpart, int64_t M, int N, int K,
                        hipStream_t st) {
  WgArgs a{};
  a.A = static_cast<const char*>(dY->data); a.bsa = dY->blk_stride;
  a.B = static_cast<const char*>(X->data); a.bsb = X->blk_stride;
  const uint4* sx4 = static_cast<const uint4*>(sx);
  const uint4* sy4 = static_cast<const uint4*>(sy);
  const uint4* sc4 = static_cast<const uint4*>(sc);
  a.amax_a = amax_x; a.amax_b = amax_dy;
  a.M = static_cast<int>(M); a.N = N; a.K = K;
  a.k_per_split = p.k_per_split; a.tiles_k = p.tiles_k; a.tiles_n = p.tiles_n;
  a.slab = slab; a.cpart = cpart;
  const dim3 g(static_cast<unsigned>(p.tiles_k * p.tiles_n * p.splits)), b(WG_THREADS);
  if (p.tn == 4 && p.tm == 2) wgrad_pl_k<4, 2><<<g, b, 0, st>>>(a, sx4, sy4, sc4, kflag);
  else if (p.tn == 2 && p.tm == 4) wgrad_pl_k<2, 4><<<g, b, 0, st>>>(a, sx4, sy4, sc4, kflag);
  else if (p.tn == 2 && p.tm == 2) wgrad_pl_k<2, 2><<<g, b, 0, st>>>(a, sx4, sy4, sc4, kflag);
  else if (p.tn == 1 && p.tm == 4) wgrad_pl_k<1, 4><<<g, b, 0, st>>>(a, sx4, sy4, sc4, kflag);
  else if (p.tn == 1 && p.tm == 2 && p.nbuf == 6) wgrad_pl_k<1, 2, 6><<<g, b, 0, st>>>(a, sx4, sy4, sc4, kflag);
  else if (p.tn == 1 && p.tm == 2) wgrad_pl_k<1, 2><<<g, b, 0, st>>>(a, sx4, sy4, sc4, kflag);
  else {
    set_error("wgrad_pl_launch: no kernel for tile %d x %d", p.tn, p.tm);
    return MI_ERR_INVALID;
  }
  MI_CHECK_LAUNCH("dense_bwd_weight_planes(LDS-DMA)");
  return MI_OK;
}

}  // namespace mi
