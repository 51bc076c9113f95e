// Launch plan + launcher of the LDS-DMA planes weight gradient (wgrad_pl.hip), called by gemm.hip's
// mi_dense_bwd_weight_planes, which owns the split-K slabs and their fixed-order reduction.
#pragma once
#include "common.h"

namespace mi {

struct WgradPlPlan {
  int tn, tm;          // workgroup tile: 128 tn output columns x 64 tm input features
  int tiles_n;         // N / (128 tn)
  int tiles_k;         // ceil(K / (64 tm))
  int splits;          // slabs over the examples
  int k_per_split;     // examples per slab (a multiple of 16)
  int nbuf;            // LDS stage buffers of the kernel (3 or 6 for the 128-column tile, 4 otherwise)
};

// false: shape not covered (N must be 128, 256 or 512, K a multiple of 128, M a multiple of 16)
bool wgrad_pl_plan(int64_t M, int N, int K, WgradPlPlan* p);

// sx, sy, sc: per-example f16 factors of the X rows, the dY rows and the bias gradient, kflag [M / 16]: which k-steps
// need sy (gemm.hip's wgrad_scale_k); slab [splits][K][N], cpart [splits][N] or NULL
int32_t wgrad_pl_launch(const WgradPlPlan& p, const mi_planes_t* X, const mi_planes_t* dY, const void* sx, const void* sy,
                        const void* sc, const int32_t* kflag, const float* amax_x, const float* amax_dy, float* slab, float* cpart, int64_t M, int N, int K,
                        hipStream_t st);

}  // namespace mi
