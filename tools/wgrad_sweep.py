#!/usr/bin/env python3
"""mi_dense_bwd_weight_planes (scale pass + split-K GEMM + fold) over the number of examples, for the three layer shapes of
config 3: the slope is what a k-step of a workgroup costs, the intercept the fixed part (profiles/r04_ab_layout_and_fold.md)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec import _lib
lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def chk(rc, w): _lib.check(rc, w)
class PB:
    def __init__(self, rows, K):
        self.data = torch.zeros(int(lib.mi_planes_bytes(rows, K)), dtype=torch.uint8, device="cuda")
        self.exp = torch.zeros(rows, dtype=torch.int32, device="cuda")
        self.s = _lib.Planes(self.data.data_ptr(), self.exp.data_ptr(), 64 * rows); self.ref = C.byref(self.s)
def split(x):
    pb = PB(x.shape[0], x.shape[1])
    chk(lib.mi_split_rows(x.data_ptr(), x.shape[1], x.shape[0], x.shape[1], 0, pb.ref, None, st()), "split"); return pb
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
wa = torch.randn(8192, 8192, device="cuda")
for _ in range(40): wa @ wa
torch.cuda.synchronize(); del wa
g = torch.Generator(device="cuda"); g.manual_seed(0)
for (N, K) in ((256, 512), (128, 256), (512, 1664)):
    for M in (16384, 32768, 65536, 131072):
        X = torch.relu(torch.randn(M, K, device="cuda", generator=g)); dY = torch.randn(M, N, device="cuda", generator=g) * 1e-5
        ax = torch.zeros(_lib.AMAX_SLOTS, device="cuda"); ay = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
        lib.mi_absmax(X.data_ptr(), X.numel(), ax.data_ptr(), st()); lib.mi_absmax(dY.data_ptr(), dY.numel(), ay.data_ptr(), st())
        xp, dyp = split(X), split(dY)
        ga = _lib.GemmAmax(ax.data_ptr(), ay.data_ptr(), None)
        wsz = torch.empty(int(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, N, K)) + 256, dtype=torch.uint8, device="cuda")
        dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
        t = timeit(lambda: chk(lib.mi_dense_bwd_weight_planes(xp.ref, dyp.ref, dW.data_ptr(), db.data_ptr(), M, N, K, wsz.data_ptr(), wsz.numel(), C.byref(ga), st()), "wg"))
        print("dW %4d x %3d  M = %6d: %7.1f us" % (K, N, M, t), flush=True)
        del X, dY, xp, dyp
