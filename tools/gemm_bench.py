#!/usr/bin/env python3
"""Times the fp32 MFMA GEMM entry points on the config-3 layer shapes (HIP events, interleaved rounds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec import _lib
L = _lib.load()
st = lambda: _lib.cur_stream()
p = lambda t: None if t is None else t.data_ptr()
M = 65536
shapes = [(1664, 512), (512, 256), (256, 128), (128, 1)]
g = torch.Generator(device="cuda"); g.manual_seed(0)
def rnd(*s): return torch.randn(*s, device="cuda", generator=g)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts) // 2], ts[0]
tot = 0.0
for K, N in shapes:
    X = rnd(M, K).relu_(); W = rnd(K, N) / K ** 0.5; b = rnd(N); Y = torch.empty(M, N, device="cuda")
    dY = rnd(M, N); dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
    ws = torch.empty(L.mi_dense_bwd_weight_workspace_bytes(M, N, K) + 256, dtype=torch.uint8, device="cuda")
    fl = 2.0 * M * N * K
    for name, fn in [
        ("fwd", lambda: L.mi_dense_fwd(p(X), K, p(W), p(b), p(Y), N, M, N, K, 1, 0.9, 123, st())),
        ("fwd_nodrop", lambda: L.mi_dense_fwd(p(X), K, p(W), p(b), p(Y), N, M, N, K, 1, 1.0, 123, st())),
        ("dgrad", lambda: L.mi_dense_bwd_data(p(dY), N, p(W), p(X), K, p(dX), K, M, N, K, 0.9, st())),
        ("wgrad", lambda: L.mi_dense_bwd_weight(p(X), K, p(dY), N, p(dW), p(db), M, N, K, p(ws), ws.numel(), st())),
    ]:
        med, mn = timeit(fn)
        if name != "fwd_nodrop": tot += med
        print("K=%5d N=%4d %-10s %8.1f us (min %8.1f)  %6.1f TF" % (K, N, name, med * 1e3, mn * 1e3, fl / med / 1e9))
print("sum fwd+dgrad+wgrad = %.3f ms  (%.1f TF overall)" % (tot, 3 * 2.0 * M * sum(k * n for k, n in shapes) / tot / 1e9))
