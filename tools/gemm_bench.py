#!/usr/bin/env python3
"""Times the MLP GEMM entry points on the config-3 layer shapes in the three matrix-pipe modes
(fp32-input MFMA, bf16x3 split, f16x2 split) and reports each mode's max error against fp64.
HIP events, interleaved rounds.   python tools/gemm_bench.py [--layer1-only]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec import _lib
L = _lib.load()
st = lambda: _lib.cur_stream()
p = lambda t: None if t is None else t.data_ptr()
M = 65536
shapes = [(1664, 512)] if "--layer1-only" in sys.argv else [(1664, 512), (512, 256), (256, 128)]
g = torch.Generator(device="cuda"); g.manual_seed(0)
def rnd(*s): return torch.randn(*s, device="cuda", generator=g)
def amax_of(t):
    v = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
    assert L.mi_absmax(p(t), t.numel(), p(v), st()) == 0
    return v
def timeit(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); rc = fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)); assert rc == 0
    ts.sort(); return ts[len(ts) // 2]
tot = {}
for K, N in shapes:
    X = rnd(M, K).relu_(); W = rnd(K, N) / K ** 0.5; b = rnd(N); Y = torch.empty(M, N, device="cuda")
    dY = rnd(M, N) * 1e-5; dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
    ws = torch.empty(L.mi_dense_bwd_weight_workspace_bytes(M, N, K) + 256, dtype=torch.uint8, device="cuda")
    aX, aW, adY, aout = amax_of(X), amax_of(W), amax_of(dY), torch.zeros(_lib.AMAX_SLOTS, device="cuda")
    R = 4096
    ref_f = (X[:R].double() @ W.double() + b.double()).relu()
    ref_d = dY[:R].double() @ W.double().T
    ref_w = X.double().T @ dY.double()
    fl = 2.0 * M * N * K
    for mode in ("fp32", "bf16x3", "f16x2"):
        L.mi_set_gemm_mode(0 if mode == "fp32" else 1)
        ga = lambda a, b_, o=None: _lib.GemmAmax(p(a), p(b_), p(o)) if mode == "f16x2" else None
        ops = [
            ("fwd", lambda: L.mi_dense_fwd(p(X), K, p(W), p(b), p(Y), N, M, N, K, 1, 1.0, 123, ga(aX, aW, aout), st()),
             lambda: (Y[:R].double() - ref_f).abs().max().item() / ref_f.abs().max().item()),
            ("dgrad", lambda: L.mi_dense_bwd_data(p(dY), N, p(W), None, K, p(dX), K, M, N, K, 1.0, 1, ga(adY, aW), st()),
             lambda: (dX[:R].double() - ref_d).abs().max().item() / ref_d.abs().max().item()),
            ("wgrad", lambda: L.mi_dense_bwd_weight(p(X), K, p(dY), N, p(dW), p(db), M, N, K, p(ws), ws.numel(), ga(aX, adY), st()),
             lambda: (dW.double() - ref_w).abs().max().item() / ref_w.abs().max().item()),
        ]
        for name, fn, err in ops:
            med = timeit(fn)
            tot[mode] = tot.get(mode, 0.0) + med
            print("K=%5d N=%4d %-6s %-7s %8.1f us  %6.1f TF-equiv   max err / max|ref| = %.2e" % (K, N, mode, name, med * 1e3, fl / med / 1e9, err()))
        if mode == "f16x2":
            print("   abs-max emitted by the fwd epilogue %.6g vs torch %.6g" % (aout.max().item(), Y.abs().max().item()))
L.mi_set_gemm_mode(1)
for mode, t in tot.items():
    print("sum fwd+dgrad+wgrad %-7s = %.3f ms" % (mode, t))
