#!/usr/bin/env python3
"""A/B of GEMM loop variants: loads each tools/var_*.so and times the layer-1 shapes, interleaved rounds."""
import ctypes as C, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec import _lib
libs = {}
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "var_*.so"))):
    l = C.CDLL(path)
    for name in ("mi_dense_fwd", "mi_dense_bwd_data", "mi_dense_bwd_weight", "mi_dense_bwd_weight_workspace_bytes"):
        f = getattr(l, name); f.restype, f.argtypes = _lib.SIGNATURES[name]
    libs[os.path.basename(path)] = l
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
M, K, N = 65536, 1664, 512
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.randn(M, K, device="cuda", generator=g).relu_(); W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g); Y = torch.empty(M, N, device="cuda"); dY = torch.randn(M, N, device="cuda", generator=g)
dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
fl = 2.0 * M * N * K
res = {}
for rnd in range(5):
    for name, L in libs.items():
        for op, fn in [("fwd", lambda: L.mi_dense_fwd(p(X), K, p(W), p(b), p(Y), N, M, N, K, 1, 1.0, 123, st())),
                       ("dgrad", lambda: L.mi_dense_bwd_data(p(dY), N, p(W), None, K, p(dX), K, M, N, K, 1.0, st())),
                       ("wgrad", lambda: L.mi_dense_bwd_weight(p(X), K, p(dY), N, p(dW), p(db), M, N, K, p(ws), ws.numel(), st()))]:
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); rc = fn(); e.record(); torch.cuda.synchronize()
            assert rc == 0
            if rnd: res.setdefault((name, op), []).append(s.elapsed_time(e))
for (name, op), ts in sorted(res.items()):
    ts.sort(); print("%-14s %-6s median %8.1f us  min %8.1f  -> %6.1f TF" % (name, op, ts[len(ts)//2]*1e3, ts[0]*1e3, fl/ts[len(ts)//2]/1e9))
