#!/usr/bin/env python3
"""Runs the three layer-1 GEMMs a few times (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec import _lib
L = _lib.load(); st = lambda: _lib.cur_stream(); p = lambda t: None if t is None else t.data_ptr()
M, K, N = 65536, 1664, 512
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.randn(M, K, device="cuda", generator=g).relu_(); W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g); Y = torch.empty(M, N, device="cuda"); dY = torch.randn(M, N, device="cuda", generator=g)
dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for _ in range(3):
    L.mi_dense_fwd(p(X), K, p(W), p(b), p(Y), N, M, N, K, 1, 1.0, 123, st())
    L.mi_dense_bwd_data(p(dY), N, p(W), None, K, p(dX), K, M, N, K, 1.0, st())
    L.mi_dense_bwd_weight(p(X), K, p(dY), N, p(dW), p(db), M, N, K, p(ws), ws.numel(), st())
torch.cuda.synchronize()
