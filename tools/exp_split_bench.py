#!/usr/bin/env python3
"""Experiment: bf16x3-split GEMM (NT) vs the fp32-MFMA data-gradient GEMM on the layer-1 shape."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import numpy as np, torch
from mi355x_rec import _lib
L = _lib.load()
X = C.CDLL(os.path.join(ROOT, "tools", "exp_split.so"))
X.mi_exp_gemm_nt_bf16x3.restype = C.c_int32
X.mi_exp_gemm_nt_bf16x3.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for (M, N, K) in [(65536, 1664, 512), (65536, 512, 256), (4096, 512, 1664)]:
    A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    C0 = torch.empty(M, N, device="cuda"); C1 = torch.empty(M, N, device="cuda")
    # fp32 MFMA reference: dX = dY * W^T with W stored [N_g=K_in][K_g]: mi_dense_bwd_data(dY[M,Kg], W[Kin,Kg]) -> [M,Kin]
    f0 = lambda: L.mi_dense_bwd_data(p(A), K, p(B), None, N, p(C0), N, M, K, N, 1.0, st())
    f1 = lambda: X.mi_exp_gemm_nt_bf16x3(p(A), K, p(B), K, p(C1), N, M, N, K, st())
    res = {}
    for name, fn in (("fp32 mfma", f0), ("bf16x3 split", f1)):
        assert fn() == 0; torch.cuda.synchronize(); ts = []
        for _ in range(10):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        ts.sort(); res[name] = ts[len(ts) // 2]
    n = min(M, 2048)
    ref = A[:n].double() @ B.double().T
    sc = ref.pow(2).mean().sqrt()
    e0 = float((C0[:n].double() - ref).abs().max() / sc); e1 = float((C1[:n].double() - ref).abs().max() / sc)
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d: fp32 mfma %7.1f us (%5.1f TF, err %.2e) | bf16x3 %7.1f us (%5.1f TF-equiv, err %.2e)" % (
        M, N, K, res["fp32 mfma"] * 1e3, fl / res["fp32 mfma"] / 1e9, e0, res["bf16x3 split"] * 1e3, fl / res["bf16x3 split"] / 1e9, e1))
