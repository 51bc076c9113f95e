#!/usr/bin/env python3
"""Phase timing of the split GEMM main loop from in-kernel s_memtime stamps.
Needs a library built with -DMI_GEMM_STAMPS:
  make -C recommender-tensorflow_amd/csrc OBJDIR=/tmp/bst OUT=../../tools/probe/libstamps.so \
       CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DMI_GEMM_STAMPS"
Per k-tile: [MFMAs + split/stores of t+1 + loads of t+2] | barrier | fragment reads | barrier | loop-back."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import numpy as np, torch
from mi355x_rec import _lib
L = C.CDLL(os.path.join(ROOT, "tools", "probe", "libstamps.so"))
for name in ("mi_dense_fwd", "mi_dense_bwd_data", "mi_dense_bwd_weight", "mi_absmax", "mi_set_gemm_mode", "mi_dense_bwd_weight_workspace_bytes"):
    f = getattr(L, name); f.restype, f.argtypes = _lib.SIGNATURES[name]
L.mi_gemm_stamps_read.restype = C.c_int32; L.mi_gemm_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
M, K, N = 65536, 1664, 512
X = torch.randn(M, K, device="cuda").relu_(); W = torch.randn(K, N, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda"); Y = torch.empty(M, N, device="cuda"); dY = torch.randn(M, N, device="cuda") * 1e-4
dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
ws = torch.empty(L.mi_dense_bwd_weight_workspace_bytes(M, N, K) + 256, dtype=torch.uint8, device="cuda")
def amax_of(t):
    v = torch.zeros(_lib.AMAX_SLOTS, device="cuda"); assert L.mi_absmax(p(t), t.numel(), p(v), st()) == 0; return v
aX, aW, adY = amax_of(X), amax_of(W), amax_of(dY)
L.mi_set_gemm_mode(1)
for mode in ("f16x2", "bf16x3"):
    ga = (lambda a, b_: _lib.GemmAmax(p(a), p(b_), None)) if mode == "f16x2" else (lambda a, b_: None)
    for op, fn, nk in (("fwd", lambda: L.mi_dense_fwd(p(X), K, p(W), p(b), p(Y), N, M, N, K, 1, 1.0, 123, ga(aX, aW), st()), 52),
                       ("dgrad", lambda: L.mi_dense_bwd_data(p(dY), N, p(W), None, K, p(dX), K, M, N, K, 1.0, 1, ga(adY, aW), st()), 16),
                       ("wgrad", lambda: L.mi_dense_bwd_weight(p(X), K, p(dY), N, p(dW), p(db), M, N, K, p(ws), ws.numel(), ga(aX, adY), st()), 64)):
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); assert fn() == 0; e1.record(); torch.cuda.synchronize()
        buf = np.zeros(64 * 64 * 8, dtype=np.int64)
        assert L.mi_gemm_stamps_read(buf.ctypes.data, buf.nbytes) == 0
        d = buf.reshape(64, 64, 8)[:, 2:nk - 2, :5]      # skip the first tiles (cold) and the peeled last one
        ph = np.diff(d, axis=2)
        tot = np.diff(d[:, :, 0], axis=1)
        print("%-6s %-5s kernel %7.1f us | cycles per k-tile %6.0f = main %6.0f + barrier %4.0f + frag reads %4.0f + barrier %4.0f + loop %4.0f"
              % (mode, op, e0.elapsed_time(e1) * 1e3, tot.mean(), ph[..., 0].mean(), ph[..., 1].mean(), ph[..., 2].mean(),
                 ph[..., 3].mean(), tot.mean() - ph.sum(-1).mean()))
