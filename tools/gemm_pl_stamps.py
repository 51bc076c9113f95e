#!/usr/bin/env python3
"""Where a k-step of the planes GEMM (csrc/gemm_pl.hip) goes: in-kernel clock64() stamps of the layer-1 forward.
Needs a library built with -DMI_PL_STAMPS:
  make -C recommender-tensorflow_amd/csrc OBJDIR=/tmp/bpl OUT=../../tools/probe/libplstamps.so \
       CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DMI_PL_STAMPS"
Per k-step and wave group (group 0 = waves 0-3, group 1 = waves 4-7, one barrier behind):
  L: issue DMA + fragment reads | wait LDS | wait own DMA share | barrier | C: 24 MFMAs | barrier"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import numpy as np, torch
from mi355x_rec import _lib
L = C.CDLL(os.path.join(ROOT, "tools", "probe", "libplstamps.so"))
for name in ("mi_dense_fwd_planes", "mi_split_rows", "mi_planes_bytes"):
    f = getattr(L, name); f.restype, f.argtypes = _lib.SIGNATURES[name]
L.mi_pl_stamps_read.restype = C.c_int32; L.mi_pl_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
st = lambda: torch.cuda.current_stream().cuda_stream
class PB:
    def __init__(self, rows, K):
        self.data = torch.zeros(int(L.mi_planes_bytes(rows, K)), dtype=torch.uint8, device="cuda")
        self.exp = torch.zeros(rows, dtype=torch.int32, device="cuda")
        self.s = _lib.Planes(self.data.data_ptr(), self.exp.data_ptr(), 64 * rows); self.ref = C.byref(self.s)
def split(x, transpose=False):
    rows, K = (x.shape[1], x.shape[0]) if transpose else x.shape
    pb = PB(rows, K)
    assert L.mi_split_rows(x.data_ptr(), x.shape[1], rows, K, 1 if transpose else 0, pb.ref, None, st()) == 0
    return pb
M, K, N = 65536, 1664, 512
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.randn(M, K, device="cuda", generator=g).relu_(); W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
b = torch.zeros(N, device="cuda")
xp, wt, yp = split(X), split(W, transpose=True), PB(M, N)
for _ in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); assert L.mi_dense_fwd_planes(xp.ref, wt.ref, b.data_ptr(), None, N, yp.ref, M, N, K, 1, 0.9, 7, None, None, 0, st()) == 0; e1.record()
    torch.cuda.synchronize()
buf = np.zeros(32 * 2 * 128 * 8, dtype=np.int64)
assert L.mi_pl_stamps_read(buf.ctypes.data, buf.nbytes) == 0
d = buf.reshape(32, 2, 128, 8)[:, :, 4:100, :7].astype(np.float64)     # k-steps 4..99 of 104
ph = np.diff(d, axis=3)                                                 # 6 phases
tot = np.diff(d[:, :, :, 0], axis=2)
names = ["issue DMA + frag reads", "wait LDS (lgkmcnt)", "wait own DMA share (vmcnt)", "barrier 1", "C: 24 MFMAs", "barrier 2"]
print("layer-1 forward 512 x 1664, planes out: kernel %.1f us; clock64 ticks per k-step %.0f" % (e0.elapsed_time(e1) * 1e3, tot.mean()))
for grp in (0, 1):
    print("  group %d: " % grp + " | ".join("%s %.0f" % (n, ph[:, grp, :, i].mean()) for i, n in enumerate(names)) +
          " | loop-back %.0f" % (tot[:, grp].mean() - ph[:, grp].sum(-1).mean()))
