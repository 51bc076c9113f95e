#!/usr/bin/env python3
"""Life of a workgroup of the planes GEMM (csrc/gemm_pl.hip), from in-kernel clock64() marks: kernel entry -> first k-step
(prologue fill) -> end of the k loop -> end of the epilogue (stores drained), for 32 workgroups spread over the grid, plus
the per-k-step phases.  Needs a library built with -DMI_PL_STAMPS:
  make -C recommender-tensorflow_amd/csrc OBJDIR=/tmp/bpl OUT=../../tools/probe/libplstamps.so \
       CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DMI_PL_STAMPS"
usage: gemm_pl_timeline.py [fwd1|dgrad1|fwd2|dgrad2|fwd3|dgrad3 ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import numpy as np, torch
from mi355x_rec import _lib
L = C.CDLL(os.path.join(ROOT, "tools", "probe", "libplstamps.so"))
for name in ("mi_dense_fwd_planes", "mi_dense_bwd_data_planes", "mi_split_rows", "mi_planes_bytes"):
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
M = 65536
g = torch.Generator(device="cuda"); g.manual_seed(0)
DIMS = {1: (1664, 512), 2: (512, 256), 3: (256, 128)}          # layer: (inputs, units)
def run(which):
    layer = int(which[-1]); fan, h = DIMS[layer]
    if which.startswith("fwd"):
        X = torch.randn(M, fan, device="cuda", generator=g).relu_(); W = torch.randn(fan, h, device="cuda", generator=g) / fan ** 0.5
        b = torch.zeros(h, device="cuda"); xp, wt, yp = split(X), split(W, transpose=True), PB(M, h)
        nk = fan // 16
        # (as in a training step: dropout, mask bits out, abs-max out, no fp32 copy)
        bits = torch.zeros(M, h // 32, dtype=torch.int32, device="cuda"); am = torch.zeros(64, device="cuda")
        fn = lambda: L.mi_dense_fwd_planes(xp.ref, wt.ref, b.data_ptr(), None, h, yp.ref, M, h, fan, 1, 0.9, 7, am.data_ptr(), bits.data_ptr(), h // 32, st())
    else:
        dY = torch.randn(M, h, device="cuda", generator=g) * 1e-4
        dY[torch.rand(M, h, device="cuda", generator=g) < 0.5] = 0
        W = torch.randn(fan, h, device="cuda", generator=g) / fan ** 0.5
        dyp, wp = split(dY), split(W)
        nk = h // 16
        if layer == 1:
            dX = torch.empty(M, fan, device="cuda")
            fn = lambda: L.mi_dense_bwd_data_planes(dyp.ref, wp.ref, None, dX.data_ptr(), fan, None, M, h, fan, 1.0, None, None, 0, st())
        else:
            dxp = PB(M, fan)
            bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (M, fan // 32), dtype=torch.int32, device="cuda", generator=g)
            am = torch.zeros(64, device="cuda")
            fn = lambda: L.mi_dense_bwd_data_planes(dyp.ref, wp.ref, None, None, fan, dxp.ref, M, h, fan, 0.9, am.data_ptr(), bits.data_ptr(), fan // 32, st())
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); assert fn() == 0; e1.record(); torch.cuda.synchronize()
    buf = np.zeros(32 * 2 * 128 * 8, dtype=np.int64)
    assert L.mi_pl_stamps_read(buf.ctypes.data, buf.nbytes) == 0
    d = buf.reshape(32, 2, 128, 8).astype(np.float64)
    marks = d[:, :, :6, 7]                                   # [wg][grp][entry, first k-step, loop end, epilogue end]
    ok = marks[:, 0, 0] > 0
    t0 = marks[ok][:, :, 0].min()
    print("%s: kernel %.1f us, nk = %d; clock64 ticks (100 MHz? no: shader clock)" % (which, e0.elapsed_time(e1) * 1e3, nk))
    kern_ticks = marks[ok][:, :, 3].max() - t0
    print("  sampled span (first entry -> last exit) %.0f ticks -> %.2f GHz if that is the kernel" % (kern_ticks, kern_ticks / (e0.elapsed_time(e1) * 1e6)))
    print("  wg  entry    prologue  k-loop   epilogue   (group 0; ticks)")
    for i in np.nonzero(ok)[0]:
        m0 = marks[i, 0]
        print("  %2d %8.0f %8.0f %8.0f %8.0f" % (i, m0[0] - t0, m0[1] - m0[0], m0[2] - m0[1], m0[3] - m0[2]))
    mm = marks[ok][:, 0]
    print("  mean: prologue %.0f, k-loop %.0f (%.0f per k-step), epilogue %.0f" % ((mm[:, 1] - mm[:, 0]).mean(), (mm[:, 2] - mm[:, 1]).mean(),
                                                                                (mm[:, 2] - mm[:, 1]).mean() / nk, (mm[:, 3] - mm[:, 2]).mean()))
    print("  epilogue: element loops %.0f, barrier + planes conversion + store issue %.0f, store drain %.0f" %
          ((mm[:, 4] - mm[:, 2]).mean(), (mm[:, 5] - mm[:, 4]).mean(), (mm[:, 3] - mm[:, 5]).mean()))
    lo, hi = min(2, nk - 1), min(nk, 100)
    ph = np.diff(d[ok][:, :, lo:hi, :7], axis=3)
    names = ["issue+frag reads", "wait LDS", "wait DMA", "barrier 1", "MFMAs", "barrier 2"]
    for grp in (0, 1):
        print("  group %d per k-step: " % grp + " | ".join("%s %.0f" % (n, ph[:, grp, :, i].mean()) for i, n in enumerate(names)))
for w in (sys.argv[1:] or ["dgrad1", "fwd1", "fwd2", "dgrad2"]):
    run(w)
