#!/usr/bin/env python3
"""Times the planes GEMMs (csrc/gemm_pl.hip) shape by shape at config-3 sizes, HIP events over many
launches on random data.  TF/s are fp32-equivalent (2 M N K / t); the executed f16 MFMA rate is 3x that."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch  # noqa: E402
from mi355x_rec import _lib  # noqa: E402

lib = _lib.load()
st = lambda: _lib.cur_stream()


class PB:
    def __init__(self, rows, K):
        self.data = torch.zeros(int(lib.mi_planes_bytes(rows, K)), dtype=torch.uint8, device="cuda")
        self.exp = torch.zeros(rows, dtype=torch.int32, device="cuda")
        self.s = _lib.Planes(self.data.data_ptr(), self.exp.data_ptr(), 64 * rows)
        self.ref = C.byref(self.s)


def split(x, transpose=False):
    rows, K = (x.shape[1], x.shape[0]) if transpose else x.shape
    pb = PB(rows, K)
    _lib.check(lib.mi_split_rows(x.data_ptr(), x.shape[1], rows, K, 1 if transpose else 0, pb.ref, None, st()), "split")
    return pb


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    M = 65536
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for name, N, K in (("fwd L1", 512, 1664), ("fwd L2", 256, 512), ("fwd L3", 128, 256),
                       ("dgrad L1 (fp32 out)", 1664, 512), ("dgrad L2", 512, 256), ("dgrad L3", 256, 128))[:int(os.environ.get("PL_BENCH_FWD", "6"))]:
        X = torch.randn(M, K, device="cuda", generator=g)
        W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
        b = torch.zeros(N, device="cuda")
        xp, wt = split(X), split(W, transpose=True)
        Y = torch.empty(M, N, device="cuda")
        yp = PB(M, N) if N <= 512 else None
        flops = 2.0 * M * N * K
        if N > 512:
            for tile in ("1", "2", "4"):
                os.environ["MI_PL_TILE"] = tile
                t = timeit(lambda: _lib.check(lib.mi_dense_fwd_planes(xp.ref, wt.ref, b.data_ptr(), Y.data_ptr(), N, None, M, N, K, 1, 0.9,
                                                                      7, None, None, 0, st()), "fwd"))
                print("%-22s N=%4d K=%4d column tile %4d  %7.1f us  %6.1f TF/s fp32-equiv" % (name, N, K, 128 * int(tile), t * 1e3, flops / t / 1e9))
            os.environ.pop("MI_PL_TILE")
        for label, y, ypl in (("fp32+planes", Y, yp), ("fp32 only", Y, None), ("planes only", None, yp)):
            if (y is None and ypl is None) or (label != "fp32 only" and ypl is None):
                continue
            t = timeit(lambda: _lib.check(lib.mi_dense_fwd_planes(xp.ref, wt.ref, b.data_ptr(), None if y is None else y.data_ptr(), N,
                                                                  None if ypl is None else ypl.ref, M, N, K, 1, 0.9, 7, None, None, 0, st()), "fwd"))
            print("%-22s N=%4d K=%4d %-12s %7.1f us  %6.1f TF/s fp32-equiv" % (name, N, K, label, t * 1e3, flops / t / 1e9))
        del X, W, xp, wt, Y, yp
    # weight gradients from planes
    for name, N, K in (("wgrad L1", 512, 1664), ("wgrad L2", 256, 512), ("wgrad L3", 128, 256)):
        X = torch.randn(M, K, device="cuda", generator=g)
        dY = torch.randn(M, N, device="cuda", generator=g) * 1e-4
        xp, dyp = split(X), split(dY)
        ax = torch.zeros(_lib.AMAX_SLOTS, device="cuda"); ay = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
        lib.mi_absmax(X.data_ptr(), X.numel(), ax.data_ptr(), st()); lib.mi_absmax(dY.data_ptr(), dY.numel(), ay.data_ptr(), st())
        ga = _lib.GemmAmax(ax.data_ptr(), ay.data_ptr(), None)
        ws = torch.empty(int(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, N, K)) + 256, dtype=torch.uint8, device="cuda")
        dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
        flops = 2.0 * M * N * K
        t = timeit(lambda: _lib.check(lib.mi_dense_bwd_weight_planes(xp.ref, dyp.ref, dW.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(),
                                                                     ws.numel(), C.byref(ga), st()), "wgrad"))
        print("%-22s N=%4d K=%4d planes       %7.1f us  %6.1f TF/s fp32-equiv" % (name, N, K, t * 1e3, flops / t / 1e9))
        t = timeit(lambda: _lib.check(lib.mi_dense_bwd_weight(X.data_ptr(), K, dY.data_ptr(), N, dW.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(),
                                                              ws.numel(), C.byref(ga), st()), "wgrad"))
        print("%-22s N=%4d K=%4d fp32 operands %6.1f us  %6.1f TF/s fp32-equiv" % (name, N, K, t * 1e3, flops / t / 1e9))
        del X, dY, xp, dyp
    # the weight splits of one step
    ws = [torch.randn(k, n, device="cuda", generator=g) for k, n in ((1664, 512), (512, 256), (256, 128))]
    outs = [(PB(w.shape[1], w.shape[0]), PB(w.shape[0], w.shape[1])) for w in ws]

    def all_splits():
        for w, (pt, pn) in zip(ws, outs):
            lib.mi_split_rows(w.data_ptr(), w.shape[1], w.shape[1], w.shape[0], 1, pt.ref, None, st())
            lib.mi_split_rows(w.data_ptr(), w.shape[1], w.shape[0], w.shape[1], 0, pn.ref, None, st())
    print("weight splits of a step (3 layers x 2 orientations): %.1f us" % (timeit(all_splits) * 1e3))


if __name__ == "__main__":
    main()
