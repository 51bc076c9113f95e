#!/usr/bin/env python3
"""Fits t = a + b * (K / 16) for the planes GEMM at a fixed output shape: a is what a tile pays outside the
k loop (prologue fill, epilogue, stores), b the cost of one 16-deep k step.  HIP events, random data."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from gemm_pl_bench import PB, lib, split, st, timeit, _lib  # noqa: E402


def main():
    M = 65536
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for N, epi in ((512, 0), (512, 1), (256, 0), (128, 0), (1664, 1)):
        for K in (16, 64, 256, 512, 1024, 1664):
            X = torch.randn(M, K, device="cuda", generator=g)
            W = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
            b = torch.zeros(N, device="cuda")
            xp, wt = split(X), split(W, transpose=True)
            Y = torch.empty(M, N, device="cuda")
            yp = PB(M, N) if N <= 512 else None
            act = PB(M, N) if (epi and N <= 512) else None
            res = []
            for label, y, ypl in (("fp32+planes", Y, yp), ("fp32 only", Y, None), ("planes only", None, yp)):
                if (label != "fp32 only" and ypl is None):
                    res.append(float("nan")); continue
                if epi == 0:
                    fn = lambda: _lib.check(lib.mi_dense_fwd_planes(xp.ref, wt.ref, b.data_ptr(), None if y is None else y.data_ptr(), N,
                                                                    None if ypl is None else ypl.ref, M, N, K, 1, 0.9, 7, None, None, 0, st()), "fwd")
                else:
                    fn = lambda: _lib.check(lib.mi_dense_fwd_planes(xp.ref, wt.ref, b.data_ptr(), None if y is None else y.data_ptr(), N,
                                                                    None if ypl is None else ypl.ref, M, N, K, 0, 1.0, 0, None, None, 0, st()), "fwd")
                res.append(timeit(fn) * 1e3)
            print("N=%4d act=%d K=%4d  fp32+planes %7.1f  fp32 only %7.1f  planes only %7.1f us" % (N, 1 - epi, K, *res), flush=True)
            del X, W, xp, wt, Y, yp


if __name__ == "__main__":
    main()
