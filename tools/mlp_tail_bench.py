#!/usr/bin/env python3
"""The small layers of config 3's MLP (512 -> 256 -> 128 -> 1) and the planes path's helper launches, each timed alone
in the form the train step launches it (HIP events over many launches, B = 65536, activations half zero): forward with
planes / mask bits out, the data gradients with the mask read from the activation's planes or from the one-bit mask, the
logits layer's matrix-vector kernels, the weight gradients with their scale / fold launches.  What the step's `gemm_ms`
is made of besides the three layer-1 GEMMs.

usage: python tools/mlp_tail_bench.py [M]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch  # noqa: E402
from mi355x_rec import _lib  # noqa: E402

if os.environ.get("MI_TUNING_LIB"):          # the tools' build (make -C csrc tuning): reads the MI_* tuning switches
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libmi355x_rec_tuning.so")
lib = _lib.load()
st = lambda: _lib.cur_stream()
chk = _lib.check


class PB:
    def __init__(self, rows, K):
        self.data = torch.zeros(int(lib.mi_planes_bytes(rows, K)), dtype=torch.uint8, device="cuda")
        self.exp = torch.zeros(rows, dtype=torch.int32, device="cuda")
        self.s = _lib.Planes(self.data.data_ptr(), self.exp.data_ptr(), 64 * rows)
        self.ref = C.byref(self.s)


def split(x, transpose=False):
    rows, K = (x.shape[1], x.shape[0]) if transpose else x.shape
    pb = PB(rows, K)
    chk(lib.mi_split_rows(x.data_ptr(), x.shape[1], rows, K, 1 if transpose else 0, pb.ref, None, st()), "split")
    return pb


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    dims = [512, 256, 128]
    keep = 0.9
    rows = []
    acts = [torch.relu(torch.randn(M, d, device="cuda", generator=g)) for d in dims]        # half zero, like a relu layer's output
    actp = [split(a) for a in acts]
    bits = []
    for a in acts:
        w = (a > 0).view(M, -1, 32).to(torch.int64)
        bits.append((w << torch.arange(32, device="cuda")).sum(2).to(torch.int32).contiguous())
    Ws = [torch.randn(dims[i], dims[i + 1], device="cuda", generator=g) / dims[i] ** 0.5 for i in range(2)]
    amax = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
    # (the first second of a fresh process runs slow whatever it runs — clocks ramp: the second thing timed cost 70 us, the
    # same launch 58 us later on; a second of matrix work first)
    wa = torch.randn(8192, 8192, device="cuda")
    for _ in range(40):
        wa @ wa
    torch.cuda.synchronize()
    del wa
    # ---- forward: layer 2 (512 -> 256, planes + bits out), layer 3 (256 -> 128, fp32 + bits out), logits (gemv)
    for i, (K, N) in enumerate(((512, 256), (256, 128))):
        wt = split(Ws[i], transpose=True)
        b = torch.zeros(N, device="cuda")
        yp = PB(M, N)
        Y = torch.empty(M, N, device="cuda")
        mb = torch.empty(M, N // 32, dtype=torch.int32, device="cuda")
        last = i == 1
        for label, mbp in (("no bits", None), ("+ mask bits", mb)):
            t = timeit(lambda: chk(lib.mi_dense_fwd_planes(actp[i].ref, wt.ref, b.data_ptr(), Y.data_ptr() if last else None, N,
                                                           None if last else yp.ref, M, N, K, 1, keep, 7, amax.data_ptr(),
                                                           None if mbp is None else mbp.data_ptr(), N // 32, st()), "fwd"))
            rows.append(("forward %d -> %d (%s out), %s" % (K, N, "fp32" if last else "planes", label), t))
    w4 = torch.randn(128, device="cuda", generator=g)
    y1 = torch.empty(M, device="cuda")
    rows.append(("logits layer forward (gemv)", timeit(lambda: chk(lib.mi_dense_fwd(acts[2].data_ptr(), 128, w4.data_ptr(), None, y1.data_ptr(), 1, M, 1, 128, 0, 1.0, 0, None, st()), "gemv"))))
    # ---- backward
    dl = torch.randn(M, device="cuda", generator=g) * 1e-5
    dyp2 = PB(M, 128)
    for label, xa, mb in (("mask from fp32 activation", acts[2], None), ("mask bits", None, bits[2])):
        t = timeit(lambda: chk(lib.mi_dense_bwd_data_vec_planes(dl.data_ptr(), 1, w4.data_ptr(), None if xa is None else xa.data_ptr(), 128, keep,
                                                                None, 0, dyp2.ref, M, 128, amax.data_ptr(),
                                                                None if mb is None else mb.data_ptr(), 4, st()), "vec"))
        rows.append(("logits layer data gradient -> planes, " + label, t))
    dW4 = torch.empty(128, device="cuda"); db4 = torch.empty(1, device="cuda")
    ws = torch.empty(int(lib.mi_dense_bwd_weight_workspace_bytes(M, 1, 128)) + 256, dtype=torch.uint8, device="cuda")
    rows.append(("logits layer weight gradient (gemv + fold)", timeit(lambda: chk(lib.mi_dense_bwd_weight(
        acts[2].data_ptr(), 128, dl.data_ptr(), 1, dW4.data_ptr(), db4.data_ptr(), M, 1, 128, ws.data_ptr(), ws.numel(), None, st()), "wg"))))
    # the fused form of the five launches around the head (round 4)
    lin = torch.randn(M, device="cuda", generator=g) * 0.1; fm = torch.randn(M, device="cuda", generator=g) * 0.1
    yl = (torch.rand(M, device="cuda", generator=g) < 0.25).to(torch.uint8)
    lb = torch.zeros(1, device="cuda"); b4 = torch.zeros(1, device="cuda")
    logits = torch.empty(M, device="cuda"); loss = torch.empty(1, device="cuda"); dsum = torch.empty(1, device="cuda")
    hws = torch.empty(int(lib.mi_head_workspace_bytes(M)) + 256, dtype=torch.uint8, device="cuda")
    rows.append(("head (logits sum, loss, d_logit)", timeit(lambda: chk(lib.mi_sigmoid_ce_head(
        lin.data_ptr(), lb.data_ptr(), fm.data_ptr(), y1.data_ptr(), yl.data_ptr(), M, 1.0 / M, logits.data_ptr(), loss.data_ptr(), dl.data_ptr(),
        dsum.data_ptr(), hws.data_ptr(), hws.numel(), st()), "head"))))
    tws = torch.empty(int(lib.mi_logits_head_fused_workspace_bytes(M, 128)) + 256, dtype=torch.uint8, device="cuda")
    rows.append(("FUSED logits layer + head + its backward (mi_logits_head_fused)", timeit(lambda: chk(lib.mi_logits_head_fused(
        acts[2].data_ptr(), 128, w4.data_ptr(), b4.data_ptr(), lin.data_ptr(), lb.data_ptr(), fm.data_ptr(), yl.data_ptr(), M, 128, 1.0 / M,
        bits[2].data_ptr(), 4, keep, y1.data_ptr(), logits.data_ptr(), loss.data_ptr(), dl.data_ptr(), dsum.data_ptr(), dW4.data_ptr(), db4.data_ptr(),
        dyp2.ref, None, 128, amax.data_ptr(), tws.data_ptr(), tws.numel(), st()), "tail"))))
    # ... and with the last hidden layer's GEMM in front of it, in one launch (+ a fold)
    wt3 = split(Ws[1], transpose=True); b3 = torch.zeros(128, device="cuda")
    pws = torch.empty(int(lib.mi_hidden_logits_head_fused_workspace_bytes(M, 128)) + 256, dtype=torch.uint8, device="cuda")
    rows.append(("FUSED layer 256 -> 128 + logits layer + head + backward (mi_hidden_logits_head_fused)", timeit(lambda: chk(
        lib.mi_hidden_logits_head_fused(actp[1].ref, wt3.ref, b3.data_ptr(), M, 128, 256, 1, keep, 7, w4.data_ptr(), b4.data_ptr(), lin.data_ptr(),
                                        lb.data_ptr(), fm.data_ptr(), yl.data_ptr(), 1.0 / M, y1.data_ptr(), logits.data_ptr(), loss.data_ptr(),
                                        dl.data_ptr(), dsum.data_ptr(), dW4.data_ptr(), db4.data_ptr(), dyp2.ref, amax.data_ptr(),
                                        pws.data_ptr(), pws.numel(), st()), "top"))))
    dl = torch.randn(M, device="cuda", generator=g) * 1e-5
    for i, (K, N) in ((1, (256, 128)), (0, (512, 256))):            # data gradient of layer i+2: dX [M, K] = dY [M, N] W^T, mask of the K-wide activation
        dY = torch.randn(M, N, device="cuda", generator=g) * 1e-5
        dyp, wp = split(dY), split(Ws[i])
        dxp = PB(M, K)
        for label, xa, mb in (("mask from planes", actp[i], None), ("mask bits", None, bits[i])):
            t = timeit(lambda: chk(lib.mi_dense_bwd_data_planes(dyp.ref, wp.ref, None if xa is None else xa.ref, None, K, dxp.ref, M, N, K, keep,
                                                                amax.data_ptr(), None if mb is None else mb.data_ptr(), K // 32, st()), "dg"))
            rows.append(("data gradient %d <- %d, %s" % (K, N, label), t))
        ax = torch.zeros(_lib.AMAX_SLOTS, device="cuda"); ay = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
        lib.mi_absmax(acts[i].data_ptr(), acts[i].numel(), ax.data_ptr(), st()); lib.mi_absmax(dY.data_ptr(), dY.numel(), ay.data_ptr(), st())
        ga = _lib.GemmAmax(ax.data_ptr(), ay.data_ptr(), None)
        wsz = torch.empty(int(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, N, K)) + 256, dtype=torch.uint8, device="cuda")
        dW = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
        t = timeit(lambda: chk(lib.mi_dense_bwd_weight_planes(actp[i].ref, dyp.ref, dW.data_ptr(), db.data_ptr(), M, N, K, wsz.data_ptr(), wsz.numel(),
                                                              C.byref(ga), st()), "wgrad"))
        rows.append(("weight gradient %d x %d (scales + GEMM + fold)" % (K, N), t))
    # layer 1's weight gradient alone, for its scale / fold overhead
    X = torch.relu(torch.randn(M, 1664, device="cuda", generator=g)); dY = torch.randn(M, 512, device="cuda", generator=g) * 1e-5
    xp, dyp = split(X), split(dY)
    ax = torch.zeros(_lib.AMAX_SLOTS, device="cuda"); ay = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
    lib.mi_absmax(X.data_ptr(), X.numel(), ax.data_ptr(), st()); lib.mi_absmax(dY.data_ptr(), dY.numel(), ay.data_ptr(), st())
    ga = _lib.GemmAmax(ax.data_ptr(), ay.data_ptr(), None)
    wsz = torch.empty(int(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, 512, 1664)) + 256, dtype=torch.uint8, device="cuda")
    dW = torch.empty(1664, 512, device="cuda"); db = torch.empty(512, device="cuda")
    rows.append(("weight gradient 1664 x 512 (scales + GEMM + fold)", timeit(lambda: chk(lib.mi_dense_bwd_weight_planes(
        xp.ref, dyp.ref, dW.data_ptr(), db.data_ptr(), M, 512, 1664, wsz.data_ptr(), wsz.numel(), C.byref(ga), st()), "wgrad"))))
    for name, t in rows:
        print("%-62s %8.1f us" % (name, t))
    if os.environ.get("MI_TUNING_LIB"):
        print("-- weight-gradient plan variants (tuning build) --")
        shapes = ((512, 1664), (256, 512), (128, 256))
        ops = {}
        for N, K in shapes:
            X = torch.relu(torch.randn(M, K, device="cuda", generator=g)); dY = torch.randn(M, N, device="cuda", generator=g) * 1e-5
            ax = torch.zeros(_lib.AMAX_SLOTS, device="cuda"); ay = torch.zeros(_lib.AMAX_SLOTS, device="cuda")
            lib.mi_absmax(X.data_ptr(), X.numel(), ax.data_ptr(), st()); lib.mi_absmax(dY.data_ptr(), dY.numel(), ay.data_ptr(), st())
            ops[(N, K)] = (split(X), split(dY), ax, ay, torch.empty(K, N, device="cuda"), torch.empty(N, device="cuda"))
            del X, dY
        for env in ({}, {"MI_WGRAD_TM_N256": "2"}, {"MI_WGRAD_NBUF_N128": "6"}, {"MI_WGRAD_MIN_KSTEPS": "16"},
                    {"MI_WGRAD_NBUF_N128": "6", "MI_WGRAD_MIN_KSTEPS": "16"}, {"MI_WGRAD_TM_N256": "2", "MI_WGRAD_MIN_KSTEPS": "16"},
                    {"MI_WGRAD_MIN_KSTEPS": "64"}, {"MI_WGRAD_TM_N256": "2", "MI_WGRAD_MIN_KSTEPS": "64"},
                    {"MI_WGRAD_N256_AS_128": "1"}, {"MI_WGRAD_N256_AS_128": "1", "MI_WGRAD_MIN_KSTEPS": "64"},
                    {"MI_WGRAD_N256_AS_128": "1", "MI_WGRAD_MIN_KSTEPS": "128"}):
            for k_ in ("MI_WGRAD_TM_N256", "MI_WGRAD_NBUF_N128", "MI_WGRAD_MIN_KSTEPS", "MI_WGRAD_N256_AS_128"):
                os.environ.pop(k_, None)
            os.environ.update(env)
            line = []
            for N, K in shapes:
                xp, dyp, ax, ay, dW, db = ops[(N, K)]
                ga = _lib.GemmAmax(ax.data_ptr(), ay.data_ptr(), None)
                wsz = torch.empty(int(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, N, K)) + 256, dtype=torch.uint8, device="cuda")
                t = timeit(lambda: chk(lib.mi_dense_bwd_weight_planes(xp.ref, dyp.ref, dW.data_ptr(), db.data_ptr(), M, N, K, wsz.data_ptr(), wsz.numel(),
                                                                      C.byref(ga), st()), "wgrad"))
                line.append("%d x %d: %6.1f us" % (K, N, t))
            print("%-60s %s" % (env or "default", "   ".join(line)))


if __name__ == "__main__":
    main()
