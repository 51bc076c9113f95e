#!/usr/bin/env python3
"""us per call of the two sort + unique entries at config 3's size (1.7 M keys).  With MI_TUNING_LIB=1 the tools' build is
loaded (make -C recommender-tensorflow_amd/csrc tuning) and MI_SORT_FUSED=0|1 / MI_SORT_BITS=7..10 choose the per-field form:
   MI_TUNING_LIB=1 MI_SORT_FUSED=0 python tools/sort_bench.py fields      # round 4's 14 launches
   MI_TUNING_LIB=1 MI_SORT_BITS=7 python tools/sort_bench.py fields       # fused, 3 passes of 7 bits
   python tools/sort_bench.py fields                                      # the shipped library: fused, 2 passes of 10 bits"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
from mi355x_rec import _lib
if os.environ.get("MI_TUNING_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libmi355x_rec_tuning.so")
lib = _lib.load(); st = lambda: _lib.cur_stream()
g = torch.Generator(device="cuda"); g.manual_seed(0)
i32 = dict(dtype=torch.int32, device="cuda")


def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


what = sys.argv[1] if len(sys.argv) > 1 else "both"
if what in ("rows", "both"):
    n = 65536 * 26
    for R in (26_000_000, 1 << 21, 1 << 20, 1 << 14):
        rows = torch.randint(0, R, (n,), generator=g, **i32)
        se, uq, sg, nu = torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(n + 1, **i32), torch.empty(1, **i32)
        ws = torch.empty(int(lib.mi_sort_unique_workspace_bytes(n)) + 256, dtype=torch.uint8, device="cuda")
        f = lambda: lib.mi_sort_unique_rows(rows.data_ptr(), n, R, se.data_ptr(), uq.data_ptr(), sg.data_ptr(), nu.data_ptr(), ws.data_ptr(), ws.numel(), st())
        bits = (R - 1).bit_length(); passes = (bits + 8) // 9; nb = (bits + passes - 1) // passes
        print("rows   R=%9d bits=%2d passes=%d x %d bits: %.1f us" % (R, bits, passes, nb, timeit(f)))
if what in ("fields", "both"):
    for B, F, V, zipf in ((65536, 26, 1_000_000, False), (65536, 26, 1_000_000, True), (16384, 40, 1_250_000, False), (131072, 26, 1_000_000, False)):
        if zipf:
            u = torch.rand(B, F, device="cuda", generator=g, dtype=torch.float64); s_ = 1.05
            hmax = (V ** (1 - s_) - 1) / (1 - s_)
            ids = (((u * hmax) * (1 - s_) + 1) ** (1 / (1 - s_)) - 1).clamp_(0, V - 1).to(torch.int32).contiguous()
        else:
            ids = torch.randint(0, V, (B, F), generator=g, **i32)
        off = (torch.arange(F, device="cuda", dtype=torch.int64) * V)
        n = B * F
        se, uq, sg, nu = torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(n + 1, **i32), torch.empty(1, **i32)
        ws = torch.empty(int(lib.mi_sort_unique_fields_workspace_bytes(B, F)) + 256, dtype=torch.uint8, device="cuda")
        f = lambda: lib.mi_sort_unique_fields(ids.data_ptr(), off.data_ptr(), B, F, V, se.data_ptr(), uq.data_ptr(), sg.data_ptr(), nu.data_ptr(), ws.data_ptr(), ws.numel(), int(os.environ.get("BESIDE", "0")), st())
        rc = f(); torch.cuda.synchronize()
        print("fields B=%6d F=%2d V=%8d %s: rc=%d unique=%d  %.1f us  (MI_SORT_FUSED=%s MI_SORT_BITS=%s)" % (
            B, F, V, "zipf   " if zipf else "uniform", rc, int(nu.item()), timeit(f), os.environ.get("MI_SORT_FUSED", "-"), os.environ.get("MI_SORT_BITS", "-")))
