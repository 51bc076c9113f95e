#!/usr/bin/env python3
"""Gather-kernel microbench: HBM read rate of mi_embed_fm_linear_fwd (read-only form) vs table size /
outputs requested.  HIP events, median of 20."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec import _lib
L = _lib.load()
st = lambda: _lib.cur_stream()
p = lambda t: None if t is None else t.data_ptr()
B, F, E = 65536, 26, 64
g = torch.Generator(device="cuda"); g.manual_seed(0)


def run(V, want_lin=True, want_sumv=True, want_concat=False, sorted_ids=False, want_amax=False):
    R = V * F
    table = torch.randn(R, E, device="cuda", generator=g)
    lin_w = torch.randn(R, device="cuda", generator=g)
    off = (torch.arange(F, device="cuda", dtype=torch.int64) * V)
    ids = torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g)
    if sorted_ids:
        ids = ids.sort(0).values.contiguous()
    sumv = torch.empty(B, E, device="cuda") if want_sumv else None
    fm = torch.empty(B, device="cuda") if want_sumv else None
    lin = torch.empty(B, device="cuda") if want_lin else None
    concat = torch.empty(B, F * E, device="cuda") if want_concat else None
    amax = torch.zeros(64, device="cuda") if want_amax else None
    fn = lambda: L.mi_embed_fm_linear_fwd(p(table), p(lin_w) if want_lin else None, p(off), p(ids), B, F, E, p(concat),
                                          F * E, p(sumv), p(fm), p(lin), p(amax), 1, 0, st())
    assert fn() == 0, L.mi_last_error()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort()
    t = ts[len(ts) // 2]
    print("V=%8d table %6.2f GB lin=%d sumv=%d concat=%d sorted=%d amax=%d : %7.1f us  %6.0f GB/s (rows only)" % (
        V, R * E * 4 / 1e9, want_lin, want_sumv, want_concat, sorted_ids, want_amax, t * 1e3, B * F * E * 4 / t / 1e6))


for V in (10_000, 100_000, 1_000_000, 4_000_000):
    run(V)
run(1_000_000, want_lin=False)
run(1_000_000, want_lin=False, want_sumv=False) if False else None
run(1_000_000, want_lin=True, want_sumv=True, want_concat=True)
run(1_000_000, sorted_ids=True)
run(1_000_000, want_amax=True)
run(1_000_000, want_lin=False, want_amax=True)
