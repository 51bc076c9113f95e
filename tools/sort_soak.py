#!/usr/bin/env python3
"""Soak of the one-launch-per-pass sort (mi_sort_unique_fields, beside = 0: workgroups of a field wait for each other inside a
launch): N calls on fresh random ids, each compared with the 14-launch form (beside = 1) bit for bit, with and without a
bandwidth-heavy copy loop on another stream taking wave slots and HBM away.  usage: python tools/sort_soak.py [N]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
from mi355x_rec import _lib
lib = _lib.load(); st = lambda: _lib.cur_stream()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator(device="cuda"); g.manual_seed(0)
i32 = dict(dtype=torch.int32, device="cuda")
bad = 0
for B, F, V in ((65536, 26, 1_000_000), (8192, 64, 3_000_000), (4096, 5, 100)):
    off = torch.arange(F, device="cuda", dtype=torch.int64) * V
    n = B * F
    outs = lambda: (torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(n + 1, **i32), torch.empty(1, **i32))
    wsA = torch.empty(int(lib.mi_sort_unique_fields_workspace_bytes(B, F)) + 256, dtype=torch.uint8, device="cuda")
    wsB = torch.empty_like(wsA)
    src = torch.empty(1 << 27, device="cuda"); dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    for noisy in (False, True):
        for it in range(N if B == 65536 else N // 4):
            ids = torch.randint(0, V, (B, F), generator=g, **i32)
            if it % 7 == 0:
                ids[:, it % F] = ids[0, it % F]                      # a field with one id: a segment of B duplicates
            a, b = outs(), outs()
            if noisy:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    dst.copy_(src); src.copy_(dst)
            rc = lib.mi_sort_unique_fields(ids.data_ptr(), off.data_ptr(), B, F, V, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), wsA.data_ptr(), wsA.numel(), 0, st())
            rc |= lib.mi_sort_unique_fields(ids.data_ptr(), off.data_ptr(), B, F, V, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), wsB.data_ptr(), wsB.numel(), 1, st())
            U = int(b[3].item())
            ok = rc == 0 and int(a[3].item()) == U and torch.equal(a[0], b[0]) and torch.equal(a[1][:U], b[1][:U]) and torch.equal(a[2][:U + 1], b[2][:U + 1])
            bad += 0 if ok else 1
        torch.cuda.synchronize()
        print("B=%d F=%d V=%d %s: %d calls, %d mismatches so far" % (B, F, V, "beside a copy loop" if noisy else "alone", N if B == 65536 else N // 4, bad), flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED: %d" % bad)
sys.exit(0 if bad == 0 else 1)
