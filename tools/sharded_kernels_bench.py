"""GPU microbench of the kernels only the row-sharded (multi-GPU) step runs, one line per kernel with its own byte
count and GB/s — so that the next `mi_route_requests`-class cost (19 ms of serialised atomics, found only when a
whole-step bench ran: DESIGN section 4) shows up by itself.  One GPU, no process group: the kernels are driven through
the C ABI with the buffers one rank of an N-rank step would hand them.

    python tools/sharded_kernels_bench.py [--world 8] [--chunks 1] [--batch 65536] [--fields 26] [--vocab 1000000] [--emb 64]

Rank 0's view of a step at config-3 sizes: B x F entries, owners = row % world; all requests are kept (what a one-rank
run of bench.py --force-shard sees is --world 1).  Bytes are the algorithmic ones of each kernel (what it must read and
write once), the same accounting as bench.py's rooflines."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
from mi355x_rec import _lib  # noqa: E402


def timed(f, reps=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=1000000)
    ap.add_argument("--emb", type=int, default=64)
    a = ap.parse_args()
    lib = _lib.load()
    st = _lib.cur_stream
    dev = "cuda"
    W, C, B, F, V, E = a.world, a.chunks, a.batch, a.fields, a.vocab, a.emb
    n = B * F
    R = F * V
    Rl = (R + W - 1) // W
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    i32 = torch.int32
    ids = torch.randint(0, V, (B, F), device=dev, dtype=i32, generator=g)
    field_off = (torch.arange(F, device=dev, dtype=torch.int64) * V)
    rows = torch.empty(n, dtype=i32, device=dev)
    key = torch.empty(n, dtype=i32, device=dev)
    se, uq = torch.empty(n, dtype=i32, device=dev), torch.empty(n, dtype=i32, device=dev)
    sg, nu = torch.empty(n + 1, dtype=i32, device=dev), torch.empty(1, dtype=i32, device=dev)
    ws = torch.empty(int(lib.mi_sort_unique_workspace_bytes(n)) + 256, dtype=torch.uint8, device=dev)
    send_rows, slot = torch.empty(n, dtype=i32, device=dev), torch.empty(n, dtype=i32, device=dev)
    counts = torch.empty(C * W, dtype=i32, device=dev)
    out = []

    def line(name, us, nbytes, note=""):
        out.append((name, us, nbytes))
        print("%-34s %8.1f us  %8.1f MB  %7.0f GB/s  %s" % (name, us, nbytes / 1e6, nbytes / us / 1e3, note))

    ck = lambda rc: _lib.check(rc, "sharded_kernels_bench")
    line("mi_global_rows", timed(lambda: ck(lib.mi_global_rows(ids.data_ptr(), field_off.data_ptr(), B, F, rows.data_ptr(), st()))), 8 * n)
    epc = n // C if C > 1 else 0
    line("mi_shard_keys", timed(lambda: ck(lib.mi_shard_keys(rows.data_ptr(), n, W, epc, Rl, 0, key.data_ptr(), st()))), 8 * n)
    key_range = C * W * Rl
    f_sort = lambda: ck(lib.mi_sort_unique_rows(key.data_ptr(), n, key_range, se.data_ptr(), uq.data_ptr(), sg.data_ptr(), nu.data_ptr(),
                                                ws.data_ptr(), ws.numel(), st()))
    bits = (key_range - 1).bit_length()
    passes = (bits + 8) // 9
    line("mi_sort_unique_rows (route keys)", timed(f_sort), n * (8 * 2 * passes + 16), "%d-bit keys, %d passes; bytes = key+payload in and out per pass" % (bits, passes))
    U = int(nu.item())
    line("mi_route_requests", timed(lambda: ck(lib.mi_route_requests(uq.data_ptr(), nu.data_ptr(), n, Rl, C * W, send_rows.data_ptr(),
                                                                      counts.data_ptr(), st()))), 8 * U, "U = %d distinct requests of %d entries" % (U, n))
    line("mi_segment_slots", timed(lambda: ck(lib.mi_segment_slots(sg.data_ptr(), se.data_ptr(), nu.data_ptr(), n, slot.data_ptr(), st()))), 8 * n + 4 * U)

    # owner side: this rank is asked for ~U rows of its R / W (every rank's requests to it; uniform ids: as many as it sends)
    nr = U
    recv_ids = torch.randint(0, Rl, (nr,), device=dev, dtype=i32, generator=g)
    ose, ouq = torch.empty(nr, dtype=i32, device=dev), torch.empty(nr, dtype=i32, device=dev)
    osg = torch.empty(nr + 1, dtype=i32, device=dev)
    bits = (Rl - 1).bit_length()
    passes = (bits + 8) // 9
    line("mi_sort_unique_rows (owner)", timed(lambda: ck(lib.mi_sort_unique_rows(recv_ids.data_ptr(), nr, Rl, ose.data_ptr(), ouq.data_ptr(),
                                                                                 osg.data_ptr(), nu.data_ptr(), ws.data_ptr(), ws.numel(), st()))),
         nr * (8 * 2 * passes + 16), "%d-bit keys, %d passes" % (bits, passes))
    n_own = int(nu.item())
    table = torch.randn(Rl, E, device=dev)
    lin_state = torch.zeros(Rl, 4, device=dev)
    own_rows, own_lin = torch.empty(nr, E, device=dev), torch.empty(nr, device=dev)
    line("mi_gather_rows", timed(lambda: ck(lib.mi_gather_rows(table.data_ptr(), lin_state.data_ptr(), recv_ids.data_ptr(), nr, E, own_rows.data_ptr(),
                                                               own_lin.data_ptr(), 4, 0, 0, st()))), nr * (8 * E + 4 + 16 + 4),
         "rows read + written, the 16-byte wide record read")

    # requester side: per-request gradient sums from d_concat (the chunk's entries)
    Bc = B // C
    d_concat = torch.randn(Bc, F * E, device=dev)
    sumv, dlogit = torch.randn(Bc, E, device=dev), torch.randn(Bc, device=dev)
    d_rows, d_lin = torch.empty(U, E, device=dev), torch.empty(U, device=dev)
    # the segments of chunk 0: requests [0, u0)
    u0 = int(counts.view(C, W)[0].sum().item())
    for fm in (True, False):
        f = lambda: ck(lib.mi_entry_grads_segsum(own_rows.data_ptr() if fm else None, sg.data_ptr(), se.data_ptr(), 0, u0,
                                                 d_concat.data_ptr(), F * E, sumv.data_ptr() if fm else None, dlogit.data_ptr() if fm else None,
                                                 dlogit.data_ptr(), 0, F, E, d_rows.data_ptr(), d_lin.data_ptr(), 0, 0, 0, st()))
        ne = Bc * F
        line("mi_entry_grads_segsum (%s)" % ("DeepFM" if fm else "no FM term"), timed(f),
             ne * (4 * E + 4 + 4) + u0 * (4 * E + 4 + (4 * E if fm else 0)) + (ne * 4 * E if fm else 0),
             "entries' d_concat slices%s in, one row per request out" % (" + sumv + the row" if fm else ""))

    # owner side: the apply from received per-request gradients (Adam: w, m, v read and written + the gradient)
    print("(mi_sparse_apply on received gradients: timed by bench.py --force-shard's instrumented pass — it needs the "
          "engine's optimizer structs; %d rows of this owner's %d would be touched)" % (n_own, Rl))
    tot = sum(us for _, us, _ in out)
    print("sum of the above: %.0f us per step (the route sort and the owners' sort run AHEAD of the step when the next batch "
          "is announced: parallel._route_ahead / _own_ahead)" % tot)


if __name__ == "__main__":
    main()
