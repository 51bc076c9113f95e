#!/usr/bin/env python3
"""The other BASELINE.json shapes on one GPU (sanity at full size + step time; not bench lines):
  c5-rank : one rank's share of config 5 — 40 fields x 1.25 M rows (50 M rows, E=128: a 25.6 GB table,
            76.8 GB with Adam slots: byte offsets beyond 2^32), hidden [512,256,128], B = 16384
  c4      : Wide&Deep (linear_deep) shape — 26 categorical fields x 1 M rows + 13 numeric columns,
            B = 65536, Ftrl (wide) + Adagrad (deep), SUM loss, E = 64 and the CLI default E = 4"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec

def run(name, m, B, n_num, vocab, steps=20):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    m.init_variables(g, lin_scale=1e-3)
    F = len(vocab)
    bs = []
    for _ in range(4):
        ids = torch.stack([torch.randint(0, v, (B,), device="cuda", generator=g) for v in vocab], 1).to(torch.int32).contiguous()
        y = (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)
        x = torch.log1p(torch.empty(B, n_num, device="cuda").exponential_(generator=g)) if n_num else None
        bs.append((ids, y, x))
    losses = []
    for i in range(5):
        l, _ = m.train_step(*bs[i % 4]); losses.append(l.item())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        l, _ = m.train_step(*bs[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    losses.append(l.item())
    if os.environ.get("MI_BREAKDOWN"):
        # per-entry HIP-event times of a few more steps (each launch bracketed: ~10 us of bubble per launch)
        m.timers = {}
        for i in range(6):
            m.train_step(*bs[i % 4])
        torch.cuda.synchronize()
        tm, m.timers = m.timers, None
        rows = sorted(((sum(s_.elapsed_time(e_) for s_, e_ in ev) / 6, k_, len(ev) // 6) for k_, ev in tm.items()), reverse=True)
        for ms, k_, cnt in rows:
            print("    %-44s %7.3f ms/step  (%d launches)" % (k_, ms, cnt))
    assert all(map(lambda v: v == v and abs(v) < 1e9, losses)), losses
    print("%-8s B=%6d F=%2d E=%3d  %.3f ms/step  %.2f M examples/s  loss %.5f -> %.5f  mem %.1f GB" % (
        name, B, F, m.E, dt * 1e3, B / dt / 1e6, losses[0], losses[-1], torch.cuda.max_memory_allocated() / 1e9))

which = sys.argv[1:] or ["c5-rank", "c4"]
if "c5-rank" in which:
    vocab = [1_250_000] * 40
    m = DeepFM(vocab, embedding_size=128, hidden_units=[512, 256, 128], dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001))
    run("c5-rank", m, 16384, 0, vocab)
    del m; torch.cuda.empty_cache()
if "c4" in which:
    for E in (64, 4):
        vocab = [1_000_000] * 26
        m = DeepFM(vocab, n_numeric=13, numeric="raw", embedding_size=E, hidden_units=[512, 256, 128] if E == 64 else [16, 16], use_mf=False,
                   dropout=0.1, optimizer=OptimizerSpec("Adagrad", 0.05), linear_optimizer=OptimizerSpec("Ftrl", 0.1961), reduction="sum")
        run("c4 E=%d" % E, m, 65536, 13, vocab)
        del m; torch.cuda.empty_cache()
