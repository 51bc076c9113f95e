#!/usr/bin/env python3
"""Step latency of the reference's default configuration (trainers.deep_fm: E=4, hidden [16,16], B=32,
the 26 MovieLens fields) — launch-bound, not roofline-bound: reports us per train step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec
VOCAB = [2] * 19 + [1000, 2000, 50, 1000, 7, 8, 3]
for B in (32, 1024):
    m = DeepFM(VOCAB, embedding_size=4, hidden_units=[16, 16], dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001))
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    m.init_variables(g, lin_scale=1e-3)
    ids = torch.stack([torch.randint(0, v, (B,), device="cuda", generator=g) for v in VOCAB], 1).to(torch.int32).contiguous()
    y = (torch.rand(B, device="cuda", generator=g) < 0.3).to(torch.uint8)
    for name, step in (("eager (one launch per kernel)", m.train_step), ("hipGraph replay", m.graph_train_step)):
        for _ in range(20): step(ids, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300
        for _ in range(n): step(ids, y)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("B=%5d  %-30s %.1f us/step  (%.0f steps/s, %.0f examples/s)" % (B, name, dt / n * 1e6, n / dt, n * B / dt))
