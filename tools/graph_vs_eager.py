#!/usr/bin/env python3
"""Config-3 train steps, eager (plain sequence, no announced next batch) or as hipGraph replays — for a rocprofv3 kernel trace
of each (tools/trace_gaps.py on the result): where does the replay lose against the eager step?
usage: graph_vs_eager.py eager|graph [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
F, V, E, B = 26, 1_000_000, 64, 65536
m = DeepFM([V] * F, embedding_size=E, hidden_units=[512, 256, 128], dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001), catchup="bounded")
g = torch.Generator(device="cuda"); g.manual_seed(1)
m.init_variables(g, lin_scale=1e-3)
bs = [(torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g), (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)) for _ in range(8)]
step = m.train_step if mode == "eager" else m.graph_train_step
for i in range(24):
    step(*bs[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    step(*bs[i % 8])
torch.cuda.synchronize()
print("%s: %.3f ms/step" % (mode, (time.perf_counter() - t0) / n * 1e3))
