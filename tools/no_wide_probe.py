#!/usr/bin/env python3
"""Upper bound of what the wide part's 16-byte records cost a config-3 step: the same steps with and without the wide part
(use_linear): its catch-up, forward, the apply's record traffic and every fork / join of its stream go away with it; the Adam
stamps then live in a plain int32 array.  usage: python tools/no_wide_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec
F, V, E, H, B = 26, 1_000_000, 64, [512, 256, 128], 65536
g = torch.Generator(device="cuda"); g.manual_seed(1)
NB = 96
batches = [(torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g),
            (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)) for _ in range(NB)]
res = {}
for wide in (True, False, True, False):
    m = DeepFM([V] * F, embedding_size=E, hidden_units=H, dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001), use_linear=wide)
    m.init_variables(g, lin_scale=1e-3)
    i = [0]
    def step():
        a, b = batches[i[0] % NB]; i[0] += 1
        return m.train_step(a, b, next_ids=batches[i[0] % NB][0])
    for _ in range(70): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40 * 1e3
    res.setdefault(wide, []).append(dt)
    print("use_linear=%s: %.3f ms/step" % (wide, dt), flush=True)
    del m; torch.cuda.empty_cache()
print("the wide part costs the step %.3f ms" % (min(res[True]) - min(res[False])))
