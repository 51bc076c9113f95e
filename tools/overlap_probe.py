#!/usr/bin/env python3
"""Would the Adam catch-up hide under the backward GEMMs — or under the sparse apply?    Runs
config-3 train steps with an EXTRA catch-up of the same size on shadow copies of the table and slots — (a) not at
all, (b) serialised on the main stream before the phase, (c) on a side stream beside it — and compares step
times.  (b) - (a) is the catch-up's cost; (c) - (a) is what is left of it when overlapped.
PROBE=backward (default): beside the backward GEMMs; PROBE=apply: beside mi_sparse_apply_fused."""
import os, sys, time
PHASE = os.environ.get("PROBE", "backward")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec
F, V, E, H, B = 26, 1_000_000, 64, [512, 256, 128], 65536
m = DeepFM([V] * F, embedding_size=E, hidden_units=H, dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001))
g = torch.Generator(device="cuda"); g.manual_seed(1)
m.init_variables(g, lin_scale=1e-3)
batches = [(torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g), (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)) for _ in range(64)]
t2, m2, v2 = m.table.clone(), torch.rand_like(m.table) * 1e-6, torch.rand_like(m.table) * 1e-10 + 1e-12
last2 = torch.ones(m.R, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
mode = {"v": "none"}
orig = m._backward_dense if PHASE == "backward" else m._apply
def patched(*a, **kw):
    if mode["v"] != "none" and m.step > 20:
        uniq, nu = m._ws["uniq_by_gap"] if "uniq_by_gap" in m._ws else m._ws["own_uniq"], m._ws["own_nu"]
        n = B * F
        last2.fill_(m.step - 14)                                  # every shadow row is 14 steps stale
        s = m.sched.spec
        def launch():
            m.k.mi_sparse_catchup(t2, m2, v2, None, None, None, last2, uniq[:n], nu, n, E, m.step, m.sched.table, s.beta1, s.beta2, s.epsilon, 1, 1, 0)
        if mode["v"] == "serial":
            launch()
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                launch()
    out = orig(*a, **kw)
    if mode["v"] == "side" and m.step > 20:
        torch.cuda.current_stream().wait_stream(side)
    return out
if PHASE == "backward":
    m._backward_dense = patched
else:
    m._apply = patched
for i in range(30):
    m.train_step(*batches[i % 64])
res = {}
for md in ("none", "serial", "side", "none", "serial", "side"):
    mode["v"] = md
    for i in range(5): m.train_step(*batches[(i + 7) % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40): m.train_step(*batches[(i + 13) % 64])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40 * 1e3
    res.setdefault(md, []).append(dt)
for md, v in res.items():
    print("%-7s %s ms/step" % (md, " ".join("%.3f" % x for x in v)))
a, b, c = (min(res[k]) for k in ("none", "serial", "side"))
print("extra catch-up: %.3f ms serialised, %.3f ms beside the %s (%.0f %% hidden)" % (b - a, c - a, "backward GEMMs" if PHASE == "backward" else "sparse apply", 100 * (1 - (c - a) / (b - a))))
