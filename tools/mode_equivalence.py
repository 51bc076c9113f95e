#!/usr/bin/env python3
"""Training-level check of the three matrix-pipe paths: the same model, data and seeds trained for 400
steps with gemm = fp32 / bf16x3 / f16x2; reports the loss curve distance and the final evaluation
loss / AUC on held-out data.  (Trajectories are chaotic at the last bit — see DESIGN.md §5 — so the
curves differ by rounding noise; what must agree is where they go.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import numpy as np, torch
from mi355x_rec.engine import DeepFM, OptimizerSpec

F, V, E, H, B, STEPS = 12, 5000, 64, [512, 256, 128], 4096, 400
rng = np.random.default_rng(0)
w_true = rng.standard_normal((F, V)).astype(np.float32)
def batch(n):
    ids = rng.integers(0, V, (n, F)).astype(np.int32)
    s = w_true[np.arange(F)[None, :], ids].sum(1) + 0.5 * w_true[0, ids[:, 0]] * w_true[1, ids[:, 1]]
    y = (s + rng.standard_normal(n) > 0).astype(np.uint8)
    return torch.from_numpy(ids).cuda(), torch.from_numpy(y).cuda()
train = [batch(B) for _ in range(STEPS)]
held = [batch(B) for _ in range(8)]
curves, finals = {}, {}
for mode in ("fp32", "bf16x3", "f16x2"):
    m = DeepFM([V] * F, embedding_size=E, hidden_units=H, dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001), gemm=mode, seed=3)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    m.init_variables(g)
    losses = []
    for ids, y in train:
        l, _ = m.train_step(ids, y)
        losses.append(l.clone())
    curves[mode] = torch.stack(losses).view(-1).cpu().numpy()
    tot, logits_all, y_all = 0.0, [], []
    for ids, y in held:
        l, lg = m.loss(ids, y)
        tot += float(l.item()); logits_all.append(lg.cpu().numpy().copy()); y_all.append(y.cpu().numpy())
    lg, yy = np.concatenate(logits_all), np.concatenate(y_all)
    order = np.argsort(lg); ranks = np.empty_like(order, dtype=np.float64); ranks[order] = np.arange(1, len(lg) + 1)
    npos = yy.sum(); auc = (ranks[yy == 1].sum() - npos * (npos + 1) / 2) / (npos * (len(yy) - npos))
    finals[mode] = (tot / len(held), auc)
ref = curves["fp32"]
for mode in ("fp32", "bf16x3", "f16x2"):
    c = curves[mode]
    print("%-7s loss[0] %.6f  loss[99] %.6f  loss[399] %.6f | max |curve - fp32 curve| %.2e | held-out loss %.6f  AUC %.5f" % (
        mode, c[0], c[99], c[-1], np.max(np.abs(c - ref)), finals[mode][0], finals[mode][1]))
