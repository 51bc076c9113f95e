#!/usr/bin/env python3
"""Generates tests/golden/deepfm_small.npz from the oracle (python tests/golden/make_golden.py).

SELF-GENERATED, NOT REFERENCE OUTPUT: TensorFlow 1.12 cannot run here and the reference holds no
fixtures, so this file pins the oracle against drift and gives the HIP path a frozen
target that does not depend on importing oracle code at the same commit.  Contents: the MovieLens
26-field schema at trainers.deep_fm's defaults (E=4, hidden [16,16], B=32), injected weights,
3 Adam steps; expected logits/loss per step in fp32 (and fp64 logits of step 0), final variables.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import deepfm as O, optimizers as OO  # noqa: E402
from oracle.columns import ml100k_fields, sorted_fields  # noqa: E402


def build():
    vocab = [f[4] for f in sorted_fields(ml100k_fields())]
    rng = np.random.default_rng(20240521)
    E, hidden, B, steps = 4, [16, 16], 32, 3
    p = O.init_params(rng, vocab, E, hidden, dtype=np.float32, lin_scale=0.02)
    p.lin_bias[:] = 0.05
    out = {"vocab": np.array(vocab), "E": E, "hidden": np.array(hidden), "steps": steps,
           "table0": np.concatenate(p.emb, 0), "lin_w0": np.concatenate(p.lin_w, 0), "lin_bias0": p.lin_bias.copy()}
    for i, (k, b) in enumerate(p.mlp):
        out["kernel0_%d" % i] = k.copy(); out["bias0_%d" % i] = b.copy()
    ids = np.stack([np.stack([rng.integers(0, v, B) for v in vocab], 1) for _ in range(steps)]).astype(np.int32)
    ids[:, 5] = ids[:, 2]                                   # duplicate examples inside each batch
    y = (rng.random((steps, B)) < 0.3).astype(np.uint8)
    out["ids"], out["labels"] = ids, y
    c64 = O.forward(p.astype(np.float64), ids[0])
    out["logits64_step0"] = c64["logits"]
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    losses, logits = [], []
    for s in range(steps):
        lo, lg = O.train_step(p, st, ids[s], y[s])
        losses.append(np.float32(lo)); logits.append(lg.copy())
    out["loss"], out["logits"] = np.array(losses), np.stack(logits)
    out["table_final"] = np.concatenate(p.emb, 0); out["lin_w_final"] = np.concatenate(p.lin_w, 0)
    out["lin_bias_final"] = p.lin_bias.copy()
    for i, (k, b) in enumerate(p.mlp):
        out["kernel_final_%d" % i] = k.copy(); out["bias_final_%d" % i] = b.copy()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "deepfm_small.npz"), **build())
    print("written")
