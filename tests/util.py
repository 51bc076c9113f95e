"""Shared helpers for the parity tests (oracle side = checker, HIP side = thing under test)."""
import numpy as np
import torch

from oracle import deepfm as O
from oracle import optimizers as OO

MASK64 = (1 << 64) - 1


def _mix32(x):
    x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7feb352d)
    x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846ca68b)
    return x ^ (x >> np.uint32(16))


def dropout_mask(seed, M, N, keep):
    """Host replica of the kernels' counter-based dropout mask (csrc/common.h, mi_drop_*): keep flag {0, 1} per element
    (kept values are divided by keep, as tf.nn.dropout does).  Two decisions per 32-bit hash: element (row, col) is kept
    iff its 16 bits of hash(row, col >> 1) — low half for an even column, high half for an odd one — are below keep * 2^16."""
    u = np.uint32
    row = np.arange(M, dtype=np.uint32)[:, None]
    col = np.arange(N, dtype=np.uint32)[None, :]
    with np.errstate(over="ignore"):
        s = u(seed & 0xFFFFFFFF) ^ (u((seed >> 32) & 0xFFFFFFFF) * u(0xC2B2AE35))
        rowkey = _mix32((row * u(0x9E3779B1)) ^ s)
        x = rowkey + (col >> u(1)) * u(0x85EBCA77)
        x = ~x + (x << u(15))
        x = x ^ (x >> u(12))
        x = x + (x << u(2))
        x = x ^ (x >> u(4))
        x = x + (x << u(3))
        x = x ^ (x >> u(11))
        x = x + (x << u(11))
        x = x ^ (x >> u(16))
    bits = np.where((col & u(1)) == 1, x >> u(16), x & u(0xFFFF))
    thresh = u(np.float32(keep) * np.float32(65536.0))
    return (bits < thresh).astype(np.float32)


def make_problem(seed, vocab, E, hidden, B, n_numeric=0, lin_scale=0.05, dup=True, use_dnn=True):
    rng = np.random.default_rng(seed)
    p = O.init_params(rng, vocab, E, hidden, n_numeric=n_numeric, dtype=np.float32, lin_scale=lin_scale,
                      use_dnn=use_dnn)
    p.lin_bias[:] = 0.1
    for k, b in p.mlp:
        b[:] = (rng.standard_normal(b.shape) * 0.05).astype(np.float32)
    ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
    if dup and B > 3:
        ids[B // 2] = ids[0]          # duplicate rows inside one batch
        ids[B - 1, 0] = ids[1, 0]
    x = rng.standard_normal((B, n_numeric)).astype(np.float32) if n_numeric else None
    y = (rng.random(B) < 0.3).astype(np.uint8)
    return p, ids, x, y


def dev(a, device="cuda"):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-30))) if a.size else 0.0


def max_err_scaled(a, b):
    """max |a-b| / max(|b|, rms(b)): a relative error that does not blow up on entries near 0."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    floor = np.sqrt(np.mean(b * b)) + 1e-30
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
