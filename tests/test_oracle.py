"""CPU: the oracle itself, pinned on the anchors SURVEY 8c lists (the reference has no tests and
TensorFlow cannot run here: PARITY UNPINNED — these are identities the reference's formulas must
satisfy and hand-computed traces, not reference outputs)."""
import numpy as np
import pytest

from oracle import deepfm as O
from oracle import optimizers as OO
from oracle.metrics import BinaryMetrics


def _problem(dtype=np.float64, nn=2):
    rng = np.random.default_rng(0)
    V, E, B = [5, 7, 3], 4, 6
    p = O.init_params(rng, V, E, [8, 6], n_numeric=nn, dtype=dtype, lin_scale=0.1)
    p.lin_bias[:] = 0.3
    for k, b in p.mlp:
        b[:] = rng.standard_normal(b.shape) * 0.1
    ids = np.stack([rng.integers(0, v, B) for v in V], 1).astype(np.int32)
    ids[1] = ids[0]
    x = rng.standard_normal((B, nn)).astype(dtype) if nn else None
    y = rng.integers(0, 2, B)
    return p, ids, x, y


def test_fm_equals_sum_of_pairwise_dot_products():
    p, ids, x, y = _problem()
    c = O.forward(p, ids, x)
    mat = c["concat"].reshape(len(ids), -1, 4)
    assert np.allclose(c["fm"], O.fm_pairwise(mat), rtol=1e-12, atol=1e-12)     # deep_fm.py:81-87 identity
    c1 = O.forward(p, ids[:, :1], None, use_dnn=False, use_linear=False)        # one field: no pairs
    assert np.all(c1["fm"] == 0)


def test_finite_difference_gradients_fp64():
    p, ids, x, y = _problem()
    loss_of = lambda: O.head(O.forward(p, ids, x)["logits"], y)[0]
    c = O.forward(p, ids, x)
    _, dl, _, _ = O.head(c["logits"], y)
    dense, d_rows, d_lin = O.backward(p, c, dl)
    eps = 1e-6
    for var, g in zip(p.dense_list(), dense):
        g = g.reshape(var.shape)
        for i in np.ndindex(*var.shape):
            old = var[i]
            var[i] = old + eps; lp = loss_of()
            var[i] = old - eps; lm = loss_of()
            var[i] = old
            assert abs((lp - lm) / (2 * eps) - g[i]) < 1e-8
    for f in range(3):
        G = np.zeros_like(p.emb[f]); np.add.at(G, ids[:, f], d_rows[:, f, :])
        for i in np.ndindex(*G.shape):
            old = p.emb[f][i]
            p.emb[f][i] = old + eps; lp = loss_of()
            p.emb[f][i] = old - eps; lm = loss_of()
            p.emb[f][i] = old
            assert abs((lp - lm) / (2 * eps) - G[i]) < 1e-8
        GL = np.zeros_like(p.lin_w[f]); np.add.at(GL, ids[:, f], d_lin[:, f])
        for i in range(len(GL)):
            old = p.lin_w[f][i]
            p.lin_w[f][i] = old + eps; lp = loss_of()
            p.lin_w[f][i] = old - eps; lm = loss_of()
            p.lin_w[f][i] = old
            assert abs((lp - lm) / (2 * eps) - GL[i]) < 1e-8


def test_sigmoid_cross_entropy_known_answers():
    x = np.array([0.0, 3.0, -3.0, 19.0, -19.0])
    y = np.array([1, 0, 1, 1, 0])
    loss, d, per, sig = O.head(x, y, "sum")
    assert per[0] == pytest.approx(np.log(2.0))
    naive = -(y * np.log(1 / (1 + np.exp(-x))) + (1 - y) * np.log(1 - 1 / (1 + np.exp(-x))))
    assert np.allclose(per, naive, rtol=1e-9)                     # stable form == naive form (|x| < 20)
    assert np.allclose(d, 1 / (1 + np.exp(-x)) - y)
    lm, dm, _, _ = O.head(x, y, "mean")
    assert lm == pytest.approx(loss / 5) and np.allclose(dm, d / 5)      # contrib head: SUM_OVER_BATCH_SIZE
    pr = O.predictions(x.astype(np.float32))
    assert pr["class_id"].tolist() == [0, 1, 0, 1, 0] and np.array_equal(pr["probabilities"], pr["logistic"])


def test_adam_first_step_closed_form_and_untouched_rows_drift():
    hp = OO.Hyper("Adam", 0.001)
    W = np.array([[1.0, -2.0], [0.5, 0.25], [3.0, 4.0]])
    m, v = OO.slot_init(hp, W)
    pw = OO.AdamPowers(hp, np.float64)
    g = np.array([[0.1, -0.2]])
    OO.sparse_apply(hp, W, m, v, np.array([1]), g, pw.lr_t(hp.lr)); pw.finish()
    # step 1 with m = v = 0: delta = -lr * g / (|g| + eps*sqrt(1-b2)) ~= -lr * sign(g)
    exp = np.array([0.5, 0.25]) - 0.001 * g[0] / (np.abs(g[0]) + 1e-8 * np.sqrt(1 - 0.999))
    assert np.allclose(W[1], exp, rtol=1e-12)
    assert np.array_equal(W[0], [1.0, -2.0]) and np.array_equal(W[2], [3.0, 4.0])    # m = v = 0: no move
    # step 2 touches row 0 only: row 1 (stale m, v) must KEEP MOVING (TF's whole-table update)
    w1 = W[1].copy()
    OO.sparse_apply(hp, W, m, v, np.array([0]), np.array([[0.3, 0.3]]), pw.lr_t(hp.lr)); pw.finish()
    m1 = 0.9 * (0.1 * g[0]); v1 = 0.999 * (0.001 * g[0] ** 2)
    lr2 = 0.001 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert np.allclose(W[1], w1 - lr2 * m1 / (np.sqrt(v1) + 1e-8), rtol=1e-12)
    assert not np.array_equal(W[1], w1)
    # duplicates are summed before the moments see them: v gets (sum g)^2
    W2 = np.zeros((2, 1)); m2, v2 = OO.slot_init(hp, W2)
    OO.sparse_apply(hp, W2, m2, v2, np.array([0, 0]), np.array([[1.0], [2.0]]), 0.001)
    assert v2[0, 0] == pytest.approx(0.001 * 9.0) and m2[0, 0] == pytest.approx(0.1 * 3.0)


def test_dense_and_sparse_adam_agree_on_touched_rows():
    hp = OO.Hyper("Adam", 0.001)
    rng = np.random.default_rng(1)
    W = rng.standard_normal((4, 3)).astype(np.float32); Wd = W.copy()
    ms, vs = OO.slot_init(hp, W); md, vd = OO.slot_init(hp, Wd)
    pw = OO.AdamPowers(hp, np.float32)
    for _ in range(3):
        g = rng.standard_normal((4, 3)).astype(np.float32)
        OO.sparse_apply(hp, W, ms, vs, np.arange(4), g, pw.lr_t(hp.lr))
        OO.dense_apply(hp, Wd, md, vd, g, pw.lr_t(hp.lr))
        pw.finish()
    assert np.allclose(W, Wd, rtol=2e-6, atol=1e-7)     # same rule, op order differs in the last bits


def test_other_optimizers_one_step_by_hand():
    g = np.array([0.5, -1.0]); w0 = np.array([1.0, 2.0])
    w = w0.copy(); a, _ = OO.slot_init(OO.Hyper("Adagrad", 0.05), w)
    OO.dense_apply(OO.Hyper("Adagrad", 0.05), w, a, _, g)
    assert np.allclose(w, w0 - 0.05 * g / np.sqrt(0.1 + g * g))
    w = w0.copy(); hp = OO.Hyper("Ftrl", 0.2); a, l = OO.slot_init(hp, w)
    OO.dense_apply(hp, w, a, l, g)
    na = 0.1 + g * g; lin = g - (np.sqrt(na) - np.sqrt(0.1)) / 0.2 * w0
    assert np.allclose(w, -lin / (np.sqrt(na) / 0.2)) and np.allclose(a, na) and np.allclose(l, lin)
    w = w0.copy(); OO.dense_apply(OO.Hyper("SGD", 0.1), w, None, None, g)
    assert np.allclose(w, w0 - 0.1 * g)
    w = w0.copy(); hp = OO.Hyper("RMSProp", 0.01); ms, mom = OO.slot_init(hp, w)
    OO.dense_apply(hp, w, ms, mom, g)
    assert np.allclose(w, w0 - 0.01 * g / np.sqrt(1 + 0.1 * (g * g - 1) + 1e-10))
    with pytest.raises(KeyError):
        OO.Hyper("Nadam")


def test_sync_data_parallel_identity():
    """sum over ranks of gradients scaled by 1/B_global == big-batch gradient."""
    p, ids, x, y = _problem()
    c = O.forward(p, ids, x); _, dl, _, _ = O.head(c["logits"], y)
    full, _, _ = O.backward(p, c, dl)
    acc = None
    for sl in (slice(0, 3), slice(3, 6)):
        c = O.forward(p, ids[sl], x[sl]); _, dl, _, _ = O.head(c["logits"], y[sl], global_batch=6)
        part, _, _ = O.backward(p, c, dl)
        acc = part if acc is None else [a + b for a, b in zip(acc, part)]
    for a, b in zip(acc, full):
        assert np.allclose(a, b, rtol=1e-12, atol=1e-14)


def test_auc_metrics():
    bm = BinaryMetrics()
    x = np.array([-4, -3, -2, 2, 3, 4], np.float32); y = np.array([0, 0, 0, 1, 1, 1])
    bm.update(x, y)
    r = bm.result()
    assert r["accuracy"] == 1.0 and r["auc"] > 0.999 and r["precision"] == 1.0 and r["recall"] == 1.0
    assert r["label/mean"] == 0.5 and r["accuracy_baseline"] == 0.5
    rng = np.random.default_rng(0)
    bm = BinaryMetrics()
    for _ in range(3):                                   # streaming: several batches
        bm.update(rng.standard_normal(4000).astype(np.float32), rng.integers(0, 2, 4000))
    assert abs(bm.result()["auc"] - 0.5) < 0.03
    from mi355x_rec.metrics import confusion_from_hist
    # the device histogram form gives back the same confusion counts
    hist = np.zeros((2, 201), np.int64)
    from oracle.metrics import auc_thresholds
    th = auc_thresholds()
    e = np.exp(-np.abs(x)); p = np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(np.float32)
    for pi, yi in zip(p, y):
        hist[yi, int((th < pi).sum())] += 1
    tp, fp, tn, fn = confusion_from_hist(hist)
    bm2 = BinaryMetrics(); bm2.update(x, y)
    assert np.array_equal(tp, bm2.tp) and np.array_equal(fp, bm2.fp) and np.array_equal(fn, bm2.fn)


def test_layer_summary_and_dropout_backward_rule():
    s = O.layer_summary(np.array([[0.0, 1.0], [2.0, 0.0]]))
    assert s["fraction_of_zero_values"] == 0.5 and s["max"] == 2.0                  # model_utils.py:4-6
    # dropout: backward through (relu -> mask) only needs the stored post-dropout activation
    p, ids, x, y = _problem(nn=0)
    rng = np.random.default_rng(3)
    masks = [(rng.random((6, 8)) < 0.7).astype(np.float64), (rng.random((6, 6)) < 0.7).astype(np.float64)]
    loss_of = lambda: O.head(O.forward(p, ids, None, dropout_masks=masks, keep_prob=0.7)["logits"], y)[0]
    c = O.forward(p, ids, None, dropout_masks=masks, keep_prob=0.7); _, dl, _, _ = O.head(c["logits"], y)
    dense, _, _ = O.backward(p, c, dl, masks)
    k0 = p.mlp[0][0]; eps = 1e-6
    for i in [(0, 0), (3, 5), (11, 7)]:
        old = k0[i]; k0[i] = old + eps; lp = loss_of(); k0[i] = old - eps; lm = loss_of(); k0[i] = old
        assert abs((lp - lm) / (2 * eps) - dense[0][i]) < 1e-8


def test_torch_cpu_restatement_matches_numpy_oracle():
    """oracle/cpu_torch.py (bench.py's multi-threaded cpu_baseline) == oracle/deepfm.py over 3 Adam steps"""
    import torch
    from oracle import cpu_torch as T
    vocab, E, hidden, B = [9, 13, 5, 6], 8, [16, 8], 64
    rng = np.random.default_rng(4)
    p = O.init_params(rng, vocab, E, hidden, dtype=np.float32, lin_scale=0.05)
    st_t = T.State(vocab, E, hidden)
    st_t.load_numpy(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    for _ in range(3):
        ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
        ids[1] = ids[0]
        y = (rng.random(B) < 0.3).astype(np.uint8)
        lo, logit_o = O.train_step(p, st, ids, y)
        lt, logit_t = T.train_step(st_t, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(y.astype(np.float32)))
        assert abs(float(lt) - float(lo)) < 1e-6 * abs(float(lo)) + 1e-7
        assert np.allclose(logit_t.numpy(), logit_o, rtol=1e-5, atol=1e-6)
    for f in range(len(vocab)):
        assert np.max(np.abs(st_t.emb[f].numpy() - p.emb[f])) < 1e-6
        assert np.max(np.abs(st_t.lin_w[f].numpy() - p.lin_w[f])) < 1e-6
    for (k, b), (ko, bo) in zip(st_t.mlp, p.mlp):
        assert np.max(np.abs(k.numpy() - ko)) < 1e-6 and np.max(np.abs(b.numpy() - bo)) < 1e-6
