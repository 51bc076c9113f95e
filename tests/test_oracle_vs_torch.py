"""The numpy oracle against an INDEPENDENT implementation of the same mathematics: PyTorch's autograd and torch.optim on CPU.

TensorFlow 1.12 cannot run here and the reference holds no fixtures (oracle/__init__.py: PARITY UNPINNED), so the oracle's
hand-written backward and optimizer rules are otherwise checked only against themselves (finite differences, closed forms).
Here the DeepFM graph of trainers/deep_fm.py:36-125 is written once more with plain torch ops and differentiated by
autograd — no line of oracle/ is used to produce the expected values — and the oracle's dense Adam / Adagrad / SGD rules are
stepped beside torch.optim's.  What this pins: the calculus and the update algebra.  What it cannot pin: that TF 1.12
computes exactly this graph (that is the restatement's claim)."""
import numpy as np
import pytest
import torch

from oracle import deepfm as O, optimizers as OO


def _torch_params(p):
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    return {"emb": [t(a) for a in p.emb], "lin_w": [t(a) for a in p.lin_w], "lin_bias": t(p.lin_bias),
            "mlp": [(t(k), t(b)) for k, b in p.mlp], "num_emb": None if p.num_emb is None else t(p.num_emb),
            "lin_num": None if p.lin_num is None else t(p.lin_num)}


def _torch_deepfm(tp, ids, y, x_num=None, masks=None, keep=1.0):
    """deep_fm.py:36-125 with torch ops: linear_model (:39), input_layer of embedding columns (:52-54), numeric embeddings
    (:62-73), FM second order (:79-87), hidden layers + dropout (:98-103), logits layer (:108), sigmoid cross-entropy,
    mean over the batch (:118-125)."""
    B, F = ids.shape
    idx = torch.from_numpy(ids.astype(np.int64))
    lin = sum(tp["lin_w"][f][idx[:, f]] for f in range(F)) + tp["lin_bias"][0]
    parts = [tp["emb"][f][idx[:, f]] for f in range(F)]
    if x_num is not None:
        xn = torch.from_numpy(x_num.astype(np.float64))
        lin = lin + (xn * tp["lin_num"][None, :]).sum(1)
        parts += [xn[:, j:j + 1] * tp["num_emb"][j][None, :] for j in range(xn.shape[1])]
    mat = torch.stack(parts, 1)                                         # [B, d, E]
    fm = 0.5 * (mat.sum(1) ** 2 - (mat ** 2).sum(1)).sum(1)
    net = mat.reshape(B, -1)
    for i, (k, b) in enumerate(tp["mlp"][:-1]):
        net = torch.relu(net @ k + b)
        if masks is not None:
            net = net / keep * torch.from_numpy(masks[i].astype(np.float64))
    k, b = tp["mlp"][-1]
    logits = lin + fm + (net @ k + b)[:, 0]
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.from_numpy(y.astype(np.float64)), reduction="mean")
    return logits, loss


@pytest.mark.parametrize("numeric,dropout", [(0, False), (3, False), (0, True), (2, True)])
def test_forward_loss_and_every_gradient_equal_torch_autograd(numeric, dropout):
    rng = np.random.default_rng(11 + numeric)
    vocab, E, hidden, B = [7, 12, 5, 9, 4], 6, [10, 8], 48
    p = O.init_params(rng, vocab, E, hidden, n_numeric=numeric, dtype=np.float64, lin_scale=0.1)
    for k, b in p.mlp:                                                  # (biases start at 0: give them values)
        b += rng.standard_normal(b.shape) * 0.1
    p.lin_bias += 0.3
    ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
    ids[3] = ids[0]; ids[4] = ids[0]                                    # duplicate rows inside the batch
    y = (rng.random(B) < 0.4).astype(np.uint8)
    x_num = rng.standard_normal((B, numeric)) if numeric else None
    keep = 0.8
    masks = [(rng.random((B, h)) < keep) for h in hidden] if dropout else None
    c = O.forward(p, ids, x_num, dropout_masks=masks, keep_prob=keep if dropout else 1.0)
    loss, d_logits, _, _ = O.head(c["logits"], y)
    dense, d_rows, d_lin = O.backward(p, c, d_logits, masks)

    tp = _torch_params(p)
    logits_t, loss_t = _torch_deepfm(tp, ids, y, x_num, masks, keep)
    loss_t.backward()
    assert np.allclose(c["logits"], logits_t.detach().numpy(), rtol=1e-12, atol=1e-12)
    assert abs(float(loss) - loss_t.item()) < 1e-13
    # dense variables, in Params.dense_list() order: (kernel, bias) per layer, linear bias, numeric embeddings, numeric weights
    expect = []
    for k, b in tp["mlp"]:
        expect += [k.grad.numpy(), b.grad.numpy()]
    expect.append(tp["lin_bias"].grad.numpy())
    if numeric:
        expect += [tp["num_emb"].grad.numpy(), tp["lin_num"].grad.numpy()]
    assert len(expect) == len(dense)
    for got, exp in zip(dense, expect):
        assert np.allclose(np.asarray(got).reshape(exp.shape), exp, rtol=1e-10, atol=1e-13)
    # the tables: autograd's dense table gradient = the oracle's per-entry gradients summed per row
    for f, v in enumerate(vocab):
        ge = np.zeros((v, E)); np.add.at(ge, ids[:, f], d_rows[:, f, :])
        gl = np.zeros(v); np.add.at(gl, ids[:, f], d_lin[:, f])
        assert np.allclose(ge, tp["emb"][f].grad.numpy(), rtol=1e-10, atol=1e-13)
        assert np.allclose(gl, tp["lin_w"][f].grad.numpy(), rtol=1e-10, atol=1e-13)


def test_dense_adam_adagrad_sgd_equal_torch_optim():
    """TF's Adam puts epsilon next to sqrt(v) ("epsilon hat"), torch's next to sqrt(v / (1 - beta2^t)): the two are the same
    update when torch's eps is eps_tf / sqrt(1 - beta2^t), set anew every step.  Adagrad: initial accumulator 0.1, eps 0."""
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal((6, 5))
    grads = [rng.standard_normal((6, 5)) * 10.0 ** rng.integers(-3, 1) for _ in range(12)]
    for name in ("Adam", "Adagrad", "SGD"):
        hp = OO.Hyper(name, lr=0.01)
        w = w0.copy()
        s0, s1 = OO.slot_init(hp, w)
        powers = OO.AdamPowers(hp, np.float64) if name == "Adam" else None
        wt = torch.tensor(w0.copy(), requires_grad=True)
        opt = {"Adam": lambda: torch.optim.Adam([wt], lr=0.01, betas=(0.9, 0.999), eps=1e-8),
               "Adagrad": lambda: torch.optim.Adagrad([wt], lr=0.01, initial_accumulator_value=0.1, eps=0.0),
               "SGD": lambda: torch.optim.SGD([wt], lr=0.01)}[name]()
        for t, g in enumerate(grads, 1):
            OO.dense_apply(hp, w, s0, s1, g, powers.lr_t(hp.lr) if powers else None)
            if powers:
                powers.finish()
                opt.param_groups[0]["eps"] = hp.epsilon / np.sqrt(1.0 - 0.999 ** t)
            wt.grad = torch.tensor(g)
            opt.step()
            assert np.allclose(w, wt.detach().numpy(), rtol=1e-11, atol=1e-14), (name, t)


def test_tf_sparse_adam_is_dense_adam_on_the_zero_padded_gradient():
    """AdamOptimizer._apply_sparse decays m, v and moves EVERY row of the variable — i.e. it is dense Adam fed a
    gradient that is zero on the rows the batch did not touch (duplicates summed first).  torch.optim.Adam on that dense
    gradient, stepped beside the oracle's sparse rule."""
    rng = np.random.default_rng(6)
    V, E = 20, 4
    w = rng.standard_normal((V, E)) * 0.3
    hp = OO.Hyper("Adam", lr=0.001)
    s0, s1 = OO.slot_init(hp, w)
    powers = OO.AdamPowers(hp, np.float64)
    wt = torch.tensor(w.copy(), requires_grad=True)
    opt = torch.optim.Adam([wt], lr=0.001, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 9):
        idx = rng.integers(0, V, 7)
        idx[1] = idx[0]
        vals = rng.standard_normal((7, E))
        OO.sparse_apply(hp, w, s0, s1, idx, vals, powers.lr_t(hp.lr))
        powers.finish()
        dense_g = np.zeros((V, E)); np.add.at(dense_g, idx, vals)
        opt.param_groups[0]["eps"] = hp.epsilon / np.sqrt(1.0 - 0.999 ** t)
        wt.grad = torch.tensor(dense_g)
        opt.step()
        assert np.allclose(w, wt.detach().numpy(), rtol=1e-11, atol=1e-14), t


def test_streaming_metrics_against_scikit_learn():
    """oracle/metrics.py (tf.metrics.auc's 200-threshold trapezoid, accuracy / precision / recall at 0.5, mean log loss)
    against scikit-learn's exact values on the same predictions: the thresholded AUC is an approximation — 200 thresholds
    on well-spread probabilities agree with the exact ROC area to ~1e-3 — the rest must agree to rounding."""
    sk = pytest.importorskip("sklearn.metrics")
    from oracle.metrics import BinaryMetrics
    rng = np.random.default_rng(12)
    n = 6000
    y = rng.integers(0, 2, n)
    x = (rng.standard_normal(n) + 1.2 * (y - 0.5)).astype(np.float32)          # informative logits (AUC ~ 0.8)
    bm = BinaryMetrics()
    for part in np.array_split(np.arange(n), 5):                                # streaming: five batches
        bm.update(x[part], y[part])
    r = bm.result()
    p = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    cls = (p.astype(np.float32) > 0.5).astype(int)
    assert abs(r["auc"] - sk.roc_auc_score(y, p)) < 2e-3
    assert abs(r["accuracy"] - sk.accuracy_score(y, cls)) < 1e-12
    assert abs(r["precision"] - sk.precision_score(y, cls)) < 1e-12
    assert abs(r["recall"] - sk.recall_score(y, cls)) < 1e-12
    assert abs(r["average_loss"] - sk.log_loss(y, p)) < 1e-6
    assert abs(r["auc_precision_recall"] - sk.average_precision_score(y, p)) < 2e-2   # (different interpolations of the PR curve)


@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_rmsprop_equals_torch_optim_where_the_two_definitions_coincide(momentum):
    """TF: ms += (g^2 - ms)(1 - decay), ms starting at 1; mom = momentum * mom + lr * g / sqrt(ms + eps); var -= mom.
    torch: eps OUTSIDE the root, lr applied to the buffer, square_avg starting at 0.  With eps = 0 and torch's state seeded with
    ones the two are the same recursion (mom_tf = lr * buf_torch)."""
    rng = np.random.default_rng(8)
    w0 = rng.standard_normal((5, 4))
    hp = OO.Hyper("RMSProp", lr=0.01, decay=0.9, momentum=momentum, epsilon=0.0)
    w = w0.copy()
    s0, s1 = OO.slot_init(hp, w)
    wt = torch.tensor(w0.copy(), requires_grad=True)
    opt = torch.optim.RMSprop([wt], lr=0.01, alpha=0.9, eps=0.0, momentum=momentum)
    opt.state[wt]["step"] = torch.tensor(0.0)
    opt.state[wt]["square_avg"] = torch.ones_like(wt)
    if momentum > 0:
        opt.state[wt]["momentum_buffer"] = torch.zeros_like(wt)
    for t in range(10):
        g = rng.standard_normal((5, 4))
        OO.dense_apply(hp, w, s0, s1, g)
        wt.grad = torch.tensor(g)
        opt.step()
        assert np.allclose(w, wt.detach().numpy(), rtol=1e-11, atol=1e-14), t
