"""-m gpu: the whole model_fn-shaped path (engine.DeepFM over the C ABI) against the oracle's
restatement of trainers/deep_fm.py:36-125 on identical weights and inputs.

Forward logits / loss: 1e-5 relative to the fp64 oracle (north_star's bar).
Training: after k optimizer steps every variable must agree with the fp32 oracle to 2e-6 absolute
(updates are ~lr = 1e-3 per step, so this is ~1e-3 of one step; Adam divides by sqrt(v)+1e-8, which
amplifies the 1e-7-relative differences between MFMA and OpenBLAS summation order for gradients
below ~3e-7 — see DESIGN.md "Parity").
"""
import numpy as np
import pytest
import torch

from oracle import deepfm as O
from oracle import optimizers as OO
from tests.util import dev, dropout_mask, make_problem, max_err_scaled

pytestmark = pytest.mark.gpu

ML100K_VOCAB = [2, 2, 7, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 2000, 2, 2, 50, 8, 2, 2, 2, 2, 1000, 2, 2, 1000]  # sorted order


def _engine(vocab, E, hidden, n_numeric=0, **kw):
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    opt = kw.pop("optimizer", OptimizerSpec("Adam", 0.001))
    kw.setdefault("catchup", "exact")       # (the library's default is "bounded": the tests that mean it say so)
    return DeepFM(vocab, n_numeric=n_numeric, embedding_size=E, hidden_units=hidden, optimizer=opt, **kw)


CONFIGS = [
    # vocab, E, hidden, B, n_numeric
    (ML100K_VOCAB, 4, [16, 16], 32, 0),              # trainers.deep_fm defaults (config 2)
    ([50, 30, 20, 40], 64, [512, 256, 128], 300, 0),  # config-3 shaped, small vocab
    ([11, 5, 9], 8, [32], 257, 2),                    # numeric columns (deep_fm.py:62-73)
    ([7] * 40, 128, [64, 32], 65, 0),                 # config-5 shaped
    ([6, 5, 4], 12, [10, 7], 5, 1),                   # nothing a multiple of 4 or of a tile: scalar-load GEMM paths
]


@pytest.mark.parametrize("vocab,E,hidden,B,nn", CONFIGS)
def test_forward_logits_and_loss(vocab, E, hidden, B, nn):
    p, ids, x, y = make_problem(1, vocab, E, hidden, B, n_numeric=nn)
    m = _engine(vocab, E, hidden, nn)
    m.load_oracle_params(p)
    loss, logits = m.loss(dev(ids), dev(y), dev(x))
    p64 = p.astype(np.float64)
    c = O.forward(p64, ids, None if x is None else x.astype(np.float64))
    l64 = O.head(c["logits"], y)[0]
    assert max_err_scaled(logits.cpu().numpy(), c["logits"]) < 1e-5
    assert abs(loss.item() - l64) / abs(l64) < 1e-5


@pytest.mark.parametrize("flags", [(True, True, True), (True, False, False), (False, True, False),
                                   (False, False, True), (True, False, True), (False, True, True)])
def test_component_flags(flags):
    """use_linear / use_mf / use_dnn (deep_fm.py:16-18,37,76,93) incl. training."""
    ul, um, ud = flags
    vocab, E, hidden, B = [9, 13, 5, 6], 8, [16, 8], 64
    p, ids, x, y = make_problem(2, vocab, E, hidden, B, use_dnn=ud)
    m = _engine(vocab, E, hidden, use_linear=ul, use_mf=um, use_dnn=ud)
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    for _ in range(3):
        loss_o, _ = O.train_step(p, st, ids, y, None, ul, um, ud)
        loss_g, _ = m.train_step(dev(ids), dev(y))
        assert abs(loss_g.item() - float(loss_o)) / abs(float(loss_o)) < 2e-5
    _compare_vars(m, p, 2e-6)


def test_model_fn_errors():
    from mi355x_rec.engine import DeepFM
    with pytest.raises(ValueError, match="At least 1 feature column"):
        DeepFM([], n_numeric=0)
    with pytest.raises(ValueError, match="At least 1 of linear, mf or dnn"):
        DeepFM([3, 4], use_linear=False, use_mf=False, use_dnn=False)


def _compare_vars(m, p, atol):
    g = m.export_numpy()
    for f in range(len(p.emb)):
        if g["emb"] is not None:
            assert np.max(np.abs(g["emb"][f] - p.emb[f])) < atol, ("emb", f)
        if g["lin_w"] is not None:
            assert np.max(np.abs(g["lin_w"][f] - p.lin_w[f])) < atol, ("lin_w", f)
    for i, (k, b) in enumerate(g["mlp"]):
        assert np.max(np.abs(k - p.mlp[i][0])) < atol, ("kernel", i)
        assert np.max(np.abs(b - p.mlp[i][1])) < atol, ("bias", i)
    assert abs(g["lin_bias"][0] - p.lin_bias[0]) < atol
    if "num_emb" in g:
        assert np.max(np.abs(g["num_emb"] - p.num_emb)) < atol
        assert np.max(np.abs(g["lin_num"] - p.lin_num)) < atol


def _device_relu_masks(m, B):
    """Which hidden units the device's last train step let through (activation > 0), per hidden layer: read from the
    stored activations — fp32, or the planes where a layer's output exists as planes only."""
    out = []
    for i, h in enumerate(m.hidden):
        if i in m._acts_in_planes:
            a = torch.empty(B, h, device="cuda")
            m.k.mi_merge_rows(m._pl["x%dp" % (i + 1)].struct, B, h, a, h)
        else:
            a = m._ws["act%d" % i][:B * h].view(B, h)
        out.append((a > 0).cpu().numpy())
    return out


# Multi-step Adam trajectories amplify last-bit differences: a hidden unit whose pre-activation is 0 to within the
# rounding of its dot product (7e-9 at step 3 of the 4-field config-3-shaped case) lands on either side of relu
# depending on summation order, and TF-form Adam then moves the few weights whose gradient is near zero by ~lr per
# step whichever way the sign falls.  Round 3: the test no longer loosens its bars for that case (3e-4 / 3e-3 in round
# 2).  The device's relu decisions are read back after each step and handed to the oracle (oracle.forward(relu_masks));
# they may differ from the oracle's own sign test ONLY on units whose pre-activation is within 1e-6 of 0 — asserted —
# and with both sides on the same branch every case is held to the standard bars over all five steps.
@pytest.mark.parametrize("gemm", ["f16x2", "fp32", "f16x2+bounded"])
@pytest.mark.parametrize("vocab,E,hidden,B,nn", CONFIGS)
def test_adam_training_matches_oracle(vocab, E, hidden, B, nn, gemm):
    """5 train steps with fresh batches (rows sit out steps, duplicates inside a batch): the lazy
    catch-up path must reproduce TF Adam's dense-equivalent sparse update.  "+bounded": the bounded-error replay
    (MI_CATCHUP_BOUNDED) — same bars."""
    catchup = "bounded" if gemm.endswith("bounded") else "exact"
    gemm = gemm.split("+")[0]
    # Two bars on the logits.  Step 0 — identical weights on both sides, which is what north_star's "logits within 1e-5 on
    # identical inputs" words — is held to 5e-6 (and test_forward_logits_and_loss to 1e-5 against fp64).  From step 1 on the
    # two sides no longer hold identical weights: every variable may differ by up to var_atol = 2e-6 (Adam's division by
    # sqrt(v) + 1e-8 turns a 1e-7-relative difference in a gradient near 3e-7 into a visibly different update), and a logit
    # is a sum over F E + sum(hidden) such variables times O(0.1 .. 1) activations: 5e-5 is that propagated difference (a
    # few dozen coherent 2e-6 terms), NOT a looser kernel bar — the kernels' own error on given weights stays at the step-0
    # level, which the forward tests pin on the final weights of other trajectories.
    logit_tol, var_atol = 5e-5, 2e-6
    p, ids, x, y = make_problem(3, vocab, E, hidden, B, n_numeric=nn)
    m = _engine(vocab, E, hidden, nn, gemm=gemm, catchup=catchup)
    if gemm == "fp32":
        m.GAP_SORT_MIN = 1               # also exercise the sort-rows-by-staleness path of the catch-up
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    rng = np.random.default_rng(0)
    flipped = 0
    for step in range(5):
        ids_s = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
        ids_s[B // 2] = ids_s[0]
        loss_g, logit_g = m.train_step(dev(ids_s), dev(y), dev(x))
        masks = _device_relu_masks(m, B)
        pre = O.forward(p, ids_s, x)["pre"]
        for mk, q in zip(masks, pre):
            diff = mk != (q > 0)
            flipped += int(diff.sum())
            assert not diff.any() or float(np.abs(q[diff]).max()) < 1e-6, (step, float(np.abs(q[diff]).max()))
        loss_o, logit_o = O.train_step(p, st, ids_s, y, x, relu_masks=masks)
        assert abs(loss_g.item() - float(loss_o)) / abs(float(loss_o)) < 2e-5, step
        assert max_err_scaled(logit_g.cpu().numpy(), logit_o) < logit_tol, step
        if step == 0:
            assert max_err_scaled(logit_g.cpu().numpy(), logit_o) < 5e-6
            _compare_vars(m, p, 2e-6)
    assert flipped <= 4, flipped          # (a handful of marginal units at most)
    _compare_vars(m, p, var_atol)
    assert m.step == 5


# A config-3-shaped problem (26 fields, E = 64, hidden [512, 256, 128]: the shape bench.py measures) whose trajectory
# has NO marginal unit: seed 319 keeps every hidden pre-activation of all five steps at least 1e-6 away from 0 in the
# oracle (4.0e-6 here; asserted below, so a change of the host's BLAS that moved it would be seen) — a relu decision
# cannot depend on summation order, no mask is replayed, and every GEMM mode and both catch-up modes are held to the
# standard bars.
CONFIG3_SAFE_SEED = 319


@pytest.mark.parametrize("gemm,catchup", [("f16x2", "exact"), ("fp32", "exact"), ("bf16x3", "exact"), ("f16x2", "bounded")])
def test_config3_shape_trajectory_is_tight(gemm, catchup):
    vocab, E, hidden, B = [40 + 3 * i for i in range(26)], 64, [512, 256, 128], 64
    p, ids, x, y = make_problem(CONFIG3_SAFE_SEED, vocab, E, hidden, B)
    m = _engine(vocab, E, hidden, gemm=gemm, catchup=catchup)
    assert m.planes == (gemm == "f16x2")
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    rng = np.random.default_rng(CONFIG3_SAFE_SEED)
    margin = np.inf
    for step in range(5):
        ids_s = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
        ids_s[B // 2] = ids_s[0]
        margin = min(margin, min(float(np.abs(q).min()) for q in O.forward(p, ids_s)["pre"]))
        loss_o, logit_o = O.train_step(p, st, ids_s, y)
        loss_g, logit_g = m.train_step(dev(ids_s), dev(y))
        assert abs(loss_g.item() - float(loss_o)) / abs(float(loss_o)) < 2e-5, step
        assert max_err_scaled(logit_g.cpu().numpy(), logit_o) < (5e-6 if step == 0 else 5e-5), step
    assert margin >= 1e-6, margin
    _compare_vars(m, p, 2e-6)


@pytest.mark.parametrize("name,lr", [("Adagrad", 0.05), ("Ftrl", 0.1), ("RMSProp", 0.001), ("SGD", 0.05)])
def test_other_optimizers_training(name, lr):
    """get_optimizer's other choices (model_utils.py:58-64)."""
    from mi355x_rec.engine import OptimizerSpec
    vocab, E, hidden, B = [9, 13, 5, 6], 8, [16, 8], 64
    p, ids, x, y = make_problem(4, vocab, E, hidden, B)
    m = _engine(vocab, E, hidden, optimizer=OptimizerSpec(name, lr))
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper(name, lr))
    for _ in range(3):
        loss_o, _ = O.train_step(p, st, ids, y)
        loss_g, _ = m.train_step(dev(ids), dev(y))
        assert abs(loss_g.item() - float(loss_o)) / abs(float(loss_o)) < 5e-5
    _compare_vars(m, p, 2e-5 if name != "RMSProp" else 2e-4)


def test_dropout_training_step_matches_oracle_with_same_mask():
    """TRAIN-mode dropout (deep_fm.py:102-103): the kernel's counter-based mask is replayed on the
    host and handed to the oracle, so both sides drop the same units."""
    vocab, E, hidden, B = [9, 13, 5, 6], 8, [32, 16], 128
    p, ids, x, y = make_problem(5, vocab, E, hidden, B)
    m = _engine(vocab, E, hidden, dropout=0.25, seed=7)
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    for _ in range(2):
        masks = [dropout_mask(m._layer_seed(i), B, h, 0.75) for i, h in enumerate(hidden)]
        loss_o, _ = O.train_step(p, st, ids, y, dropout_masks=masks, keep_prob=0.75)
        loss_g, _ = m.train_step(dev(ids), dev(y))
        assert abs(loss_g.item() - float(loss_o)) / abs(float(loss_o)) < 2e-5
    _compare_vars(m, p, 2e-6)


def test_sum_reduction_head():
    """canned estimators: loss_reduction=SUM."""
    vocab, E, hidden, B = [9, 13, 5], 4, [8], 40
    p, ids, x, y = make_problem(6, vocab, E, hidden, B)
    m = _engine(vocab, E, hidden, reduction="sum")
    m.load_oracle_params(p)
    loss, logits = m.loss(dev(ids), dev(y))
    c = O.forward(p.astype(np.float64), ids)
    assert abs(loss.item() - O.head(c["logits"], y, "sum")[0]) / loss.item() < 1e-5


def _fp64_logits(m, rows, x=None):
    """deep_fm.py:36-111 / linear_deep.py:32-39 in torch fp64 on the examples whose global rows are `rows` [n, F] (and raw
    numeric values x [n, ND]): wide sum + bias (+ x . w), FM second order (use_mf), the MLP on [rows | x] — no dropout."""
    v = m.table[rows].double()
    out = m.lin_w[rows].double().sum(1) + m.dense[m.lin_bias_off].double()
    if x is not None and m.lin_num_off is not None:
        out = out + x.double() @ m.dense[m.lin_num_off:m.lin_num_off + m.n_numeric].double()
    if m.use_mf:
        out = out + 0.5 * ((v.sum(1) ** 2).sum(1) - (v * v).sum((1, 2)))
    net = v.reshape(rows.shape[0], -1)
    if x is not None:
        net = torch.cat([net, x.double()], 1)
    for i in range(len(m.layers)):
        k = m.kernel(i).double()
        net = net @ (k[:net.shape[1]] if i == 0 else k) + m.bias(i).double()
        if i < len(m.layers) - 1:
            net = net.clamp_min(0)
    return out + net[:, 0]


def test_full_size_properties():
    """BASELINE config 3 at full size (B=65536, 26 x 1M rows, E=64, [512,256,128]) through
    size-independent properties: gather is an exact copy of the addressed rows, FM equals the
    pairwise-dot identity on sampled examples, the loss falls over a few steps, only touched rows
    change, and a replay from the same state gives the same bits (determinism)."""
    F, V, E, B = 26, 1_000_000, 64, 65536
    m = _engine([V] * F, E, [512, 256, 128])
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    m.init_variables(g, lin_scale=1e-3)
    ids = torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g)
    y = (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)
    table0 = m.table.clone()
    loss0, logits0 = m.loss(ids, y)
    l0 = loss0.item()
    rows = (ids.long() + m.field_off[None, :])
    sel = torch.arange(0, B, 997, device="cuda")
    # the eval forward at full size on the planes path (gather -> planes -> three LDS-DMA GEMMs -> logits layer -> head)
    # against a torch fp64 forward of sampled examples: north_star's 1e-5 on the logits
    assert m.planes and "x0p" in m._pl
    ref = _fp64_logits(m, rows[sel])
    err = (logits0[sel].double() - ref).abs() / ref.abs().clamp_min(ref.abs().mean())
    assert float(err.max()) < 1e-5, float(err.max())
    # the materialising form of the gather kernel is an exact copy of the addressed rows
    concat = torch.empty(B, F * E, device="cuda")
    m.k.mi_embed_fm_linear_fwd(m.table, None, m.field_off, ids, B, F, E, concat, F * E, None, None, None, None, 1, m.ts)
    concat = concat.view(B, F, E)
    assert torch.equal(concat[sel], m.table[rows[sel]])
    assert "concat" not in m._ws          # the training path never materialises it (gathered layer-1 operand)
    sumv = m._ws["sumv"][:B * E].view(B, E)
    assert float((sumv[sel].double() - concat[sel].double().sum(1)).abs().max()) < 1e-5
    v = concat[sel].double()
    pair = 0.5 * ((v.sum(1) ** 2).sum(1) - (v * v).sum((1, 2)))
    fm = m._ws["fm"][:B][sel].double()
    assert float(((fm - pair).abs() / pair.abs().clamp_min(pair.abs().mean())).max()) < 5e-5
    losses = [m.train_step(ids, y)[0].item() for _ in range(4)]
    assert losses[0] == pytest.approx(l0, rel=1e-6)
    assert losses[-1] < losses[0]
    changed = (m.table != table0).any(1)
    touched = torch.zeros(m.R, dtype=torch.bool, device="cuda"); touched[rows.reshape(-1)] = True
    assert not bool((changed & ~touched).any())
    assert int(changed.sum()) > 0.9 * int(touched.sum())
    # determinism: same state + same batch -> same bits
    keys = ("table", "t_s0", "t_s1", "lin_w", "l_s0", "l_s1", "last_step", "dense", "d_s0", "d_s1")
    snap = {k: getattr(m, k).clone() for k in keys}
    step = m.step
    a = m.train_step(ids, y)[1].clone(); ta = m.table.clone()
    for k in keys:
        getattr(m, k).copy_(snap[k])
    m.step = step
    b = m.train_step(ids, y)[1]
    assert torch.equal(a, b) and torch.equal(ta, m.table)


def _props_after_steps(m, ids, y, x, table0, rows, steps=3, replay=True):
    """loss falls, only touched rows change, a replay from the same state gives the same bits"""
    losses = [m.train_step(ids, y, x)[0].item() for _ in range(steps)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    changed = (m.table != table0).any(1)
    touched = torch.zeros(m.R, dtype=torch.bool, device="cuda"); touched[rows.reshape(-1)] = True
    assert not bool((changed & ~touched).any())
    assert int(changed.sum()) > 0.9 * int(touched.sum())
    if not replay:
        return losses
    keys = [k for k in ("table", "t_s0", "t_s1", "lin_w", "l_s0", "l_s1", "last_step", "dense", "d_s0", "d_s1", "dl_s0",
                        "dl_s1") if getattr(m, k, None) is not None]
    snap = {k: getattr(m, k).clone() for k in keys}
    step = m.step
    a = m.train_step(ids, y, x)[1].clone(); ta = m.table.clone(); da = m.dense.clone()
    for k in keys:
        getattr(m, k).copy_(snap[k])
    m.step = step
    b = m.train_step(ids, y, x)[1]
    assert torch.equal(a, b) and torch.equal(ta, m.table) and torch.equal(da, m.dense)
    return losses


def test_config4_full_size_properties():
    """BASELINE config 4 at full size on one GPU: Wide&Deep (trainers/linear_deep.py:32-39) with 26 x 1M-row
    categorical columns + 13 dense columns, E=64, [512,256,128], B=65536, Ftrl on the wide part + Adagrad on
    the deep part, SUM loss, dropout 0.1 — through size-independent properties (the oracle comparison at this
    model shape with small vocabularies is tests/test_canned_parity.py::test_config4_shape_small_vocab)."""
    from mi355x_rec.engine import OptimizerSpec
    F, V, E, B, ND = 26, 1_000_000, 64, 65536, 13
    m = _engine([V] * F, E, [512, 256, 128], ND, numeric="raw", use_mf=False, dropout=0.1, reduction="sum",
                optimizer=OptimizerSpec("Adagrad", 0.05), linear_optimizer=OptimizerSpec("Ftrl", min(0.2, 1 / np.sqrt(F + ND))))
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    m.init_variables(g, lin_scale=1e-3)
    ids = torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g)
    x = torch.log1p(-torch.log(torch.rand(B, ND, device="cuda", generator=g).clamp_min(1e-12)))   # log1p(Exp(1)), SURVEY 8d
    y = (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)
    assert m.D_in == F * E + ND and m.D == 1792 and m.pl_numeric and m.lin_opt.name == "Ftrl"    # (the planes weight gradient's k-tile)
    table0, lin0 = m.table.clone(), m.lin_w.clone()
    rows = ids.long() + m.field_off[None, :]
    loss0_t, logits0 = m.loss(ids, y, x)
    loss0 = loss0_t.item()                       # (the returned tensors are workspace the next step overwrites)
    # config 4's eval forward at full size on the planes path (the gather writes rows + numeric columns as planes) against a
    # torch fp64 forward of sampled examples: 1e-5 on the logits
    sel64 = torch.arange(0, B, 997, device="cuda")
    ref = _fp64_logits(m, rows[sel64], x[sel64])
    err = (logits0[sel64].double() - ref).abs() / ref.abs().clamp_min(ref.abs().mean())
    assert float(err.max()) < 1e-5, float(err.max())
    # the input_layer concat exists as planes only (the gather writes embedding rows, the numeric values and the zero pad
    # under one exponent per example: engine.pl_numeric): back to fp32, every value within 2^-21 of its example's maximum
    concat = torch.empty(B, m.D, device="cuda")
    m.k.mi_merge_rows(m._pl["x0p"].struct, B, m.D, concat, m.D)
    sel = torch.arange(0, B, 1009, device="cuda")
    want = torch.cat([m.table[rows[sel]].reshape(len(sel), F * E), x[sel]], 1)
    bound = want.abs().max(1, keepdim=True).values * 2.0 ** -21
    assert bool(((concat[sel, :F * E + ND] - want).abs() <= bound).all())                # the rows and the values themselves
    assert float(concat[:, F * E + ND:].abs().max()) == 0.0                             # zero pad
    losses = _props_after_steps(m, ids, y, x, table0, rows)
    assert losses[0] == pytest.approx(loss0, rel=2e-2)     # (the training forward drops 10 % of the units)
    assert float(m.kernel(0)[m.D_in:].abs().max()) == 0.0                               # pad rows stay zero
    lin_changed = m.lin_w != lin0
    touched = torch.zeros(m.R, dtype=torch.bool, device="cuda"); touched[rows.reshape(-1)] = True
    assert not bool((lin_changed & ~touched).any()) and int(lin_changed.sum()) > 0.9 * int(touched.sum())
    # the two optimizers' slots: Adagrad accumulators only grow from 0.1, Ftrl's second slot is in use
    assert float(m.t_s0.min()) >= 0.1 and float(m.l_s1.abs().max()) > 0 and m.t_s1 is None


def test_config5_one_rank_share_properties():
    """One rank's share of BASELINE config 5 (8 GPUs: 40 fields x 10M rows / 8 = 1.25M rows per field and rank,
    E=128, B=131072/8 = 16384): a 25.6 GB table (76.8 GB with Adam's slots) whose byte offsets pass 2^32 and
    whose element offsets pass 2^31.  Exact-copy gather across the whole table, then the training
    properties."""
    F, V, E, B = 40, 1_250_000, 128, 16384
    m = _engine([V] * F, E, [512, 256, 128])
    assert m.table.numel() * 4 > 2 ** 34 and m.table.numel() > 2 ** 32
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    # row-dependent contents without a 25 GB random fill: value = hash-like function of (row, col)
    c = torch.arange(E, device="cuda", dtype=torch.float32).unsqueeze(0)
    for r0 in range(0, m.R, 1 << 22):
        r = torch.arange(r0, min(m.R, r0 + (1 << 22)), device="cuda", dtype=torch.float32).unsqueeze(1)
        m.table[r0:r0 + (1 << 22)] = torch.sin(r * 0.37 + c * 1.3) * 0.09
    del r, c
    m.lin_w.normal_(0, 1e-3, generator=g)
    for i, (_, _, fan, h) in enumerate(m.layers):
        lim = (6.0 / (fan + h)) ** 0.5
        m.kernel(i).uniform_(-lim, lim, generator=g)
    ids = torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g)
    ids[:, -1] = torch.randint(V - 1000, V, (B,), device="cuda", dtype=torch.int32, generator=g)   # the table's last rows
    y = (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)
    rows = ids.long() + m.field_off[None, :]
    assert int(rows.max()) * E * 4 > 2 ** 34
    concat = torch.empty(B, F * E, device="cuda")
    m.k.mi_embed_fm_linear_fwd(m.table, None, m.field_off, ids, B, F, E, concat, F * E, None, None, None, None, 1, m.ts)
    sel = torch.arange(0, B, 331, device="cuda")
    assert torch.equal(concat[sel].view(-1, F, E), m.table[rows[sel]])
    del concat
    # the gathered layer-1 operand reads the same rows: eval logits vs a torch fp64 forward on sampled examples
    loss0, logits = m.loss(ids, y)
    v = m.table[rows[sel]].double()
    lin = m.lin_w[rows[sel]].double().sum(1) + m.dense[m.lin_bias_off].double()
    fm = 0.5 * ((v.sum(1) ** 2).sum(1) - (v * v).sum((1, 2)))
    net = v.reshape(len(sel), -1)
    for i in range(len(m.layers)):
        net = net @ m.kernel(i).double() + m.bias(i).double()
        if i < len(m.layers) - 1:
            net = net.clamp_min(0)
    ref = lin + fm + net[:, 0]
    err = (logits[sel].double() - ref).abs() / ref.abs().clamp_min(ref.abs().mean())
    assert float(err.max()) < 1e-5
    table0 = m.table.clone()
    _props_after_steps(m, ids, y, None, table0, rows, replay=False)    # (the replay check would hold 200 GB of clones)


# (the third case: B * F = 26,624 entries — the multi-kernel radix sort and the staleness sort inside the captured step,
# no side stream: a linear graph.  It faulted at replay while the sorts zeroed their counters with hipMemsetAsync.)
@pytest.mark.parametrize("vocab,E,hidden,B", [(ML100K_VOCAB, 4, [16, 16], 32), ([50, 30, 20, 40], 64, [512, 256, 128], 128),
                                              (ML100K_VOCAB, 4, [16, 16], 1024)])
def test_graph_train_step_replays_the_eager_step_bitwise(vocab, E, hidden, B):
    """The train step captured into one hipGraph (global step, Adam lr_t and the dropout seeds live in a device-
    resident step state advanced by the graph's first node) must leave the model bit for bit where the same
    sequence of eager steps leaves it: fresh batch every step, dropout on, rows sitting out steps, an eager step
    in between (the device step is resynchronised)."""
    from mi355x_rec.engine import OptimizerSpec
    p, ids, x, y = make_problem(9, vocab, E, hidden, B)
    ms = []
    for _ in range(2):
        m = _engine(vocab, E, hidden, dropout=0.25, seed=5, optimizer=OptimizerSpec("Adam", 0.001))
        m.load_oracle_params(p)
        ms.append(m)
    eager, graph = ms
    rng = np.random.default_rng(2)
    for step in range(9):
        ids_s = dev(np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32))
        ys = dev((rng.random(B) < 0.3).astype(np.uint8))
        le, ge = eager.train_step(ids_s, ys)
        if step == 5:
            lg, gg = graph.train_step(ids_s, ys)                  # an eager step between replays
        else:
            lg, gg = graph.graph_train_step(ids_s, ys)
        assert torch.equal(le, lg) and torch.equal(ge, gg), step
    assert graph._graph is not None and graph.step == eager.step == 9
    for k in ("table", "t_s0", "t_s1", "lin_state", "dense", "d_s0", "d_s1"):
        assert torch.equal(getattr(eager, k), getattr(graph, k)), k


def test_graph_train_step_with_column_subsets_and_two_batch_shapes_replays_the_eager_step_bitwise():
    """the captured step of the canned DNNLinearCombined model with dnn_feature_columns != linear_feature_columns
    (per-column dimensions, wide / deep column subsets, numeric columns only one part reads: the wide_idx index_select, the
    _frozen index_fill_, _k0_rows) — bit for bit the eager sequence; and a second batch shape (an epoch's short last batch)
    gets a graph of its own without evicting the first (one capture per shape, not one per change of shape)."""
    from mi355x_rec.engine import OptimizerSpec
    vocab, E, hidden, nn = [9, 13, 5, 7], 8, [16], 3
    kw = dict(numeric="raw", use_mf=False, reduction="sum", optimizer=OptimizerSpec("Adagrad", 0.05), linear_optimizer=OptimizerSpec("Ftrl", 0.18),
              field_dims=[8, 3, 0, 4], wide_fields=[True, False, True, True], deep_numeric=[True, False, True], wide_numeric=[False, True, True])
    ms = []
    for _ in range(2):
        m = _engine(vocab, E, hidden, nn, dropout=0.25, seed=5, **kw)
        g = torch.Generator(device="cuda"); g.manual_seed(3)
        m.init_variables(g, lin_scale=1e-2)
        ms.append(m)
    eager, graph = ms
    assert graph.graph_ok()
    rng = np.random.default_rng(2)
    captures = []
    orig = graph._capture
    graph._capture = lambda *a, **k: (captures.append((graph.step, a[0].shape[0])), orig(*a, **k))[1]
    for step, B in enumerate([32, 32, 32, 32, 11, 11, 11, 32, 32, 11, 32, 11]):
        ids_s = dev(np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32))
        xs = dev(rng.standard_normal((B, nn)).astype(np.float32))
        ys = dev((rng.random(B) < 0.3).astype(np.uint8))
        le, ge = eager.train_step(ids_s, ys, xs)
        lg, gg = graph.graph_train_step(ids_s, ys, xs)
        assert torch.equal(le, lg) and torch.equal(ge, gg), step
    # both shapes hold a graph, and once both have been seen (the second shape's sizing step may move a workspace, which
    # costs the first its graph once) alternating between them captures nothing
    assert len(graph._graphs) == 2 and all(st <= 7 for st, _ in captures) and len(captures) <= 3, captures
    for k in ("table", "t_s0", "lin_state", "dense", "d_s0"):
        assert torch.equal(getattr(eager, k), getattr(graph, k)), k
    assert float(graph.dense.index_select(0, graph._frozen).abs().max()) == 0.0


def test_layer_summaries_after_a_replay_know_what_the_captured_step_kept_on_the_chip():
    """with hip_graph "on" and B >= 4096 the captured step runs the last hidden layer inside the fused top launch — its
    output never reaches memory.  layer_summaries() after a REPLAY must leave that layer out (not report a stale buffer), and after
    an eval forward (which writes every layer) show it again."""
    from mi355x_rec.engine import OptimizerSpec
    vocab, E, hidden, B = [50, 30, 20, 40], 64, [256, 128], 4096
    m = _engine(vocab, E, hidden, dropout=0.1, seed=5, optimizer=OptimizerSpec("Adam", 0.001))
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    m.init_variables(g, lin_scale=1e-2)
    rng = np.random.default_rng(2)
    batch = lambda: (dev(np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)), dev((rng.random(B) < 0.3).astype(np.uint8)))
    for _ in range(4):                                   # eager (sizes), capture, two replays
        m.graph_train_step(*batch())
    assert m._graph is not None and m._graph["top"]
    s = m.layer_summaries()
    assert "dnn/hiddenlayer_0" in s and "dnn/hiddenlayer_1" not in s and "dnn/logits" in s
    m.summaries_next = True                              # an eager step that is looked at keeps every layer in memory
    m.train_step(*batch())
    assert "dnn/hiddenlayer_1" in m.layer_summaries()
    m.summaries_next = False
    m.graph_train_step(*batch())
    assert "dnn/hiddenlayer_1" not in m.layer_summaries()
    m.loss(*batch())
    assert "dnn/hiddenlayer_1" in m.layer_summaries()


@pytest.mark.parametrize("kind", ["deepfm-numeric-embeddings", "wide-and-deep-raw-numeric"])
def test_graph_train_step_with_numeric_columns_replays_the_eager_step_bitwise(kind):
    """Round 4: numeric columns inside the captured step (a third input copy) — DeepFM's numeric embeddings
    (deep_fm.py:62-73) with one Adam, and the canned Wide&Deep of config 4 (raw numeric columns, Ftrl + Adagrad, SUM loss:
    no Adam schedule at all) at the CLI's small-batch shape: bit for bit the eager sequence."""
    from mi355x_rec.engine import OptimizerSpec
    vocab, E, hidden, B, nn = ML100K_VOCAB, 4, [16, 16], 32, 3
    raw = kind.startswith("wide")
    kw = dict(numeric="raw", use_mf=False, reduction="sum", optimizer=OptimizerSpec("Adagrad", 0.05),
              linear_optimizer=OptimizerSpec("Ftrl", 0.18)) if raw else dict(optimizer=OptimizerSpec("Adam", 0.001))
    ms = []
    for _ in range(2):
        m = _engine(vocab, E, hidden, nn, dropout=0.25, seed=5, **kw)
        g = torch.Generator(device="cuda"); g.manual_seed(3)
        m.init_variables(g, lin_scale=1e-2)
        ms.append(m)
    eager, graph = ms
    assert graph.graph_ok()
    rng = np.random.default_rng(2)
    for step in range(8):
        ids_s = dev(np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32))
        xs = dev(rng.standard_normal((B, nn)).astype(np.float32))
        ys = dev((rng.random(B) < 0.3).astype(np.uint8))
        le, ge = eager.train_step(ids_s, ys, xs)
        lg, gg = graph.graph_train_step(ids_s, ys, xs)
        assert torch.equal(le, lg) and torch.equal(ge, gg), step
    assert graph._graph is not None and graph._graph["x"] is not None and graph.step == eager.step == 8
    for k in ("table", "t_s0", "lin_state", "dense", "d_s0"):
        assert torch.equal(getattr(eager, k), getattr(graph, k)), k


def test_captured_step_has_no_memset_nodes():
    """Round 2's GPU fault: a captured LINEAR step whose sorts zeroed their counters with hipMemsetAsync
    faulted at replay ("write access to a read-only page") although — tools/graph_memset_nodes.py, round 3 — both memset
    nodes pointed into a live torch allocation made before the capture, in bounds, and no workspace moved: the runtime's
    memset nodes are not trusted; the library zeroes with a kernel of its own.  The captured B = 1024 step (no side stream:
    a chain of nodes) must contain kernel and copy nodes only."""
    import ctypes as C
    from mi355x_rec.engine import OptimizerSpec
    vocab, B = ML100K_VOCAB, 1024
    p, ids, x, y = make_problem(23, vocab, 4, [16, 16], B)
    m = _engine(vocab, 4, [16, 16], dropout=0.1, seed=3, optimizer=OptimizerSpec("Adam", 0.001))
    m.load_oracle_params(p)
    di, dy = dev(ids), dev(y)
    m.train_step(di, dy)
    torch.cuda.synchronize()
    state = torch.zeros(16, dtype=torch.uint8, device="cuda")
    m._write_step_state(state)
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    m.k.query("mi_set_step_state", state.data_ptr())
    m._capturing = True
    try:
        with torch.cuda.graph(graph):
            m.k.mi_step_advance(state, m.sched.table)
            m.train_step(di.clone(), dy.clone())
    finally:
        m._capturing = False
        m.k.query("mi_set_step_state", None)
    hip = C.CDLL("libamdhip64.so")
    h = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(h, None, C.byref(n)) == 0 and n.value > 10
    arr = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(h, arr, C.byref(n)) == 0
    types = []
    for node in arr:
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(C.c_void_p(node), C.byref(t)) == 0
        types.append(t.value)
    assert 2 not in types, types               # hipGraphNodeTypeMemset
    assert set(types) <= {0, 1}, types         # kernels and copies


def test_graph_is_recaptured_when_its_buffers_moved():
    """a captured step holds the raw addresses of the engine's workspaces, planes and lr_t table; a later
    call that needs more room (loss() on a larger batch, a restored checkpoint, layer summaries) reallocates them and
    the caching allocator may hand the old storage to someone else.  The engine counts reallocations and captures
    again when the count moved: graph steps at B = 32, an evaluation at B = 4096, a checkpoint round trip, more graph
    steps — bit for bit the eager sequence."""
    from mi355x_rec.engine import OptimizerSpec
    vocab, E, hidden, B = ML100K_VOCAB, 4, [16, 16], 32
    p, ids, x, y = make_problem(21, vocab, E, hidden, B)
    ms = []
    for _ in range(2):
        m = _engine(vocab, E, hidden, dropout=0.1, seed=3, optimizer=OptimizerSpec("Adam", 0.001))
        m.load_oracle_params(p)
        ms.append(m)
    eager, graph = ms
    rng = np.random.default_rng(5)
    big = dev(np.stack([rng.integers(0, v, 4096) for v in vocab], 1).astype(np.int32))
    big_y = dev((rng.random(4096) < 0.3).astype(np.uint8))
    captures = []
    for step in range(12):
        ids_s = dev(np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32))
        ys = dev((rng.random(B) < 0.3).astype(np.uint8))
        if step == 4:                                   # an evaluation on a larger batch: every workspace grows
            g0 = graph._graph["graph"]
            le, _ = eager.loss(big, big_y); lg, _ = graph.loss(big, big_y)
            assert torch.equal(le, lg)
            torch.empty(1 << 22, device="cuda").fill_(float("nan"))       # (whoever gets the old storage scribbles on it)
        if step == 8:                                   # a checkpoint round trip drops the capture too
            graph.load_state_dict(graph.state_dict())
            assert graph._graph is None
            graph._graph_warm = None
        le, ge = eager.train_step(ids_s, ys)
        lg, gg = graph.graph_train_step(ids_s, ys)
        assert torch.equal(le, lg) and torch.equal(ge, gg), step
        captures.append(None if getattr(graph, "_graph", None) is None else id(graph._graph["graph"]))
        if step == 4:
            assert graph._graph is not None and graph._graph["graph"] is not g0       # captured again, not replayed into freed memory
    assert graph.step == eager.step == 12
    eager.finalize_rows(); graph.finalize_rows()          # (the checkpoint brought `graph`'s stale rows up to date at step 8: same bits, other time)
    for k in ("table", "t_s0", "t_s1", "lin_state", "dense", "d_s0", "d_s1"):
        assert torch.equal(getattr(eager, k), getattr(graph, k)), k


@pytest.mark.parametrize("vocab,E,hidden,B", [([3000, 500, 4000, 50, 2500, 7000], 64, [512, 256, 128], 4096),
                                              ([2000, 1000, 50, 1000], 4, [16, 16], 8192)])
def test_presorted_next_batch_is_bitwise_the_plain_step(vocab, E, hidden, B):
    """train_step(next_ids=...): the next batch's sort runs on a side stream beside this step's catch-up and is picked
    up by the next call when it is given that very tensor — the same bits as the plain sequence; an announced batch that
    does not come, or one modified in place after it was announced, is sorted again."""
    from mi355x_rec.engine import OptimizerSpec
    p, ids0, x, y = make_problem(13, vocab, E, hidden, B)
    ms = []
    for _ in range(2):
        m = _engine(vocab, E, hidden, dropout=0.1, seed=7, optimizer=OptimizerSpec("Adam", 0.001))
        m.load_oracle_params(p)
        ms.append(m)
    plain, pre = ms
    rng = np.random.default_rng(4)
    batches = [dev(np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)) for _ in range(9)]
    ys = dev((rng.random(B) < 0.3).astype(np.uint8))
    hits = 0
    for i in range(8):
        lp, gp = plain.train_step(batches[i], ys)
        if i == 5:
            batches[i].add_(0)                                          # in-place touch: the version counter moves
        hits += pre._presorted is not None and pre._presorted["ids"] is batches[i] and pre._presorted["version"] == batches[i]._version
        lq, gq = pre.train_step(batches[i], ys, next_ids=batches[i + 1] if i != 3 else batches[0])
        assert torch.equal(lp, lq) and torch.equal(gp, gq), i
    assert hits == 5                                                    # steps 1, 2, 3, 6, 7 (0: nothing announced; 4: another batch; 5: touched)
    plain.finalize_rows(); pre.finalize_rows()
    for k in ("table", "t_s0", "t_s1", "lin_state", "dense", "d_s0", "d_s1"):
        assert torch.equal(getattr(plain, k), getattr(pre, k)), k


@pytest.mark.parametrize("catchup", ["exact", "bounded"])
def test_bench_configuration_at_full_size_is_bitwise_the_plain_sequence(catchup):
    """What bench.py times, at the size it times it (config 3: B = 65536, 26 x 1M rows, E = 64, [512, 256, 128], dropout
    0.1, a fresh batch every step): the step with the NEXT batch announced — its sort on a side stream beside this step's
    catch-up, the staleness order made a step ahead from stamps this step's apply has not written yet, the wide part's
    replay on its own stream, the 1,024-workgroup catch-up — against the plain train_step(ids, y) sequence on a second
    model from the same variables: logits and loss of every step, then table, slots, wide records (with the stamps) and
    dense variables, bit for bit, in both catch-up modes.  (40 GB of state for the two models.)"""
    from mi355x_rec.engine import OptimizerSpec
    F, V, E, B = 26, 1_000_000, 64, 65536
    ms = []
    for _ in range(2):
        m = _engine([V] * F, E, [512, 256, 128], dropout=0.1, seed=11, optimizer=OptimizerSpec("Adam", 0.001), catchup=catchup)
        g = torch.Generator(device="cuda"); g.manual_seed(7)
        m.init_variables(g, lin_scale=1e-3)
        ms.append(m)
    plain, pre = ms
    assert torch.equal(plain.table, pre.table) and torch.equal(plain.dense, pre.dense)
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    n_steps = 8
    # half of every batch's ids from a 60,000-id window per field: many rows of a batch were touched 1 .. 7 steps ago and
    # have steps to replay (uniform ids over 1M rows would leave 94 % of a batch's rows without state this early)
    batches = []
    for _ in range(n_steps + 1):
        ids = torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g)
        ids[: B // 2] = torch.randint(0, 60_000, (B // 2, F), device="cuda", dtype=torch.int32, generator=g)
        batches.append(ids.contiguous())
    ys = [(torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8) for _ in range(n_steps)]
    hits = 0
    for i in range(n_steps):
        lp, gp = plain.train_step(batches[i], ys[i])
        lp, gp = lp.clone(), gp.clone()
        ps = pre._presorted
        hits += ps is not None and ps["ids"] is batches[i] and ps.get("by_gap_step") == pre.step
        lq, gq = pre.train_step(batches[i], ys[i], next_ids=batches[i + 1])
        assert torch.equal(lp, lq) and torch.equal(gp, gq), i
    assert hits == n_steps - 1                    # every step but the first used the sort AND the staleness order made ahead
    # the catch-up had real work: rows touched again after sitting out 1 .. 6 steps
    plain.finalize_rows(); pre.finalize_rows()
    for k in ("table", "t_s0", "t_s1", "lin_state", "dense", "d_s0", "d_s1"):
        assert torch.equal(getattr(plain, k), getattr(pre, k)), k
    assert plain.step == pre.step == n_steps


def test_last_hidden_layer_inside_the_fused_head_launch_matches_the_oracle():
    """Round 4: from 4,096 examples on a 128-unit last hidden layer runs inside mi_hidden_logits_head_fused (engine._top_fusable;
    its output never reaches memory).  Three dropout steps against the oracle (masks replayed), every variable within 2e-6;
    and the protocol with the Estimator: a step announced as summarised (summaries_next) keeps the unfused launches, so that
    layer_summaries() sees that layer's output; after a fused step the entry is left out rather than stale."""
    vocab, E, hidden, B = [50, 40, 30, 60, 20, 35, 45, 25], 32, [256, 128], 4096
    p, ids, x, y = make_problem(17, vocab, E, hidden, B)
    m = _engine(vocab, E, hidden, dropout=0.1, seed=11)
    assert m.planes
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    rng = np.random.default_rng(3)
    for step in range(3):
        ids_s = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
        masks = [dropout_mask(m._layer_seed(i), B, h, 0.9) for i, h in enumerate(hidden)]
        loss_o, logit_o = O.train_step(p, st, ids_s, y, dropout_masks=masks, keep_prob=0.9)
        m.summaries_next = step == 1
        loss_g, logit_g = m.train_step(dev(ids_s), dev(y))
        fused = getattr(m, "_top_step", None) == m.step - 1
        assert fused == (step != 1)
        assert ("dnn/hiddenlayer_1" in m.layer_summaries()) == (step == 1)
        assert abs(loss_g.item() - float(loss_o)) / abs(float(loss_o)) < 2e-5, step
        assert max_err_scaled(logit_g.cpu().numpy(), logit_o) < 5e-5, step
    _compare_vars(m, p, 2e-6)
