"""Training parity of the reference's other three trainers and of BASELINE config 4, against the oracle.

  trainers/linear.py:30-34       LinearClassifier: wide part, Ftrl(min(0.2, 1/sqrt(n))), loss SUM
  trainers/deep.py:32-38         DNNClassifier: embeddings -> MLP, Adagrad(0.05), dropout, loss SUM
  trainers/linear_deep.py:32-39  DNNLinearCombinedClassifier: both, Ftrl on TF's "linear" scope, Adagrad on
                                 its "dnn" scope, ONE forward, ONE loss, one global step

The oracle's two-optimizer train op is oracle.deepfm.TrainState(p, hp, lin_hp).  Each scenario runs
twice: on CPU with numpy stand-ins for the kernels (tests/cpu_kernels.py: the engine's variable layout
and step sequencing) and, under -m gpu, through the C ABI on the real kernels.

Bars: loss 2e-5 relative per step; after 3 steps every variable within 2e-6 absolute of the fp32
oracle (Ftrl / Adagrad at lr 0.05-0.2 move weights ~1e-2 per step; where the optimizer divides by a
sqrt of accumulated squares the kernels' summation order shows at the 1e-7 level: 2e-5 there, as for
the single-optimizer cases of test_hip_model.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import deepfm as O
from oracle import optimizers as OO
from tests.util import dropout_mask, make_problem


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def device(request):
    return request.param


def _engine(device, vocab, **kw):
    from mi355x_rec.engine import DeepFM
    if device == "cpu":
        from tests.cpu_kernels import NumpyKernels
        kw["_kernels"] = NumpyKernels()
    return DeepFM(vocab, device=device, **kw)


def _spec(name, lr):
    from mi355x_rec.engine import OptimizerSpec
    return OptimizerSpec(name, lr)


def _t(a, device):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _compare(m, p, atol):
    g = m.export_numpy()
    for f in range(len(p.emb)):
        if g.get("emb") is not None:
            assert g["emb"][f].shape == p.emb[f].shape
            assert np.max(np.abs(g["emb"][f] - p.emb[f]), initial=0.0) < atol, ("emb", f)
        if g.get("lin_w") is not None and g["lin_w"][f] is not None:       # (None: the column has no linear weight)
            assert np.max(np.abs(g["lin_w"][f] - p.lin_w[f])) < atol, ("lin_w", f)
    for i, (k, b) in enumerate(g["mlp"]):
        assert k.shape == p.mlp[i][0].shape
        assert np.max(np.abs(k - p.mlp[i][0])) < atol, ("kernel", i)
        assert np.max(np.abs(b - p.mlp[i][1])) < atol, ("bias", i)
    if m.use_linear:
        assert abs(g["lin_bias"][0] - p.lin_bias[0]) < atol
    if "num_emb" in g:
        assert np.max(np.abs(g["num_emb"] - p.num_emb)) < atol
    if "lin_num" in g:
        assert np.max(np.abs(g["lin_num"] - p.lin_num)) < atol


def _run(device, vocab, E, hidden, B, nn, numeric, flags, hp, lin_hp, dropout=0.0, steps=3, seed=21, atol=2e-5,
         reduction="sum", activation="relu", field_dims=None, wide_fields=None, deep_numeric=None, wide_numeric=None):
    ul, um, ud = flags
    rng0 = np.random.default_rng(seed)
    p = O.init_params(rng0, vocab, E, hidden, n_numeric=nn, dtype=np.float32, lin_scale=0.05, use_dnn=ud, numeric=numeric,
                      field_dims=field_dims, wide_fields=wide_fields, deep_numeric=deep_numeric)
    if wide_numeric is not None:
        p.lin_num[~np.asarray(wide_numeric, bool)] = 0
    subsets = dict(field_dims=field_dims, wide_fields=wide_fields, deep_numeric=deep_numeric, wide_numeric=wide_numeric)
    p.lin_bias[:] = 0.1
    for _, b in p.mlp:
        b[:] = (rng0.standard_normal(b.shape) * 0.05).astype(np.float32)
    kw = dict(n_numeric=nn, numeric=numeric, embedding_size=E, hidden_units=hidden, use_linear=ul, use_mf=um, use_dnn=ud,
              dropout=dropout, reduction=reduction, optimizer=_spec(hp.name, hp.lr), seed=3, activation=activation, **subsets)
    if lin_hp is not None:
        kw["linear_optimizer"] = _spec(lin_hp.name, lin_hp.lr)
    m = _engine(device, vocab, **kw)
    m.load_oracle_params(p)
    st = O.TrainState(p, hp, lin_hp)
    rng = np.random.default_rng(seed + 1)
    keep = 1.0 - dropout
    for step in range(steps):
        ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32) if vocab else np.zeros((B, 0), np.int32)
        if vocab:
            ids[B // 2] = ids[0]                        # duplicates inside a batch
        x = rng.standard_normal((B, nn)).astype(np.float32) if nn else None
        y = (rng.random(B) < 0.3).astype(np.uint8)
        masks = [dropout_mask(m._layer_seed(i), B, h, keep) for i, h in enumerate(hidden)] if (dropout and ud) else None
        lo, logit_o = O.train_step(p, st, ids, y, x, ul, um, ud, reduction, masks, numeric=numeric, keep_prob=keep,
                                   activation=activation, wide_fields=wide_fields, deep_numeric=deep_numeric,
                                   wide_numeric=wide_numeric)
        lg, logit_g = m.train_step(_t(ids, device), _t(y, device), _t(x, device))
        assert abs(lg.item() - float(lo)) <= 2e-5 * abs(float(lo)) + 1e-6, step
        lo_a, lg_a = logit_o, logit_g.cpu().numpy()
        assert np.max(np.abs(lg_a - lo_a)) <= 5e-5 * max(1.0, float(np.max(np.abs(lo_a)))), step
    _compare(m, p, atol)
    assert m.step == steps
    return m, p


ADAGRAD = OO.Hyper("Adagrad", 0.05)


def _ftrl(n_cols):
    return OO.Hyper("Ftrl", min(0.2, 1.0 / math.sqrt(n_cols)))


def test_linear_classifier_ftrl_sum(device):
    """trainers/linear.py:30-34: wide part only, Ftrl, SUM loss."""
    vocab = [9, 13, 5, 6, 2]
    _run(device, vocab, 4, [], 64, 0, "raw", (True, False, False), _ftrl(len(vocab)), None)


def test_linear_classifier_with_numeric_columns(device):
    vocab = [9, 13, 5]
    _run(device, vocab, 4, [], 48, 2, "raw", (True, False, False), _ftrl(len(vocab) + 2), None)


def test_dnn_classifier_adagrad_sum_dropout(device):
    """trainers/deep.py:32-38: embeddings -> MLP, Adagrad(0.05), dropout 0.1 (deep.py:67), SUM loss."""
    _run(device, [9, 13, 5, 6], 8, [16, 8], 64, 0, "raw", (False, False, True), ADAGRAD, None, dropout=0.1)


def test_wide_deep_two_optimizers(device):
    """trainers/linear_deep.py:32-39: Ftrl on the linear scope, Adagrad on the dnn scope, one step."""
    vocab = [9, 13, 5, 6]
    _run(device, vocab, 8, [16, 8], 64, 0, "raw", (True, False, True), ADAGRAD, _ftrl(len(vocab)), dropout=0.1)


def test_wide_deep_two_optimizers_with_numeric_columns(device):
    """... with numeric_column()s in both column lists (BASELINE config 4's shape): the value itself joins
    the concat, its linear_model weight follows the linear optimizer."""
    vocab = [9, 13, 5, 6]
    m, p = _run(device, vocab, 8, [16, 8], 64, 3, "raw", (True, False, True), ADAGRAD, _ftrl(len(vocab) + 3), dropout=0.1)
    assert m.D_in == 4 * 8 + 3 and m.D == 64               # zero pad to whole k-tiles ...
    assert float(m.kernel(0)[m.D_in:].abs().max()) == 0.0     # ... whose kernel rows stay zero


def test_wide_deep_numeric_columns_written_into_the_planes(device):
    """... at an embedding size the planes gather takes (E = 32): on the GPU the numeric values and the zero pad are written
    into the planes by the gather itself (engine.pl_numeric: no fp32 concat); 131 -> 160 input columns are no whole k-tile
    of the planes weight gradient, so layer 1's weight gradient gets the concat back through mi_merge_rows.  (On the CPU
    stand-ins the same model runs on the fp32 concat.)"""
    vocab = [9, 13, 5, 6]
    m, p = _run(device, vocab, 32, [64, 32], 96, 3, "raw", (True, False, True), ADAGRAD, _ftrl(len(vocab) + 3), dropout=0.1, seed=29,
                atol=4e-6)
    assert m.D_in == 4 * 32 + 3 and m.D == 160
    if device == "cuda":
        assert m.pl_numeric and "concat" in m._ws              # (the fallback weight gradient's fp32 copy)
    assert float(m.kernel(0)[m.D_in:].abs().max()) == 0.0


def test_deepfm_two_optimizers_numeric_embeddings_follow_the_deep_optimizer(device):
    """A DeepFM handed a linear_optimizer: numeric_embeddings (deep_fm.py:64) is an input-layer variable,
    not a linear_model one — it must take `optimizer`, only lin_w / bias / numeric linear weights take
    linear_optimizer (round-1 bug: the whole tail of the dense buffer went to the linear optimizer)."""
    vocab = [11, 5, 9]
    m, p = _run(device, vocab, 8, [12], 40, 2, "embed", (True, True, True), ADAGRAD, _ftrl(5), reduction="mean")
    assert m.num_emb_off < m.wide_off <= m.lin_bias_off < m.lin_num_off


def test_numeric_only_deepfm(device):
    """deep_fm.py:57-70 allows a model with numeric columns only (no categorical column at all)."""
    _run(device, [], 8, [12, 6], 32, 3, "embed", (True, True, True), OO.Hyper("Adam", 0.001), None, reduction="mean",
         atol=2e-6)


@pytest.mark.parametrize("activation,dropout", [("sigmoid", 0.0), ("tanh", 0.2), (None, 0.0), ("sigmoid", 0.2)])
def test_activation_other_than_relu(device, activation, dropout):
    """model_fn's params["activation"] (deep_fm.py:22,100; the reference passes a TF callable, default relu)"""
    _run(device, [9, 13, 5, 6], 8, [16, 8], 48, 0, "embed", (True, True, True), OO.Hyper("Adam", 0.001), None, dropout=dropout,
         reduction="mean", atol=3e-6, activation=activation)


def test_config4_shape_small_vocab(device):
    """BASELINE config 4 at its model shape — 26 categorical + 13 dense columns, E=64, [512,256,128],
    Ftrl + Adagrad, SUM loss, dropout 0.1 — with small vocabularies so the oracle finishes in seconds."""
    if device == "cpu":
        pytest.skip("the numpy stand-in needs minutes at this width; the layout is covered by the smaller cases")
    rng = np.random.default_rng(0)
    vocab = [int(v) for v in rng.integers(20, 200, 26)]
    _run(device, vocab, 64, [512, 256, 128], 384, 13, "raw", (True, False, True), ADAGRAD, _ftrl(39), dropout=0.1,
         steps=2, atol=5e-5)


def test_canned_estimators_accept_numeric_columns(device, monkeypatch):
    """DNNLinearCombinedClassifier(linear_feature_columns=cat + numeric, dnn_feature_columns=embeddings + numeric)
    through the Estimator surface: one train call, predictions, TF-named export incl. the permuted kernel rows."""
    if device == "cpu":
        from mi355x_rec import engine
        from tests.cpu_kernels import NumpyKernels
        monkeypatch.setattr(engine, "HipKernels", NumpyKernels)
    from mi355x_rec import feature_column as fc, tf_names
    from mi355x_rec.canned import DNNLinearCombinedClassifier
    from mi355x_rec.estimator import RunConfig
    cats = [fc.categorical_column_with_identity("war", 5), fc.categorical_column_with_identity("war2", 7),
            fc.categorical_column_with_hash_bucket("zip", 11)]
    nums = [fc.numeric_column("age"), fc.numeric_column("zz")]
    est = DNNLinearCombinedClassifier(model_dir=None, linear_feature_columns=cats + nums,
                                      dnn_feature_columns=[fc.embedding_column(c, 4) for c in cats] + nums,
                                      dnn_hidden_units=[8], dnn_dropout=0.1, config=RunConfig(device=device))
    rng = np.random.default_rng(1)
    B = 16
    feats = {"war": rng.integers(0, 5, B), "war2": rng.integers(0, 7, B), "zip": ["%05d" % z for z in rng.integers(0, 99999, B)],
             "age": rng.random(B).astype(np.float32), "zz": rng.random(B).astype(np.float32)}
    labels = rng.random(B) < 0.4
    spec = est.model_fn(feats, labels, "train", est.params)
    assert np.isfinite(float(spec.loss))
    eng = est._engine()
    plan = est.params["_store"]["plan"]
    # embedding-name order (ADVICE: 'war2_embedding' < 'war_embedding'), numeric columns after the block
    assert [c.name for c in plan.categorical] == ["war2", "war", "zip"]
    assert eng.raw_numeric and eng.n_numeric == 2 and eng.lin_opt.name == "Ftrl" and eng.opt.name == "Adagrad"
    names, nnames = [c.name for c in plan.categorical], [c.name for c in plan.numeric]
    dump = tf_names.export_variables(eng, names, "dnn_linear_combined", nnames)
    k0 = dump["dnn/hiddenlayer_0/kernel"]
    assert k0.shape == (3 * 4 + 2, 8)
    # TF's input_layer order: age, war2_embedding, war_embedding, zip_embedding, zz
    ek = eng.kernel(0).cpu().numpy()
    assert np.array_equal(k0[0], ek[12]) and np.array_equal(k0[1:5], ek[0:4]) and np.array_equal(k0[13], ek[13])
    assert dump["linear/linear_model/age/weights"].shape == (1, 1)
    # round trip into a fresh engine
    est2 = DNNLinearCombinedClassifier(model_dir=None, linear_feature_columns=cats + nums,
                                       dnn_feature_columns=[fc.embedding_column(c, 4) for c in cats] + nums,
                                       dnn_hidden_units=[8], config=RunConfig(device=device))
    est2.model_fn(feats, labels, "_build", est2.params)
    tf_names.import_variables(est2._engine(), dump, names, "dnn_linear_combined", numeric_names=nnames)
    assert torch.equal(est2._engine().dense, eng.dense) and torch.equal(est2._engine().table, eng.table)


def test_combined_classifier_with_independent_column_lists(device, monkeypatch):
    """DNNLinearCombinedClassifier whose two column lists differ — a crossed-style column only the wide
    part has, an embedding column only the deep part has, embedding dimensions 6 and 3 (no multiples of 4), a numeric
    column per part — through the Estimator surface: the engine gets the union with per-part subsets, a TF-named export
    holds exactly the variables TF would create (shapes included) and round-trips into a fresh estimator."""
    if device == "cpu":
        from mi355x_rec import engine
        from tests.cpu_kernels import NumpyKernels
        monkeypatch.setattr(engine, "HipKernels", NumpyKernels)
    from mi355x_rec import feature_column as fc, tf_names
    from mi355x_rec.canned import DNNLinearCombinedClassifier
    from mi355x_rec.estimator import RunConfig
    a, b, c = (fc.categorical_column_with_identity("a", 5), fc.categorical_column_with_identity("b", 7),
               fc.categorical_column_with_hash_bucket("c", 11))
    age, inc = fc.numeric_column("age"), fc.numeric_column("inc")
    make = lambda: DNNLinearCombinedClassifier(model_dir=None, linear_feature_columns=[a, c, age],
                                               dnn_feature_columns=[fc.embedding_column(a, 6), fc.embedding_column(b, 3), inc],
                                               dnn_hidden_units=[8], dnn_dropout=0.1, config=RunConfig(device=device))
    est = make()
    rng = np.random.default_rng(2)
    B = 16
    feats = {"a": rng.integers(0, 5, B), "b": rng.integers(0, 7, B), "c": ["%05d" % z for z in rng.integers(0, 99999, B)],
             "age": rng.random(B).astype(np.float32), "inc": rng.random(B).astype(np.float32)}
    labels = rng.random(B) < 0.4
    for _ in range(3):
        spec = est.model_fn(feats, labels, "train", est.params)
    assert np.isfinite(float(spec.loss))
    eng = est._engine()
    plan = est.params["_store"]["plan"]
    names, nnames = [x.name for x in plan.categorical], [x.name for x in plan.numeric]
    assert names == ["a", "b", "c"] and nnames == ["age", "inc"]
    assert eng.E == 8 and eng.field_dims == [6, 3, 0] and eng.wide_fields == [True, False, True]
    assert eng.deep_numeric == [False, True] and eng.wide_numeric == [True, False]
    dump = tf_names.export_variables(eng, names, "dnn_linear_combined", nnames)
    pre = "dnn/input_from_feature_columns/input_layer/"
    assert sorted(dump) == sorted([pre + "a_embedding/embedding_weights", pre + "b_embedding/embedding_weights",
                                   "linear/linear_model/a/weights", "linear/linear_model/c/weights",
                                   "linear/linear_model/age/weights", "linear/linear_model/bias_weights",
                                   "dnn/hiddenlayer_0/kernel", "dnn/hiddenlayer_0/bias", "dnn/logits/kernel", "dnn/logits/bias"])
    assert dump[pre + "a_embedding/embedding_weights"].shape == (5, 6) and dump[pre + "b_embedding/embedding_weights"].shape == (7, 3)
    assert dump["dnn/hiddenlayer_0/kernel"].shape == (6 + 3 + 1, 8)
    # TF's input_layer order: a_embedding (6), b_embedding (3), inc; the engine's stored rows: a at 0..5, b at 8..10, inc at 25
    ek = eng.kernel(0).cpu().numpy()
    k0 = dump["dnn/hiddenlayer_0/kernel"]
    assert np.array_equal(k0[:6], ek[0:6]) and np.array_equal(k0[6:9], ek[8:11]) and np.array_equal(k0[9], ek[3 * 8 + 1])
    assert float(np.abs(ek[6:8]).max()) == 0.0 and float(np.abs(ek[11:25]).max()) == 0.0     # rows of no variable
    assert float(np.abs(dump["linear/linear_model/c/weights"]).max()) > 0.0                   # (Ftrl moved the wide-only column)
    est2 = make()
    est2.model_fn(feats, labels, "_build", est2.params)
    e2 = est2._engine()
    tf_names.import_variables(e2, dump, names, "dnn_linear_combined", numeric_names=nnames)
    p1 = est.model_fn(feats, None, "infer", est.params).predictions["logits"]
    p2 = est2.model_fn(feats, None, "infer", est2.params).predictions["logits"]
    assert torch.equal(torch.as_tensor(p1), torch.as_tensor(p2))


def test_out_of_vocabulary_id_without_oov_bucket_is_refused():
    """a vocabulary column with num_oov_buckets=0 emits default_value=-1 for unknown values; the
    kernels index rows unchecked, so the host refuses such ids before they reach the device."""
    from mi355x_rec import feature_column as fc
    col = fc.categorical_column_with_vocabulary_list("gender", ["F", "M"])
    plan = fc.FieldPlan([col])
    ids, _ = plan.transform({"gender": ["F", "M"]})
    assert ids.tolist() == [[0], [1]]
    with pytest.raises(ValueError, match="outside"):
        plan.transform({"gender": ["F", "X"]})


def test_two_different_adam_optimizers(device):
    """a model whose deep scope and linear scope follow two DIFFERENT Adam optimizers (TF keeps beta powers
    per optimizer: two lr_t schedules) — refused in round 2.  DeepFM with Adam(1e-3) on the deep scope and Adam(2e-2) on
    the wide part, 5 steps with fresh ids (rows sit out steps: two catch-up calls per step, each with its own table, the
    first leaving the stamps alone), against the oracle's two-optimizer train op."""
    _run(device, [9, 13, 5, 6], 8, [16, 8], 64, 0, "embed", (True, True, True), OO.Hyper("Adam", 0.001), OO.Hyper("Adam", 0.02),
         steps=5, seed=33, atol=4e-6, reduction="mean")


def test_more_hidden_layers_than_one_weight_split_launch_takes(device):
    """more than MI_MAX_WEIGHT_JOBS (8) hidden layers on the planes path — refused in round 2; now several
    mi_split_weights launches."""
    hidden = [32, 32, 16, 16, 16, 16, 16, 16, 16, 16]
    m, _ = _run(device, [9, 13, 5, 6], 16, hidden, 64, 0, "embed", (True, True, True), OO.Hyper("Adam", 0.001), None, steps=2,
                seed=35, atol=4e-6, reduction="mean")
    if device == "cuda":
        assert m.planes and len(m._ws["wjobs_train"]) == 2


def test_wide_and_deep_parts_on_different_columns_with_per_column_dimensions(device):
    """VERDICT r2 (8), the last refusals: DNNLinearCombinedClassifier(linear_feature_columns != dnn_feature_columns) with
    embedding columns of different dimensions (linear_deep.py:32-39 passes two independent lists).  Six
    categorical columns: dimensions 8 / 4 / 8 / 0 (wide only) / 4 / 8, of which columns 1 and 4 are deep only; three
    numeric columns: one in both parts, one wide only, one deep only.  Ftrl on the linear scope, Adagrad on the dnn
    scope, dropout; 4 steps against the oracle, which holds the ragged variables as TF does."""
    vocab = [9, 13, 5, 6, 7, 11]
    dims = [8, 4, 8, 0, 4, 8]
    wide = [True, False, True, True, False, True]
    m, p = _run(device, vocab, 8, [16, 8], 64, 3, "raw", (True, False, True), ADAGRAD, _ftrl(4 + 2), dropout=0.1, steps=4,
                seed=41, field_dims=dims, wide_fields=wide, deep_numeric=[True, False, True], wide_numeric=[True, True, False])
    g = m.export_numpy()
    assert [a.shape[1] for a in g["emb"]] == dims and g["emb_padding_max_abs"] == 0.0
    assert g["mlp"][0][0].shape == (sum(dims) + 2, 16)
    assert [a is None for a in g["lin_w"]] == [not w for w in wide]
    k0 = m.kernel(0).cpu().numpy()
    logical = np.zeros(k0.shape[0], bool); logical[m._k0_rows] = True
    assert float(np.abs(k0[~logical]).max()) == 0.0           # rows of no variable stayed zero through the training
    assert float(g["lin_num"][2]) == 0.0


def test_wide_and_deep_column_subsets_same_adam(device):
    """... and with ONE Adam for everything (a DeepFM-less model_fn of the same shape, lazily replayed rows): the wide
    part's kernel runs on its own columns' ids, the catch-up and the apply on every row of the batch."""
    vocab = [9, 13, 5, 6]
    _run(device, vocab, 8, [16], 48, 0, "raw", (True, False, True), OO.Hyper("Adam", 0.001), None, steps=5, seed=43, atol=4e-6,
         field_dims=[8, 4, 0, 8], wide_fields=[False, True, True, True], reduction="mean")


def test_adam_schedule_extends_past_a_restored_step():
    """lr_t(step) after restoring a checkpoint far into a run (step > 2 x table size)."""
    from mi355x_rec.engine import AdamSchedule, OptimizerSpec
    s = AdamSchedule(OptimizerSpec("Adam", 0.001), "cpu", capacity=16)
    v = s.lr_t(300)
    assert len(s.host) > 300 and s.table.numel() == len(s.host)
    pw = OO.AdamPowers(OO.Hyper("Adam", 0.001), np.float32)
    for _ in range(299):
        pw.finish()
    assert np.float32(v) == pw.lr_t(0.001)
