"""The drop-in surface: ``python -m trainers.{deep_fm,linear,deep,linear_deep}`` flags/defaults,
``model_fn(features, labels, mode, params)`` with the reference's params keys and errors,
``get_feature_columns / get_input_fn / serving_input_fn`` and the model_utils helpers
(reference: trainers/*.py).  The same scenarios run on CPU with numpy stand-ins for the kernels
(host logic) and, under -m gpu, on the real HIP path."""
import csv
import os

import numpy as np
import pytest
import torch

from trainers import deep_fm, deep, linear, linear_deep, ml_100k, model_utils, conf_utils, _cli


def _write_csv(path, n, seed):
    rng = np.random.default_rng(seed)
    occ = ["technician", "administrator", "student", "homemaker", "none", "engineer"]
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(ml_100k.COLUMNS)
        for _ in range(n):
            row = {c: (0 if d[0] == 0 else "null") for c, d in zip(ml_100k.COLUMNS, ml_100k.DEFAULTS)}
            uid, iid = int(rng.integers(1, 944)), int(rng.integers(1, 1683))
            g = rng.integers(0, 2, len(ml_100k.GENRE))
            # a learnable rule so that training visibly reduces the loss
            like = (g[1] == 1) if rng.random() < 0.9 else (g[1] == 0)
            row.update(user_id=uid, item_id=iid, rating=5 if like else int(rng.integers(1, 5)),
                       age=int(rng.integers(7, 74)), gender=str(rng.choice(["F", "M", ""])),
                       occupation=str(rng.choice(occ)), zipcode="%05d" % rng.integers(0, 99999),
                       release_year=int(rng.integers(1922, 1999)))
            row.update({k: int(v) for k, v in zip(ml_100k.GENRE, g)})
            w.writerow([row[c] for c in ml_100k.COLUMNS])


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def device(request, monkeypatch):
    if request.param == "cpu":
        from mi355x_rec import engine
        from tests.cpu_kernels import NumpyKernels
        monkeypatch.setattr(engine, "HipKernels", NumpyKernels)   # host-logic run: kernels stood in by numpy
    return request.param


@pytest.fixture
def data(tmp_path):
    _write_csv(tmp_path / "train.csv", 600, 1)
    _write_csv(tmp_path / "test.csv", 150, 2)
    return tmp_path


def test_cli_flags_and_defaults_match_reference():
    a = _cli.make_parser("deep_fm", ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout")).parse_args([])
    assert (a.train_csv, a.test_csv, a.job_dir) == ("data/ml-100k/train.csv", "data/ml-100k/test.csv", "checkpoints/deep_fm")
    assert (a.embedding_size, a.hidden_units, a.dropout, a.batch_size, a.train_steps) == (4, [16, 16], 0.1, 32, 20000)
    assert not (a.restore or a.exclude_linear or a.exclude_mf or a.exclude_dnn)
    for model, extra in (("linear", ()), ("deep", ("hidden_units", "dropout")), ("linear_deep", ("hidden_units", "dropout"))):
        b = _cli.make_parser(model, extra).parse_args(["--batch-size", "8"])
        assert b.job_dir == "checkpoints/" + model and b.batch_size == 8 and b.embedding_size == 4
    assert conf_utils.EVAL_INTERVAL == 60 and conf_utils.get_run_config().keep_checkpoint_max == 5


def test_input_fn_semantics(data):
    ev = list(ml_100k.get_input_fn(str(data / "test.csv"), "eval", batch_size=32)())
    assert [len(l) for _, l in ev] == [32, 32, 32, 32, 22]            # one pass, short last batch
    f, l = ev[0]
    assert set(f) == set(ml_100k.COLUMNS) - {"rating"} and l.dtype == bool
    assert f["user_id"].dtype == np.int32 and f["gender"][0] in ("F", "M", "null")   # "" -> default "null"
    it = ml_100k.get_input_fn(str(data / "train.csv"), batch_size=16, seed=3)()
    seen = [next(it)[0]["user_id"] for _ in range(80)]                  # 1280 > 600 rows: it repeats
    assert all(len(b) == 16 for b in seen)
    first = np.concatenate(seen[:4])
    assert not np.array_equal(first, list(ml_100k.get_input_fn(str(data / "train.csv"), "eval", 64)())[0][0]["user_id"])
    recv = ml_100k.serving_input_fn()
    assert set(recv.receiver_tensors) == {"user_id", "item_id", "age", "gender", "occupation", "zipcode",
                                          "release_year", *ml_100k.GENRE}


def test_column_transforms_of_a_whole_batch_equal_the_per_element_definition():
    """The host side of a large batch (hash-bucket and vocabulary columns, the shuffle buffer) runs on whole numpy columns;
    the results are those of the per-element definitions: utf-8 bytes of str(value) -> Fingerprint64 % buckets (ASCII,
    non-ASCII, empty and long values, bytes), vocabulary index / OOV bucket, and tf.data's swap-out shuffle buffer."""
    from mi355x_rec import feature_column as fc
    rng = np.random.default_rng(0)
    col = fc.categorical_column_with_hash_bucket("z", 1000)
    vals = np.array(["%05d" % z for z in rng.integers(0, 99999, 500)] + ["", "a", "h\u00e9llo", "x" * 70, "tail\0"], dtype=object)
    one_by_one = np.array([col.transform({"z": [v]})[0] for v in vals])
    assert np.array_equal(col.transform({"z": vals}), one_by_one)                       # (non-ASCII / NUL: the per-element path)
    assert np.array_equal(col.transform({"z": vals[:500]}), one_by_one[:500])           # the whole-column path
    assert np.array_equal(col.transform({"z": vals[:500].astype("U")}), one_by_one[:500])
    assert np.array_equal(col.transform({"z": np.array([b"ab", b"c"], dtype=object)}),
                          [col.transform({"z": ["ab"]})[0], col.transform({"z": ["c"]})[0]])
    vc = fc.categorical_column_with_vocabulary_list("g", ["F", "M"], num_oov_buckets=2)
    g = np.array(rng.choice(["F", "M", "null", "other"], 300), dtype=object)
    assert np.array_equal(vc.transform({"g": g}), np.array([vc.transform({"g": [x]})[0] for x in g]))
    # the shuffle buffer: every element of a pass exactly once, no element before the buffer is full, and an element never
    # leaves before one that entered cap or more positions after it has entered (it can only be delayed)
    n, bs = 2000, 16
    it = ml_100k.get_input_fn("synthetic:%d:3" % n, batch_size=bs, seed=1)()
    first_pass = np.concatenate([next(it)[0]["user_id"] for _ in range(n // bs)])
    ref = ml_100k.synthetic_columns(n, 3)[0]["user_id"]
    assert sorted(first_pass.tolist()) == sorted(ref.tolist())


def test_model_fn_errors_and_params(device):
    cols = ml_100k.get_feature_columns(4)["linear"]
    with pytest.raises(ValueError, match="At least 1 feature column"):
        deep_fm.model_fn({}, None, "train", {"device": device})
    with pytest.raises(ValueError, match="At least 1 of linear, mf or dnn"):
        deep_fm.model_fn({}, None, "train", {"categorical_columns": cols, "use_linear": False, "use_mf": False,
                                             "use_dnn": False, "device": device})
    with pytest.raises(KeyError):
        model_utils.get_optimizer("Nadam")
    assert model_utils.get_optimizer().name == "Adam" and model_utils.get_optimizer().lr == 0.001


def test_model_fn_three_modes(device, data):
    cols = ml_100k.get_feature_columns(4)["linear"]
    params = {"categorical_columns": cols, "device": device}            # every other key at its default
    feats, labels = next(ml_100k.get_input_fn(str(data / "train.csv"), batch_size=32, seed=0)())
    l0 = float(deep_fm.model_fn(feats, labels, "eval", params).loss)
    for _ in range(30):
        spec = deep_fm.model_fn(feats, labels, "train", params)
    assert float(spec.loss) < l0 and spec.train_op == 30
    eng = params["_store"]["engine"]
    assert (eng.E, eng.hidden, eng.dropout, eng.opt.name, eng.opt.lr) == (4, [16, 16], 0.0, "Adam", 0.001)
    pr = deep_fm.model_fn(feats, None, "infer", params).predictions
    assert set(pr) == {"logits", "logistic", "probabilities", "class_ids", "classes"}
    assert pr["probabilities"].shape == (32, 2) and torch.allclose(pr["probabilities"].sum(1), torch.ones(32, device=pr["logits"].device))


@pytest.mark.parametrize("trainer,extra", [(deep_fm, ["--hidden-units", "8", "8", "--dropout", "0.1"]), (linear, []),
                                           (deep, ["--hidden-units", "8"]), (linear_deep, ["--hidden-units", "8"])])
def test_train_and_evaluate_end_to_end(device, data, trainer, extra, capsys):
    name = trainer.__name__.split(".")[-1]
    opt = {"deep_fm": ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout"), "linear": (),
           "deep": ("hidden_units", "dropout"), "linear_deep": ("hidden_units", "dropout")}[name]
    job = str(data / "job")
    argv = ["--train-csv", str(data / "train.csv"), "--test-csv", str(data / "test.csv"), "--job-dir", job,
            "--train-steps", "120", "--batch-size", "32", "--device", device] + extra
    est = trainer.train_and_evaluate(_cli.make_parser(name, opt).parse_args(argv))
    assert est.global_step == 120
    out = capsys.readouterr().out
    assert "Saving dict for global step 120" in out and "auc = " in out and "average_loss = " in out
    assert os.path.exists(os.path.join(job, "model.ckpt-120.pt")) and os.listdir(os.path.join(job, "export", "exporter"))
    import json as _json                                   # layer_summary records every save_summary_steps (100)
    recs = [_json.loads(l) for l in open(os.path.join(job, "summaries.jsonl"))]
    assert recs and recs[0]["global_step"] == 100 and "logits" in recs[0]["layers"]
    if name == "deep_fm":
        assert {"linear/logits", "mf/logits", "dnn/hiddenlayer_0", "dnn/logits"} <= set(recs[0]["layers"])
        assert 0.0 <= recs[0]["layers"]["dnn/hiddenlayer_0"]["fraction_of_zero_values"] <= 1.0
    # --restore continues from the checkpoint; without it the job dir is wiped (deep_fm.py:147-148)
    est2 = trainer.train_and_evaluate(_cli.make_parser(name, opt).parse_args(argv[:-len(extra) or None] + extra + ["--restore", "--train-steps", "150"]))
    assert est2.global_step == 150 and "restored" in capsys.readouterr().out
    m = est2.evaluate(ml_100k.get_input_fn(str(data / "test.csv"), "eval", 32))
    assert {"accuracy", "auc", "auc_precision_recall", "average_loss", "loss", "precision", "recall", "label/mean",
            "prediction/mean", "accuracy_baseline", "global_step"} <= set(m)
    assert 0.5 < m["auc"] <= 1.0                                         # the synthetic rule is learnable
    first = next(est2.predict(ml_100k.get_input_fn(str(data / "test.csv"), "eval", 32)))
    assert first["probabilities"].shape == (2,)


def test_grouped_id_transforms_leave_the_training_unchanged(device, data, monkeypatch):
    """Estimator.train draws small batches 2,048 examples at a time and has their ids transformed in one call (the host
    cost of a 32-example step): the batches, their order and the trained variables are those of the batch-by-batch loop."""
    from mi355x_rec.estimator import Estimator
    fixed = lambda path, mode="train", batch_size=32, seed=None: ml_100k.get_input_fn(path, mode, batch_size=batch_size, seed=5)
    monkeypatch.setattr(_cli, "get_input_fn", fixed)             # (one process shuffles with a fresh seed per run)
    finals = []
    for rows in (Estimator.GROUP_ROWS, 0):
        monkeypatch.setattr(Estimator, "GROUP_ROWS", rows)
        job = str(data / ("job_g%d" % rows))
        argv = ["--train-csv", str(data / "train.csv"), "--test-csv", str(data / "test.csv"), "--job-dir", job,
                "--train-steps", "70", "--batch-size", "32", "--device", device, "--hidden-units", "8", "8", "--dropout", "0.1"]
        opt = ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout")
        est = deep_fm.train_and_evaluate(_cli.make_parser("deep_fm", opt).parse_args(argv))
        assert est.global_step == 70
        finals.append(est._engine().export_numpy())
    a, b = finals
    assert all(np.array_equal(x, y) for x, y in zip(a["emb"], b["emb"])) and all(np.array_equal(x, y) for x, y in zip(a["lin_w"], b["lin_w"]))
    assert all(np.array_equal(ka, kb) and np.array_equal(ba, bb) for (ka, ba), (kb, bb) in zip(a["mlp"], b["mlp"]))


@pytest.mark.gpu
def test_lookahead_of_large_batches_leaves_the_training_unchanged(data, monkeypatch):
    """Estimator.train holds the NEXT batch from 4,096 examples on: its ids are transformed and copied a step early and
    announced to the engine (whose sort of them then runs beside this step's catch-up).  Batches, their order, the loss of
    every step and the trained variables are those of the plain loop, bit for bit; the engine really took the sorts made
    ahead."""
    from mi355x_rec.estimator import Estimator
    from mi355x_rec.engine import DeepFM
    fixed = lambda path, mode="train", batch_size=32, seed=None: ml_100k.get_input_fn(path, mode, batch_size=batch_size, seed=5)
    monkeypatch.setattr(_cli, "get_input_fn", fixed)
    taken = []
    orig = DeepFM._take_presorted
    monkeypatch.setattr(DeepFM, "_take_presorted", lambda self, ids: (lambda ps: (taken.append(ps is not None), ps)[1])(orig(self, ids)))
    finals = []
    for min_batch in (4096, 1 << 30):
        monkeypatch.setattr(Estimator, "LOOKAHEAD_MIN_BATCH", min_batch)
        taken.clear()
        job = str(data / ("job_la%d" % min_batch))
        argv = ["--synthetic", "40960", "--job-dir", job, "--train-steps", "12", "--batch-size", "4096", "--device", "cuda",
                "--hidden-units", "32", "16", "--dropout", "0.1", "--embedding-size", "32"]
        opt = ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout")
        est = deep_fm.train_and_evaluate(_cli.make_parser("deep_fm", opt).parse_args(argv))
        assert est.global_step == 12
        finals.append((est._engine().export_numpy(), sum(taken)))
    (a, hits_a), (b, hits_b) = finals
    assert hits_a >= 8 and hits_b == 0, (hits_a, hits_b)
    for x, y in zip(a["emb"] + a["lin_w"], b["emb"] + b["lin_w"]):
        assert np.array_equal(x, y)
    for (ka, ba), (kb, bb) in zip(a["mlp"], b["mlp"]):
        assert np.array_equal(ka, kb) and np.array_equal(ba, bb)


def test_warm_start_from_tf_named_variables(device, data, capsys):
    """--warm-start-from: a dump of TensorFlow-named variables (the reference's checkpoint scopes) seeds a
    fresh job; evaluating before any further step reproduces the donor's metrics."""
    import numpy as np
    from mi355x_rec import tf_names
    opt = ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout")
    base = ["--train-csv", str(data / "train.csv"), "--test-csv", str(data / "test.csv"), "--batch-size", "32", "--device",
            device, "--hidden-units", "8", "--dropout", "0.0"]
    donor = deep_fm.train_and_evaluate(_cli.make_parser("deep_fm", opt).parse_args(base + ["--job-dir", str(data / "a"), "--train-steps", "60"]))
    store = donor.params["_store"]
    names = [c.name for c in store["plan"].categorical]
    dump = tf_names.export_variables(store["engine"], names)
    assert "input_layer/input_layer/user_id_embedding/embedding_weights" in dump and "linear/linear_model/zipcode/weights" in dump
    np.savez(str(data / "vars.npz"), **dump)
    ev = ml_100k.get_input_fn(str(data / "test.csv"), "eval", 32)
    ref = donor.evaluate(ev)
    capsys.readouterr()
    warm = deep_fm.train_and_evaluate(_cli.make_parser("deep_fm", opt).parse_args(
        base + ["--job-dir", str(data / "b"), "--train-steps", "0", "--warm-start-from", str(data / "vars.npz")]))
    assert "warm-started" in capsys.readouterr().out and warm.global_step == 0
    got = warm.evaluate(ev)
    assert abs(got["average_loss"] - ref["average_loss"]) < 1e-6 and abs(got["auc"] - ref["auc"]) < 1e-6


def test_model_utils_helpers(device):
    if device == "cpu":
        pytest.skip("helpers bind the HIP kernels directly")
    x = torch.tensor([[-2.0], [0.0], [3.0]], device="cuda")
    y = torch.tensor([0, 1, 1], device="cuda")
    pr = model_utils.get_binary_predictions(x)
    assert set(pr) == {"logits", "logistic", "probabilities", "class_id"} and pr["class_id"].flatten().tolist() == [0, 0, 1]
    ls = model_utils.get_binary_losses(y, pr)
    assert set(ls) == {"unreduced_loss", "average_loss", "loss"} and ls["loss"].item() == pytest.approx(3 * ls["average_loss"].item())
    mt = model_utils.get_binary_metric_ops(y, pr, ls)
    assert set(mt) == {"accuracy", "auc", "auc_precision_recall", "average_loss"} and mt["accuracy"] == pytest.approx(2 / 3)
    s = model_utils.layer_summary(torch.tensor([[0.0, 1.0], [2.0, 0.0]], device="cuda"))
    assert s["fraction_of_zero_values"] == 0.5 and s["max"] == 2.0
    assert s["activation"]["num"] == 4 and s["activation"]["bucket"] == [2, 1, 1] and s["activation"]["sum"] == 3.0   # tf.summary.histogram
    # get_train_op (model_utils.py:69-72): the callable that runs one minimize() step on an engine built with that optimizer
    from mi355x_rec.engine import DeepFM
    opt = model_utils.get_optimizer("Adagrad", 0.05)
    op = model_utils.get_train_op(ls["loss"], opt)
    eng = DeepFM([5, 4], embedding_size=4, hidden_units=[8], optimizer=opt, device="cuda")
    eng.init_variables()
    ids = torch.tensor([[1, 2], [3, 0], [4, 3]], dtype=torch.int32, device="cuda")
    loss, logits = op(eng, ids, torch.tensor([0, 1, 1], dtype=torch.uint8, device="cuda"))
    assert eng.step == 1 and np.isfinite(loss.item())
    with pytest.raises(ValueError):
        model_utils.get_train_op(ls["loss"], model_utils.get_optimizer("SGD", 0.1))(eng, ids, None)


def _cli_rank(rank, world, port, job_dir, q, steps=6, interval=None):
    """one process of a 2-rank `python -m torch.distributed.run ... -m trainers.deep_fm --synthetic` launch on CPU.
    interval: checkpoint / evaluation interval in seconds (conf_utils.EVAL_INTERVAL), None = the reference's 60."""
    import sys, time, traceback
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "recommender-tensorflow_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        from mi355x_rec import engine
        from tests.cpu_kernels import NumpyKernels
        engine.HipKernels = NumpyKernels                      # host logic run: kernels stood in by numpy
        from trainers import deep_fm as T, _cli, conf_utils
        from mi355x_rec import estimator as est_mod
        n_eval = [0]
        if interval is not None:
            conf_utils.EVAL_INTERVAL = interval
            _orig_eval = est_mod.Estimator.evaluate
            _orig_step = engine.DeepFM.train_step

            def counting(self, *a, **kw):
                n_eval[0] += 1
                return _orig_eval(self, *a, **kw)

            def skewed(self, *a, **kw):                       # the ranks' clocks and paces differ
                time.sleep(0.004 * (1 + rank))
                return _orig_step(self, *a, **kw)
            est_mod.Estimator.evaluate = counting
            engine.DeepFM.train_step = skewed
            time.sleep(0.03 * rank)
        args = _cli.make_parser("deep_fm", ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout")).parse_args(
            ["--job-dir", job_dir, "--synthetic", "400", "--train-steps", str(steps), "--batch-size", "16", "--device", "cpu",
             "--world-size", str(world), "--dropout", "0"])
        est = T.train_and_evaluate(args)
        eng = est._engine()
        q.put((rank, "ok", est.global_step, eng.R_local, eng.dense.clone().numpy(), sorted(os.listdir(job_dir)), n_eval[0]))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc(), None, None, None, None))


def test_cli_two_rank_launch_cpu(tmp_path):
    """the trainers read RANK / WORLD_SIZE of a torch.distributed.run launch, shard the tables by
    row, train synchronously and checkpoint per rank; --synthetic replaces the CSV files."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    job = str(tmp_path / "job")
    procs = [ctx.Process(target=_cli_rank, args=(r, 2, port, job, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=240)
        assert r[1] == "ok", r[2]
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    assert res[0][2] == res[1][2] == 6                               # one global step counter, synchronous
    assert res[0][3] + res[1][3] == 4106 and abs(res[0][3] - res[1][3]) <= 1   # rows split r % 2
    assert np.array_equal(res[0][4], res[1][4])                      # replicated dense variables stay identical
    files = res[0][5]
    assert "model.ckpt-6.rank0.pt" in files and "model.ckpt-6.rank1.pt" in files and "checkpoint.rank1.json" in files
    with pytest.raises(SystemExit):
        os.environ.pop("WORLD_SIZE", None)
        _cli.init_distributed(_cli.make_parser("linear").parse_args(["--world-size", "4"]))


def test_two_rank_run_checkpoints_and_evaluates_at_the_same_steps(tmp_path):
    """with N processes the checkpoint / evaluation cadence must not come from each rank's own clock —
    evaluate() is a collective in the row-sharded engine, and a rank that enters it while the other issues the next train
    step's all_to_all hangs the run.  Two gloo ranks of different pace, a 50 ms interval, 60 steps: several mid-run
    evaluations, the same number on both ranks, the same final step, identical dense variables."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    job = str(tmp_path / "job")
    procs = [ctx.Process(target=_cli_rank, args=(r, 2, port, job, q, 60, 0.05)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        assert r[1] == "ok", r[2]
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    assert res[0][2] == res[1][2] == 60
    assert res[0][6] == res[1][6] and res[0][6] >= 3, (res[0][6], res[1][6])
    assert np.array_equal(res[0][4], res[1][4])
    exports = [d for d in os.listdir(os.path.join(job, "export", "exporter"))]
    assert exports and all(sorted(os.listdir(os.path.join(job, "export", "exporter", d))) ==
                           ["signature.json", "variables.rank0.pt", "variables.rank1.pt"] for d in exports)


def test_synthetic_input():
    cols, n = ml_100k.synthetic_columns(50, seed=3)
    assert n == 50 and set(cols) == set(ml_100k.COLUMNS)
    f, l = next(ml_100k.get_input_fn("synthetic:64:1", "eval", batch_size=32)())
    assert len(l) == 32 and f["user_id"].dtype == np.int32 and f["gender"][0] in ("F", "M")
