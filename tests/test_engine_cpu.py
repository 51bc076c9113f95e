"""CPU tests of the host logic: the REAL engine.py orchestration (buffer layout, step sequencing,
Adam schedule, lazy catch-up bookkeeping, checkpoint state) driven with numpy stand-ins for the
device entry points (tests/cpu_kernels.py), checked against the oracle.  The HIP kernels themselves
are covered by the -m gpu tests."""
import numpy as np
import pytest
import torch

from oracle import deepfm as O
from oracle import optimizers as OO
from tests.cpu_kernels import NumpyKernels
from tests.util import make_problem


def _engine(vocab, E, hidden, nn=0, **kw):
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    opt = kw.pop("optimizer", OptimizerSpec("Adam", 0.001))
    return DeepFM(vocab, n_numeric=nn, embedding_size=E, hidden_units=hidden, optimizer=opt, device="cpu",
                  _kernels=NumpyKernels(), **kw)


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


def _check_vars(m, p, atol):
    g = m.export_numpy()
    for f in range(len(p.emb)):
        assert np.max(np.abs(g["emb"][f] - p.emb[f])) < atol
        assert np.max(np.abs(g["lin_w"][f] - p.lin_w[f])) < atol
    for i, (k, b) in enumerate(g["mlp"]):
        assert np.max(np.abs(k - p.mlp[i][0])) < atol and np.max(np.abs(b - p.mlp[i][1])) < atol
    assert abs(g["lin_bias"][0] - p.lin_bias[0]) < atol


@pytest.mark.parametrize("vocab,E,hidden,B,nn", [([9, 13, 5, 6], 8, [16, 8], 64, 0), ([11, 5, 9], 4, [12], 33, 2)])
def test_engine_orchestration_matches_oracle(vocab, E, hidden, B, nn):
    p, ids, x, y = make_problem(7, vocab, E, hidden, B, n_numeric=nn)
    m = _engine(vocab, E, hidden, nn)
    m.GAP_SORT_MIN = 1 if nn else m.GAP_SORT_MIN      # one case goes through the sort-rows-by-staleness path
    m.load_oracle_params(p)
    st = O.TrainState(p, OO.Hyper("Adam", 0.001))
    rng = np.random.default_rng(1)
    for step in range(4):
        ids_s = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
        ids_s[1] = ids_s[0]
        lo, logit_o = O.train_step(p, st, ids_s, y, x)
        lg, logit_g = m.train_step(_t(ids_s), _t(y), _t(x))
        assert abs(lg.item() - float(lo)) < 2e-6 * abs(float(lo)) + 1e-7
        assert np.allclose(logit_g.numpy(), logit_o, rtol=1e-5, atol=1e-6)
    _check_vars(m, p, 1e-6)
    # eval path leaves variables alone and agrees with the oracle forward
    loss, logits = m.loss(_t(ids), _t(y), _t(x))
    c = O.forward(p, ids, x)
    assert np.allclose(logits.numpy(), c["logits"], rtol=1e-5, atol=1e-6)


def test_state_dict_roundtrip_resumes_bitwise():
    vocab, E, hidden, B = [9, 13, 5], 4, [8], 32
    p, ids, x, y = make_problem(8, vocab, E, hidden, B)
    a = _engine(vocab, E, hidden)
    a.load_oracle_params(p)
    for _ in range(2):
        a.train_step(_t(ids), _t(y))
    sd = a.state_dict()
    b = _engine(vocab, E, hidden)
    b.load_state_dict(sd)
    la, _ = a.train_step(_t(ids), _t(y))
    lb, _ = b.train_step(_t(ids), _t(y))
    assert la.item() == lb.item() and b.step == 3
    assert torch.equal(a.table, b.table) and torch.equal(a.dense, b.dense)


def test_checkpoint_of_another_layout_is_refused(tmp_path):
    """a checkpoint is a set of flat buffers; before anything is copied its format version, the layout
    table (model shape + where every dense variable sits) and every tensor's shape must match — a checkpoint of another
    hidden / embedding size with the SAME parameter count, of an older layout, or of another optimizer is an error,
    not a silent permutation.  The file goes through torch.save / torch.load(weights_only=True) like the Estimator's."""
    a = _engine([9, 13, 5], 4, [8])
    sd = a.state_dict()
    path = str(tmp_path / "m.pt")
    torch.save(sd, path)
    sd = torch.load(path, weights_only=True)
    assert sd["format"] == a.STATE_FORMAT and sd["layout"]["segments"]["kernel_0"] == [0, 12, 8]
    _engine([9, 13, 5], 4, [8]).load_state_dict(sd)                             # the same model: fine
    b = _engine([9, 13, 5], 4, [4, 8])                                          # same P (16-float aligned segments), other layers
    assert b.P == a.P
    with pytest.raises(ValueError, match="hidden_units"):
        b.load_state_dict(sd)
    with pytest.raises(ValueError, match="vocab_sizes"):
        _engine([9, 13, 6], 4, [8]).load_state_dict(sd)
    from mi355x_rec.engine import OptimizerSpec
    with pytest.raises(ValueError, match="optimizer"):
        _engine([9, 13, 5], 4, [8], optimizer=OptimizerSpec("Adagrad", 0.05)).load_state_dict(sd)
    old = {k: v for k, v in sd.items() if k not in ("format", "layout")}        # what rounds 1-2 wrote
    with pytest.raises(ValueError, match="format"):
        _engine([9, 13, 5], 4, [8]).load_state_dict(old)
    bad = dict(sd); bad["dense"] = sd["dense"][:-16]
    with pytest.raises(ValueError, match="dense"):
        _engine([9, 13, 5], 4, [8]).load_state_dict(bad)
    assert a._alloc_gen >= 0 and a._graph_gen()[1] == a.sched.gen


def test_input_validation():
    m = _engine([3, 4], 4, [8])
    with pytest.raises(ValueError, match="int32"):
        m.train_step(torch.zeros(4, 2, dtype=torch.int64), torch.zeros(4, dtype=torch.uint8))
    with pytest.raises(ValueError, match="labels"):
        m.train_step(torch.zeros(4, 2, dtype=torch.int32), torch.zeros(4, dtype=torch.float32))
    with pytest.raises(ValueError, match="no numeric"):
        m.train_step(torch.zeros(4, 2, dtype=torch.int32), torch.zeros(4, dtype=torch.uint8), torch.zeros(4, 1))


def test_optimizer_spec_defaults_and_unknown_name():
    from mi355x_rec.engine import OptimizerSpec
    with pytest.raises(KeyError):
        OptimizerSpec("Nadam")                      # model_utils.py:65: dict lookup -> KeyError
    assert OptimizerSpec("Adam").epsilon == 1e-8 and OptimizerSpec("RMSProp").epsilon == 1e-10
    assert OptimizerSpec("Ftrl").slot_init == (0.1, 0.0) and OptimizerSpec("Adagrad").slot_init == (0.1, None)


def test_adam_schedule_is_tf_running_product():
    from mi355x_rec.engine import AdamSchedule, OptimizerSpec
    s = AdamSchedule(OptimizerSpec("Adam", 0.001), "cpu", capacity=8)
    pw = OO.AdamPowers(OO.Hyper("Adam", 0.001), np.float32)
    for step in range(1, 40):                       # crosses the capacity: table must extend itself
        assert s.lr_t(step) == float(pw.lr_t(0.001))
        pw.finish()
    assert float(s.table[39]) == s.lr_t(39)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mi355x_rec import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    from mi355x_rec.engine import DeepFM
    with pytest.raises(_lib.MiError, match="no fallback"):
        DeepFM([3, 4], device="cpu")


def test_tf_named_variables_round_trip():
    """mi355x_rec/tf_names.py: engine -> {TF-1.12 variable name: array} -> a fresh engine gives the same
    logits; names and shapes are the reference's scopes; a missing variable is an error."""
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    from mi355x_rec import tf_names
    from tests.cpu_kernels import NumpyKernels
    vocab, cols = [5, 7, 3], ["age_bucketized", "gender", "item_id"]
    p, ids, x, y = make_problem(2, vocab, 4, [8, 6], 16, n_numeric=2)
    mk = lambda: DeepFM(vocab, n_numeric=2, embedding_size=4, hidden_units=[8, 6], optimizer=OptimizerSpec("Adam", 0.001),
                        device="cpu", _kernels=NumpyKernels())
    a = mk()
    a.load_oracle_params(p)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v))
    a.train_step(t(ids), t(y), t(x))                       # variables move, Adam state becomes non-trivial
    dump = tf_names.export_variables(a, cols)
    assert dump["input_layer/input_layer/gender_embedding/embedding_weights"].shape == (7, 4)
    assert dump["linear/linear_model/item_id/weights"].shape == (3, 1)
    assert dump["linear/linear_model/bias_weights"].shape == (1,)
    assert dump["dnn/dnn/hiddenlayer_1/dense/kernel"].shape == (8, 6)
    assert dump["dnn/dnn/logits/dense/bias"].shape == (1,)
    assert dump["input_layer/numeric_embeddings"].shape == (1, 2, 4)
    b = mk()
    loaded = tf_names.import_variables(b, dump, cols)
    assert sorted(loaded) == sorted(dump)
    assert b.step == 0
    la, lb = a.predict_logits(t(ids), t(x)), b.predict_logits(t(ids), t(x))
    assert np.array_equal(la.numpy(), lb.numpy())
    del dump["dnn/dnn/logits/dense/kernel"]
    with pytest.raises(KeyError, match="logits/dense/kernel"):
        tf_names.import_variables(mk(), dump, cols)
    assert "dnn/hiddenlayer_0/kernel" in str(tf_names.variable_names("dnn", cols, 2)["mlp"])


def test_chunk_count_follows_batch_and_world_size():
    """parallel._n_chunks: a chunk is the M of every MLP GEMM — at least 32768 examples (256 CUs x 128-row tiles) at EVERY world
    size (rounds 3-5 took 16384 up to 4 ranks by arithmetic; with modelled link time 2 chunks of 32768 are ahead at 2, 4 and 8
    ranks: profiles/r05_sim_ranks.md); an explicit RowShard(chunks=...) wins; evaluation never chunks.  And every chunk runs
    its own forward / backward unless told otherwise (the same rehearsal: 4.63 against 5.38 ms at 8 ranks)."""
    from types import SimpleNamespace
    from mi355x_rec.parallel import RowShard, _n_chunks
    m = lambda world, chunks=None: SimpleNamespace(shard=RowShard(0, world, chunks=chunks))
    assert _n_chunks(m(8), 65536, True) == 2 and _n_chunks(m(8), 131072, True) == 4 and _n_chunks(m(8), 16384, True) == 1
    assert _n_chunks(m(2), 65536, True) == 2 and _n_chunks(m(4), 65536, True) == 2 and _n_chunks(m(4), 32768, True) == 1 and _n_chunks(m(4), 131072, True) == 4
    assert RowShard(0, 2).chunk_compute and RowShard(0, 8).chunk_compute and not RowShard(0, 8, chunk_compute=False).chunk_compute
    assert _n_chunks(m(8), 4096, True) == 2 and _n_chunks(m(8), 1000, True) == 1           # small batches: the old rule
    assert _n_chunks(m(8, chunks=4), 65536, True) == 4 and _n_chunks(m(8, chunks=3), 65536, True) == 2   # (a divisor of B)
    assert _n_chunks(m(8), 65536, False) == 1


def test_weights_are_split_once_per_step():
    """engine._split_weights runs once per (step, state of the dense buffer): the chunks of a pipelined multi-GPU step
    share the planes of the weights; an apply (step + 1) or a torch-side write to the buffer makes them stale."""
    calls = []

    class K(NumpyKernels):
        def mi_absmax(self, *a):
            calls.append("absmax")

        def mi_split_weights(self, *a):
            calls.append("split")
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    m = DeepFM([5, 4], embedding_size=4, hidden_units=[8, 8], optimizer=OptimizerSpec("Adam", 0.001), device="cpu", _kernels=K())
    import mi355x_rec._lib as L
    m._planes = lambda name, rows, K_: L.Planes()            # (no device planes on the CPU: the bookkeeping is what is tested)
    m._av = lambda name: None
    m._split_weights(True); m._split_weights(True); m._split_weights(False)
    assert calls == ["absmax", "split"]
    m.step += 1
    m._split_weights(False)
    assert calls == ["absmax", "split"] * 2
    m._split_weights(True)                                     # an eval split does not cover the train-only orientation
    assert calls == ["absmax", "split"] * 3
    m.dense.add_(0)
    m._split_weights(True)
    assert calls == ["absmax", "split"] * 4
