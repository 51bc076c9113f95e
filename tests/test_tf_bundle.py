"""TensorFlow tensor-bundle checkpoints in pure Python (mi355x_rec/tf_bundle.py).  PARITY UNPINNED: no TF-written bundle
exists in this container — the reader is exercised on bundles the writer half of the same module produced (multi-block
index tables, checksums, the `checkpoint` state file), on corrupted copies of them, and end to end through
--warm-start-from; the CRC-32C underneath is pinned by its published check value."""
import os

import numpy as np
import pytest

from mi355x_rec import tf_bundle as B


def test_crc32c_known_answers():
    # the CRC-32C (Castagnoli) check value of the catalogue of CRC algorithms, and RFC 3720's 32 zero / 32 0xff bytes
    assert B.crc32c(b"123456789") == 0xE3069283
    assert B.crc32c(bytes(32)) == 0x8A9136AA
    assert B.crc32c(b"\xff" * 32) == 0x62A8AB43
    # continuing from a running value == one pass; unaligned starts
    data = bytes(range(256)) * 5
    for cut in (0, 1, 7, 8, 9, 500):
        assert B.crc32c(data[cut:], B.crc32c(data[:cut])) == B.crc32c(data)
    assert B.crc32c(np.frombuffer(data, np.uint8)[3:]) == B.crc32c(data[3:])
    # TensorFlow / leveldb masking
    assert B.mask(0) == 0xa282ead8 and B.mask(0xE3069283) == ((0xE3069283 >> 15 | 0xE3069283 << 17) + 0xa282ead8) & 0xffffffff


def _tensors(rng, n):
    out = {"global_step": np.asarray(1234, np.int64), "dnn/dnn/logits/dense/bias": rng.standard_normal(1).astype(np.float32)}
    for i in range(n):
        shape = [(7, 4), (3,), (2, 3, 5), ()][i % 4]
        out["input_layer/input_layer/col_%03d_embedding/embedding_weights" % i] = rng.standard_normal(shape).astype(np.float32)
    out["some/int32"] = rng.integers(-5, 5, (4, 2)).astype(np.int32)
    out["some/empty"] = np.zeros((0, 4), np.float32)
    return out


def test_bundle_round_trip_with_a_multi_block_index(tmp_path):
    rng = np.random.default_rng(0)
    t = _tensors(rng, 400)                                   # 400 prefix-compressed entries: several 4-KB index blocks
    prefix = B.write_bundle(str(tmp_path / "model.ckpt-1234"), t)
    assert os.path.getsize(prefix + ".index") > 3 * 4096
    assert sorted(os.listdir(tmp_path)) == ["checkpoint", "model.ckpt-1234.data-00000-of-00001", "model.ckpt-1234.index"]
    got = B.read_bundle(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape and np.array_equal(got[k], t[k]), k
    # by model_dir (its `checkpoint` file), by .index file, a subset of names, shapes without data
    one = "input_layer/input_layer/col_007_embedding/embedding_weights"
    assert np.array_equal(B.read_bundle(str(tmp_path), names=[one])[one], t[one])
    assert set(B.read_bundle(prefix + ".index", names=[one])) == {one}
    assert B.list_variables(str(tmp_path))["some/int32"] == (np.int32, (4, 2))
    assert B.is_bundle(str(tmp_path)) and not B.is_bundle(str(tmp_path / "nope"))
    with pytest.raises(KeyError):
        B.read_bundle(prefix, names=["missing/variable"])


def test_corruption_is_detected(tmp_path):
    rng = np.random.default_rng(1)
    t = _tensors(rng, 40)
    prefix = B.write_bundle(str(tmp_path / "m"), t)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[17] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="checksum mismatch"):
        B.read_bundle(prefix)
    assert len(B.read_bundle(prefix, verify=False)) == len(t)          # (the same bytes, unchecked)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[100] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum mismatch in the table block"):
        B.read_bundle(prefix, verify=False)
    open(prefix + ".index", "wb").write(bytes(idx[:-1]) + b"\x00")
    with pytest.raises(ValueError, match="bad magic"):
        B.read_bundle(prefix)


def test_warm_start_from_a_tensorflow_checkpoint_directory(tmp_path, capsys):
    """--warm-start-from <model_dir>: the engine's variables written under their TensorFlow names (tf_names.export_variables)
    as a tensor bundle — plus the optimizer slots and global_step a real checkpoint also holds — seed a fresh run by name."""
    import torch
    from mi355x_rec import tf_names
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    from mi355x_rec.estimator import Estimator
    from mi355x_rec.feature_column import categorical_column_with_identity
    from mi355x_rec.model import run_batch
    from tests.cpu_kernels import NumpyKernels
    cols = [categorical_column_with_identity("b_col", 5), categorical_column_with_identity("a_col", 7)]

    def make(plan, dev, shard=None):
        return DeepFM(plan.vocab_sizes, embedding_size=4, hidden_units=[8], optimizer=OptimizerSpec("Adam", 0.001), device="cpu",
                      _kernels=NumpyKernels())

    def model_fn(features, labels, mode, params):
        return run_batch(features, labels, mode, params, make)
    feats = {"a_col": np.array([1, 3, 6, 0], np.int32), "b_col": np.array([0, 4, 2, 2], np.int32)}
    labels = np.array([1, 0, 0, 1])
    src = Estimator(model_fn, str(tmp_path / "src"), params={"categorical_columns": cols, "device": "cpu"})
    src.train(lambda: iter([(feats, labels)] * 3), steps=3)
    eng = src._engine()
    named = tf_names.export_variables(eng, [c.name for c in src.params["_store"]["plan"].categorical])
    extra = {k + "/Adam": np.zeros_like(v) for k, v in named.items()}
    extra["global_step"] = np.asarray(3, np.int64)
    B.write_bundle(str(tmp_path / "tf" / "model.ckpt-3"), {**named, **extra})
    dst = Estimator(model_fn, str(tmp_path / "dst"), params={"categorical_columns": cols, "device": "cpu"},
                    warm_start_from=str(tmp_path / "tf"))
    p_src = list(src.predict(lambda: iter([feats])))
    p_dst = list(dst.predict(lambda: iter([feats])))
    assert "warm-started" in capsys.readouterr().out
    assert np.allclose([r["logits"] for r in p_src], [r["logits"] for r in p_dst], rtol=0, atol=0)
    assert dst.global_step == 0 and torch.equal(dst._engine().table, eng.table)
