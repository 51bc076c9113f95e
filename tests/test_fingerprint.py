"""CPU: categorical id transforms (trainers/ml_100k.py:19-35) — oracle vs known answers, and the C ABI
host entries vs the oracle (integer work: bit exact).

Known answers.  The reference holds none (it has no tests); TensorFlow 1.12 (un-vendored dependency)
is absent.  Anchors used instead, all recalled from TensorFlow's own public tests / docs:
  * string_to_hash_bucket_fast op test: 'a','b','c','d' mod 10 -> 9,2,2,5 (the full 64-bit values
    below are the Fingerprint64 results those buckets come from)           [1-3 byte branch]
  * tf.strings.to_hash_bucket_fast(["Hello","TensorFlow","2.x"], 3) -> [0, 2, 2] (API docs example)
    [4-7 byte and 8-16 byte branches, weakly: mod 3]
  * round 5 — full 64-bit answers for the 4-7 and 8-16 byte branches (every MovieLens item id, zipcode and occupation
    goes through one of them): BigQuery's FARM_FINGERPRINT is farmhash Fingerprint64 — the function TensorFlow's
    string_to_hash_bucket_fast calls — and its reference page shows FARM_FINGERPRINT(CONCAT(CAST(x AS STRING), y,
    CAST(z AS STRING))) for the rows (1, "foo", true), (2, "apple", false), (3, "", true) as the signed values
    -1541654101129638711, 2794438866806483259, -4880158226897771312: "1footrue" (8 bytes), "2applefalse" (11 bytes),
    "3true" (5 bytes).  Recalled, like the anchors above; three independent 64-bit values that a restatement written
    from the published algorithm reproduces are not a coincidence.
The 17+ byte branches have no external anchor here: the C and Python restatements (written
independently from FarmHash's published algorithm) are cross-checked against each other only.
"""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import columns as OC
from oracle.fingerprint import fingerprint64

KAT = {"a": 12917804110809363939, "b": 11795596070477164822, "c": 11430444447143000872,
       "d": 4470636696479570465,
       "abc": 2640714258260161385}     # (pyfarmhash's README: farmhash.hash64('abc') — Hash64 and Fingerprint64 agree below 17 bytes)
# BigQuery FARM_FINGERPRINT documentation rows (signed int64): the 8-16 byte branch twice, the 4-7 byte branch once
KAT_SIGNED = {"1footrue": -1541654101129638711, "2applefalse": 2794438866806483259, "3true": -4880158226897771312}


def test_fingerprint64_known_answers():
    for s, v in KAT.items():
        assert fingerprint64(s) == v
    assert [fingerprint64(s) % 10 for s in "abcd"] == [9, 2, 2, 5]
    assert [fingerprint64(s) % 3 for s in ["Hello", "TensorFlow", "2.x"]] == [0, 2, 2]
    assert fingerprint64("") == 0x9AE16A3B2F90404F      # empty string returns k2
    for s, v in KAT_SIGNED.items():
        assert fingerprint64(s) == v % (1 << 64), s


def test_c_abi_fingerprint_matches_oracle_on_every_length_branch(lib):
    rnd = random.Random(7)
    for n in list(range(0, 200)) + [255, 256, 257, 1000, 4097]:
        s = bytes(rnd.randrange(256) for _ in range(n))
        assert lib.mi_fingerprint64(s, len(s)) == fingerprint64(s), n
    for s, v in KAT.items():
        assert lib.mi_fingerprint64(s.encode(), len(s)) == v
    for s, v in KAT_SIGNED.items():
        assert lib.mi_fingerprint64(s.encode(), len(s)) == v % (1 << 64), s


def test_hash_bucket_columns(lib):
    ids = np.array([1, 42, 943, 1682, 0, -7, 10 ** 12], np.int64)          # user/item ids: decimal ASCII
    out = np.empty(len(ids), np.int32)
    assert lib.mi_hash_bucket_i64(ids.ctypes.data, len(ids), 1000, out.ctypes.data) == 0
    assert np.array_equal(out, OC.hash_bucket(ids, 1000))
    assert out[0] == fingerprint64("1") % 1000 and out[5] == fingerprint64("-7") % 1000
    strs = ["technician", "administrator", "homemaker", "none", "", "85711", "T8H1N", "x" * 70]
    blob = "".join(strs).encode()
    offs = np.cumsum([0] + [len(s.encode()) for s in strs]).astype(np.int64)
    out = np.empty(len(strs), np.int32)
    assert lib.mi_hash_bucket_bytes(blob, offs.ctypes.data, len(strs), 50, out.ctypes.data) == 0
    assert np.array_equal(out, OC.hash_bucket(strs, 50))
    assert lib.mi_hash_bucket_i64(ids.ctypes.data, len(ids), 0, out.ctypes.data) < 0      # bad bucket count
    assert b"buckets" in lib.mi_last_error()


def test_bucketized_columns(lib):
    age_b = list(range(15, 66, 10))                     # ml_100k.py:24
    year_b = list(range(1930, 1991, 10))                # ml_100k.py:34
    assert OC.bucketize([14, 15, 24, 25, 65, 70], age_b).tolist() == [0, 1, 1, 2, 6, 6]
    assert OC.bucketize([1929, 1930, 1990, 2000, 0], year_b).tolist() == [0, 1, 7, 7, 0]
    x = np.array([14, 15, 24.999, 25, 65, 70, -1, 1e9], np.float32)
    b = np.array(age_b, np.float32)
    out = np.empty(len(x), np.int32)
    assert lib.mi_bucketize_f32(x.ctypes.data, len(x), b.ctypes.data, len(b), out.ctypes.data) == 0
    assert np.array_equal(out, OC.bucketize(x, age_b))
    bad = np.array([3, 2, 1], np.float32)
    assert lib.mi_bucketize_f32(x.ctypes.data, len(x), bad.ctypes.data, 3, out.ctypes.data) < 0


def test_vocabulary_and_identity_columns():
    assert OC.vocabulary_list(["F", "M", "null", b"F", "x"], ["F", "M"], 1).tolist() == [0, 1, 2, 0, 2]
    assert OC.identity([0, 1, 1, 0], 2).tolist() == [0, 1, 1, 0]
    with pytest.raises(ValueError):
        OC.identity([0, 2], 2)


def test_ml100k_schema_sorted_order():
    f = OC.ml100k_fields()
    assert len(f) == 26 and sum(v for *_, v in f) == 4106
    names = [t[0] for t in OC.sorted_fields(f)]
    assert names == ["action", "adventure", "age_bucketized", "animation", "children", "comedy", "crime",
                     "documentary", "drama", "fantasy", "filmnoir", "gender", "horror", "item_id", "musical",
                     "mystery", "occupation", "release_year_bucketized", "romance", "scifi", "thriller", "unknown",
                     "user_id", "war", "western", "zipcode"]
