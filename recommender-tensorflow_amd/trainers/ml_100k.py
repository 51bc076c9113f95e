"""MovieLens-100k schema, feature columns and CSV input pipeline.

Counterpart of the reference's ``trainers/ml_100k.py``: ``get_feature_columns`` (:18-39),
``get_input_fn`` (:42-61) and ``serving_input_fn`` (:64-88) keep their names, arguments and
return shapes; columns are ``mi355x_rec.feature_column`` descriptors and the input_fn yields numpy
batches instead of a tf.data graph."""
import csv

import numpy as np

from mi355x_rec import feature_column as fc
from mi355x_rec.estimator import ModeKeys, ServingInputReceiver

# CSV layout written by the reference's offline ETL (src/data/ml_100k.py): 42 columns.
_RATING = ["user_id", "item_id", "rating", "timestamp", "datetime", "year", "month", "day", "week", "dayofweek"]
_USER = ["age", "gender", "occupation", "zipcode", "zipcode1", "zipcode2", "zipcode3"]
_ITEM = ["title", "release", "video_release", "imdb"]
GENRE = ["unknown", "action", "adventure", "animation", "children", "comedy", "crime", "documentary", "drama",
         "fantasy", "filmnoir", "horror", "musical", "mystery", "romance", "scifi", "thriller", "war", "western"]
COLUMNS = _RATING + _USER + _ITEM + GENRE + ["release_date", "release_year"]
LABEL_COL = "rating"
_STRING_COLS = {"datetime", "gender", "occupation", "zipcode", "zipcode1", "zipcode2", "zipcode3", "title",
                "release", "video_release", "imdb", "release_date"}
# tf.decode_csv record_defaults: [0] -> int32 column, ["null"] -> string column
DEFAULTS = [["null"] if c in _STRING_COLS else [0] for c in COLUMNS]


def get_feature_columns(embedding_size=4):
    """26 categorical columns ("linear") and their embedding columns ("deep")."""
    by_key = {
        "user_id": fc.categorical_column_with_hash_bucket("user_id", 1000, np.int32),
        "item_id": fc.categorical_column_with_hash_bucket("item_id", 2000, np.int32),
        "age": fc.bucketized_column(fc.numeric_column("age"), list(range(15, 66, 10))),
        "gender": fc.categorical_column_with_vocabulary_list("gender", ["F", "M"], num_oov_buckets=1),
        "occupation": fc.categorical_column_with_hash_bucket("occupation", 50),
        "zipcode": fc.categorical_column_with_hash_bucket("zipcode", 1000),
        "release_year": fc.bucketized_column(fc.numeric_column("release_year"), list(range(1930, 1991, 10))),
    }
    order = ["user_id", "item_id", "age", "gender", "occupation", "zipcode", "release_year"]
    linear = [by_key[k] for k in order] + [fc.categorical_column_with_identity(g, 2) for g in GENRE]
    return {"linear": linear, "deep": [fc.embedding_column(c, embedding_size) for c in linear]}


_OCCUPATIONS = ["administrator", "artist", "doctor", "educator", "engineer", "entertainment", "executive", "healthcare",
                "homemaker", "lawyer", "librarian", "marketing", "none", "other", "programmer", "retired", "salesman",
                "scientist", "student", "technician", "writer"]


def synthetic_columns(n, seed=0):
    """n synthetic examples with the MovieLens-100k schema and value ranges (943 users, 1682 items, ratings 1-5
    with a learnable dependence on two genre flags): what `--synthetic N` trains on when the CSV files of the
    reference's offline ETL (src/data/ml_100k.py: needs the network) are not at hand."""
    rng = np.random.default_rng(seed)
    cols = {}
    for name, default in zip(COLUMNS, DEFAULTS):
        cols[name] = np.zeros(n, np.int32) if isinstance(default[0], int) else np.array(["null"] * n, dtype=object)
    g = rng.integers(0, 2, (n, len(GENRE)))
    like = np.where(rng.random(n) < 0.85, g[:, 1] & (1 - g[:, 8]), rng.integers(0, 2, n))
    cols.update(user_id=rng.integers(1, 944, n).astype(np.int32), item_id=rng.integers(1, 1683, n).astype(np.int32),
                rating=np.where(like == 1, 5, rng.integers(1, 5, n)).astype(np.int32),
                age=rng.integers(7, 74, n).astype(np.int32),
                gender=np.array(rng.choice(["F", "M"], n), dtype=object),
                occupation=np.array(rng.choice(_OCCUPATIONS, n), dtype=object),
                zipcode=np.array(["%05d" % z for z in rng.integers(0, 99999, n)], dtype=object),
                release_year=rng.integers(1922, 1999, n).astype(np.int32))
    cols.update({k: g[:, j].astype(np.int32) for j, k in enumerate(GENRE)})
    return cols, n


def _read_csv(path):
    """Whole file -> dict of typed numpy columns (missing / empty fields take DEFAULTS).  "synthetic:N[:seed]"
    instead of a path: N generated examples (synthetic_columns)."""
    if isinstance(path, str) and path.startswith("synthetic:"):
        parts = path.split(":")
        return synthetic_columns(int(parts[1]), int(parts[2]) if len(parts) > 2 else 0)
    with open(path, newline="") as f:
        rd = csv.reader(f)
        next(rd, None)                                    # header (dataset.skip(1))
        rows = [r for r in rd if r]
    cols = {}
    for j, (name, default) in enumerate(zip(COLUMNS, DEFAULTS)):
        raw = [r[j] if j < len(r) else "" for r in rows]
        if isinstance(default[0], int):
            cols[name] = np.array([int(float(v)) if v != "" else default[0] for v in raw], np.int32)
        else:
            cols[name] = np.array([v if v != "" else default[0] for v in raw], dtype=object)
    return cols, len(rows)


def get_input_fn(csv_path, mode=ModeKeys.TRAIN, batch_size=32, cutoff=5, seed=None):
    """input_fn() -> iterator of (features: dict name -> [B] array, labels: [B] bool).
    TRAIN: shuffle with a 16*batch_size buffer, repeat forever; EVAL: one ordered pass, the last
    batch may be short (no drop_remainder) — the tf.data semantics of the reference pipeline."""
    def input_fn():
        cols, n = _read_csv(csv_path)
        label = cols.pop(LABEL_COL) >= cutoff

        def emit(idx):
            idx = np.asarray(idx)
            return {k: v[idx] for k, v in cols.items()}, label[idx]

        if mode != ModeKeys.TRAIN:
            for s in range(0, n, batch_size):
                yield emit(np.arange(s, min(n, s + batch_size)))
            return
        rng = np.random.default_rng(seed)
        cap = 16 * batch_size
        pending = np.empty(0, np.int64)
        while True:                                        # .repeat()
            # .shuffle(cap): a buffer of cap elements; every further element i draws a slot j uniformly, the slot's
            # occupant goes out, i takes its place; at the end of the pass the buffer goes out in random order.
            # The draws of a pass come from the generator in one call, the swap chain — element i's successor is the
            # next element that draws i's slot — is followed per SLOT in numpy: out[k] = the occupant of slot j[k]
            # before draw k = the element of the latest earlier draw with the same slot (or the slot's first occupant).
            m = max(n - cap, 0)
            j = rng.integers(cap, size=m) if m else np.empty(0, np.int64)
            order = np.argsort(j, kind="stable")           # draws grouped by slot, in time order inside a group
            js = j[order]
            prev = np.empty(m, np.int64)                   # for every draw: who sits in its slot
            first = np.ones(m, bool)
            first[1:] = js[1:] != js[:-1]
            prev[order] = np.where(first, js, np.concatenate([[0], order[:-1]]) + cap)   # slot's first occupant = element js
            out = prev                                     # (element ids: the first cap elements are 0..cap-1, draw k places element cap + k)
            buf = np.arange(min(n, cap), dtype=np.int64)
            if m:
                last = np.ones(m, bool)
                last[:-1] = js[1:] != js[:-1]
                buf[js[last]] = order[last] + cap          # what sits in a drawn slot after the pass
            rng.shuffle(buf)
            stream = np.concatenate([pending, out, buf])
            nb = len(stream) // batch_size
            for b in range(nb):
                yield emit(stream[b * batch_size:(b + 1) * batch_size])
            pending = stream[nb * batch_size:]
    return input_fn


def serving_input_fn():
    """Receiver for raw (untransformed) features: ids and numbers as int32, the rest as strings;
    genre flags default to 0."""
    spec = {"user_id": "int32 [None]", "item_id": "int32 [None]", "age": "int32 [None]", "gender": "string [None]",
            "occupation": "string [None]", "zipcode": "string [None]", "release_year": "int32 [None]"}
    spec.update({g: "int32 [None] default 0" for g in GENRE})
    return ServingInputReceiver(features=dict(spec), receiver_tensors=spec)
