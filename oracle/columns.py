"""Feature-column id transforms and the MovieLens-100k schema.  Oracle: tests only.

Restates ``trainers/ml_100k.py:3-39`` (schema, column constructors) with the TensorFlow semantics
of SURVEY A.1/A.2: plain Python / numpy integer work, one id per example per field.
"""
import numpy as np

from .fingerprint import fingerprint64

# trainers/ml_100k.py:3-9
COLUMNS = ("user_id,item_id,rating,timestamp,datetime,year,month,day,week,dayofweek,"
           "age,gender,occupation,zipcode,zipcode1,zipcode2,zipcode3,"
           "title,release,video_release,imdb,unknown,action,adventure,animation,children,"
           "comedy,crime,documentary,drama,fantasy,filmnoir,horror,musical,mystery,romance,"
           "scifi,thriller,war,western,release_date,release_year").split(",")
GENRE = COLUMNS[21:40]
INT_COLUMNS = {c for c in COLUMNS} - {"datetime", "gender", "occupation", "zipcode", "zipcode1",
                                      "zipcode2", "zipcode3", "title", "release", "video_release",
                                      "imdb", "release_date"}  # ml_100k.py:11-15 ([0] defaults)


def hash_bucket(values, num_buckets):
    """categorical_column_with_hash_bucket (ml_100k.py:19,20,29,30): Fingerprint64(as_string(v)) mod N."""
    out = np.empty(len(values), np.int32)
    for i, v in enumerate(values):
        if isinstance(v, (bytes, str)):
            s = v
        else:
            s = str(int(v))  # integer dtype -> decimal ASCII
        out[i] = fingerprint64(s) % num_buckets
    return out


def bucketize(values, boundaries):
    """bucketized_column (ml_100k.py:23-24,33-34): number of boundaries <= x."""
    b = np.asarray(boundaries, np.float32)
    return np.searchsorted(b, np.asarray(values, np.float32), side="right").astype(np.int32)


def vocabulary_list(values, vocab, num_oov_buckets=1):
    """categorical_column_with_vocabulary_list (ml_100k.py:25-28): index, OOV -> len(vocab) + hash % oov."""
    table = {v: i for i, v in enumerate(vocab)}
    out = np.empty(len(values), np.int32)
    for i, v in enumerate(values):
        v = v.decode() if isinstance(v, bytes) else v
        if v in table:
            out[i] = table[v]
        else:
            out[i] = len(vocab) + fingerprint64(v) % num_oov_buckets
    return out


def identity(values, num_buckets):
    """categorical_column_with_identity (ml_100k.py:35): the value itself, must be in range."""
    v = np.asarray(values, np.int64)
    if ((v < 0) | (v >= num_buckets)).any():
        raise ValueError("identity column value out of range [0, %d)" % num_buckets)
    return v.astype(np.int32)


def ml100k_fields():
    """The 26 categorical fields of get_feature_columns (ml_100k.py:18-37) as
    (column_name, source_key, kind, arg, vocab_size), in the reference's LIST order."""
    f = [
        ("user_id", "user_id", "hash_int", 1000, 1000),
        ("item_id", "item_id", "hash_int", 2000, 2000),
        ("age_bucketized", "age", "bucket", list(range(15, 66, 10)), 7),
        ("gender", "gender", "vocab", ["F", "M"], 3),
        ("occupation", "occupation", "hash_str", 50, 50),
        ("zipcode", "zipcode", "hash_str", 1000, 1000),
        ("release_year_bucketized", "release_year", "bucket", list(range(1930, 1991, 10)), 8),
    ]
    f += [(g, g, "identity", 2, 2) for g in GENRE]
    return f


def sorted_fields(fields):
    """input_layer / linear_model iterate columns sorted by name."""
    return sorted(fields, key=lambda t: t[0])


def transform(fields, features):
    """features: dict key -> sequence.  Returns ids [B, F] int32 in the order of ``fields``."""
    cols = []
    for name, key, kind, arg, _ in fields:
        v = features[key]
        if kind in ("hash_int", "hash_str"):
            cols.append(hash_bucket(v, arg))
        elif kind == "bucket":
            cols.append(bucketize(v, arg))
        elif kind == "vocab":
            cols.append(vocabulary_list(v, arg, 1))
        elif kind == "identity":
            cols.append(identity(v, arg))
        else:
            raise ValueError(kind)
    return np.stack(cols, 1)
