"""Feature-column descriptors: the subset of ``tf.feature_column`` the reference uses
(``trainers/ml_100k.py:19-38``, ``trainers/deep_fm.py:39,52-54,62``), with TF-1.12's id semantics
(SURVEY A.1-A.3).  A column only *describes* a field; ``FieldPlan`` lowers a list of columns to
what the HIP path consumes: per-field row offsets in one fused table and an int32 id matrix
[B, F] in sorted-by-column-name order.  The hashing / bucketizing itself runs in the C ABI
(``mi_hash_bucket_*``, ``mi_bucketize_f32``)."""
import ctypes as C

import numpy as np

from . import _lib


class NumericColumn:
    def __init__(self, key, shape=(1,), default_value=None, dtype=np.float32):
        self.key, self.shape, self.default_value, self.dtype = key, tuple(shape), default_value, dtype
        self.name = key

    def values(self, features):
        return np.asarray(features[self.key], np.float32).reshape(-1)


class _Categorical:
    num_buckets = 0

    def __init__(self, key, name=None):
        self.key = key
        self.name = name or key


class HashBucketColumn(_Categorical):
    """categorical_column_with_hash_bucket: Fingerprint64(as_string(v)) mod hash_bucket_size."""

    def __init__(self, key, hash_bucket_size, dtype=str):
        super().__init__(key)
        if hash_bucket_size is None or hash_bucket_size < 1:
            raise ValueError("hash_bucket_size must be at least 1. hash_bucket_size: {}, key: {}".format(
                hash_bucket_size, key))
        self.num_buckets = int(hash_bucket_size)
        self.dtype = dtype

    def transform(self, features):
        lib = _lib.load()
        v = features[self.key]
        n = len(v)
        out = np.empty(n, np.int32)
        arr = np.asarray(v)
        if arr.dtype.kind in "iu":
            a = np.ascontiguousarray(arr, np.int64)
            _lib.check(lib.mi_hash_bucket_i64(a.ctypes.data, n, self.num_buckets, out.ctypes.data), "mi_hash_bucket_i64")
        else:
            blob = offs = None
            # the whole column at once: numpy encodes ASCII strings into a fixed-width byte matrix at C speed; the
            # padding is cut out with one boolean index.  Anything that path would change takes the per-element path
            # below (the bytes mean the same in either case: utf-8 of str(value)): non-ASCII text (the encode raises),
            # values that are neither str nor bytes (len() raises), trailing NULs, which a fixed-width array drops
            # (the total length then differs from Python's).
            if n and arr.dtype.kind in "OUS":
                try:
                    sa = arr if arr.dtype.kind == "S" else arr.astype("S")
                    w = sa.dtype.itemsize
                    lens = np.char.str_len(sa).astype(np.int64)
                    if w and int(lens.sum()) == (sum(map(len, v)) if not (isinstance(v, np.ndarray) and arr.dtype.kind in "US") else int(lens.sum())):
                        mat = np.ascontiguousarray(sa).view(np.uint8).reshape(n, w)
                        blob = mat[np.arange(w)[None, :] < lens[:, None]].tobytes()
                        offs = np.zeros(n + 1, np.int64)
                        np.cumsum(lens, out=offs[1:])
                except (UnicodeEncodeError, ValueError, TypeError):
                    blob = offs = None
            if blob is None:
                enc = [x if isinstance(x, bytes) else str(x).encode("utf-8") for x in v]
                offs = np.zeros(n + 1, np.int64)
                np.cumsum([len(e) for e in enc], out=offs[1:])
                blob = b"".join(enc)
            buf = C.create_string_buffer(blob, max(len(blob), 1))
            _lib.check(lib.mi_hash_bucket_bytes(C.addressof(buf), offs.ctypes.data, n, self.num_buckets,
                                                out.ctypes.data), "mi_hash_bucket_bytes")
        return out


class VocabularyListColumn(_Categorical):
    """categorical_column_with_vocabulary_list: index in the list; unknown -> len + hash % oov
    (or default_value when num_oov_buckets == 0)."""

    def __init__(self, key, vocabulary_list, dtype=None, default_value=-1, num_oov_buckets=0):
        super().__init__(key)
        self.vocab = [v.decode() if isinstance(v, bytes) else v for v in vocabulary_list]
        if not self.vocab:
            raise ValueError("vocabulary_list {} must be non-empty, column_name: {}".format(vocabulary_list, key))
        if num_oov_buckets and default_value != -1:
            raise ValueError("Can't specify both num_oov_buckets and default_value in {}.".format(key))
        self.default_value, self.num_oov = int(default_value), int(num_oov_buckets)
        self.num_buckets = len(self.vocab) + self.num_oov
        self._index = {v: i for i, v in enumerate(self.vocab)}

    def transform(self, features):
        lib = _lib.load()
        vals = features[self.key]
        arr = np.asarray(vals)
        if len(arr) > 64 and arr.dtype.kind in "OUS":
            # a batch holds few distinct values of a vocabulary column: look each one up once
            try:
                uniq, inv = np.unique(arr.astype("U"), return_inverse=True)
                return self._lookup(lib, uniq.tolist())[inv].astype(np.int32)
            except (UnicodeDecodeError, ValueError, TypeError):
                pass
        return self._lookup(lib, vals)

    def _lookup(self, lib, values):
        out = np.empty(len(values), np.int32)
        for i, v in enumerate(values):
            v = v.decode() if isinstance(v, bytes) else (v if isinstance(v, str) else v.item() if hasattr(v, "item") else v)
            j = self._index.get(v)
            if j is None:
                if self.num_oov:
                    b = str(v).encode("utf-8")
                    j = len(self.vocab) + lib.mi_fingerprint64(b, len(b)) % self.num_oov
                else:
                    j = self.default_value
            out[i] = j
        return out


class IdentityColumn(_Categorical):
    """categorical_column_with_identity: the integer itself, checked against [0, num_buckets)."""

    def __init__(self, key, num_buckets, default_value=None):
        super().__init__(key)
        if num_buckets < 1:
            raise ValueError("num_buckets {} < 1, column_name {}".format(num_buckets, key))
        self.num_buckets, self.default_value = int(num_buckets), default_value

    def transform(self, features):
        v = np.asarray(features[self.key], np.int64).reshape(-1)
        bad = (v < 0) | (v >= self.num_buckets)
        if bad.any():
            if self.default_value is None:
                raise ValueError("column %s: value %d outside [0, %d)" % (self.key, int(v[bad][0]), self.num_buckets))
            v = np.where(bad, self.default_value, v)
        return v.astype(np.int32)


class BucketizedColumn(_Categorical):
    """bucketized_column: id = number of boundaries <= x (len(boundaries)+1 buckets)."""

    def __init__(self, source_column, boundaries):
        if not isinstance(source_column, NumericColumn):
            raise ValueError("source_column must be a column generated with numeric_column().")
        b = [float(x) for x in boundaries]
        if not b or any(b[i] >= b[i + 1] for i in range(len(b) - 1)):
            raise ValueError("boundaries must be a sorted list.")
        super().__init__(source_column.key, source_column.name + "_bucketized")
        self.source, self.boundaries = source_column, np.asarray(b, np.float32)
        self.num_buckets = len(b) + 1

    def transform(self, features):
        lib = _lib.load()
        x = np.ascontiguousarray(self.source.values(features))
        out = np.empty(len(x), np.int32)
        _lib.check(lib.mi_bucketize_f32(x.ctypes.data, len(x), self.boundaries.ctypes.data, len(self.boundaries),
                                        out.ctypes.data), "mi_bucketize_f32")
        return out


class EmbeddingColumn:
    """embedding_column: combiner 'mean', truncated-normal init."""

    def __init__(self, categorical_column, dimension, combiner="mean"):
        if dimension is None or dimension < 1:
            raise ValueError("Invalid dimension {}.".format(dimension))
        self.categorical_column, self.dimension, self.combiner = categorical_column, int(dimension), combiner
        self.name = categorical_column.name + "_embedding"


# tf.feature_column-style constructors -----------------------------------------------------
def numeric_column(key, shape=(1,), default_value=None, dtype=np.float32):
    return NumericColumn(key, shape, default_value, dtype)


def categorical_column_with_hash_bucket(key, hash_bucket_size, dtype=str):
    return HashBucketColumn(key, hash_bucket_size, dtype)


def categorical_column_with_vocabulary_list(key, vocabulary_list, dtype=None, default_value=-1, num_oov_buckets=0):
    return VocabularyListColumn(key, vocabulary_list, dtype, default_value, num_oov_buckets)


def categorical_column_with_identity(key, num_buckets, default_value=None):
    return IdentityColumn(key, num_buckets, default_value)


def bucketized_column(source_column, boundaries):
    return BucketizedColumn(source_column, boundaries)


def embedding_column(categorical_column, dimension, combiner="mean"):
    return EmbeddingColumn(categorical_column, dimension, combiner)


class FieldPlan:
    """Lowering of (categorical_columns, numeric_columns) to the fused-table layout.

    Field order: TF's input_layer iterates the EMBEDDING columns sorted by their name, which is
    ``<categorical name>_embedding`` — that order fixes the row blocks of the first MLP
    kernel, so it is the order of the fused table.  It differs from sorting by the categorical name
    only when one name is a prefix of another followed by a character below '_' ('war' < 'war2' but
    'war2_embedding' < 'war_embedding'); linear_model sorts by the categorical name, which for such a
    pair changes nothing but the fp32 association of the wide sum (the kernels add the fields of an
    example as a lane tree, not sequentially, in either order)."""

    def __init__(self, categorical_columns, numeric_columns=()):
        cats = [c.categorical_column if isinstance(c, EmbeddingColumn) else c for c in categorical_columns]
        for c in cats:
            if not isinstance(c, _Categorical):
                raise ValueError("not a categorical column: %r" % (c,))
        for c in numeric_columns:
            if not isinstance(c, NumericColumn):
                raise ValueError("not a numeric column: %r" % (c,))
        self.categorical = sorted(cats, key=lambda c: c.name + "_embedding")
        self.numeric = sorted(numeric_columns, key=lambda c: c.name)
        names = [c.name for c in self.categorical]
        if len(set(names)) != len(names):
            raise ValueError("duplicate column names: %s" % names)
        self.vocab_sizes = [c.num_buckets for c in self.categorical]

    def transform(self, features):
        """features: dict key -> sequence of B raw values.  Returns (ids int32 [B,F], x float32 [B,n_d] or None)."""
        cols = []
        for c in self.categorical:
            v = np.asarray(c.transform(features))
            # The kernels index table[field_off[f] + id] unchecked.  TF drops ids < 0 (an out-of-vocabulary
            # value of a vocabulary column without OOV buckets: zero embedding, zero linear term) and fails
            # on ids >= num_buckets; neither may reach the device here.
            if v.size and (v.min() < 0 or v.max() >= c.num_buckets):
                bad = v[(v < 0) | (v >= c.num_buckets)][0]
                raise ValueError("column %r produced id %d outside [0, %d): give the column num_oov_buckets (or a "
                                 "default_value inside the vocabulary); the HIP path has no dropped-id row"
                                 % (c.name, int(bad), c.num_buckets))
            cols.append(v)
        ids = (np.stack(cols, 1) if cols else np.zeros((self._batch_size(features), 0))).astype(np.int32)
        x = None
        if self.numeric:
            x = np.stack([c.values(features) for c in self.numeric], 1).astype(np.float32)
        return np.ascontiguousarray(ids), x

    def _batch_size(self, features):
        for c in self.numeric:
            return len(c.values(features))
        return len(next(iter(features.values())))
