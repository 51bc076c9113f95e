"""Canned classifiers with TF-1.12's defaults, as the reference's other three trainers
instantiate them: ``tf.estimator.LinearClassifier`` (trainers/linear.py:30-34), ``DNNClassifier``
(trainers/deep.py:32-38) and ``DNNLinearCombinedClassifier`` (trainers/linear_deep.py:32-39).
All three are the same engine with different parts switched on:

  Linear     wide part only                    Ftrl(lr = min(0.2, 1/sqrt(n_columns)))     loss SUM
  DNN        embeddings (mean) -> MLP          Adagrad(0.05)                               loss SUM
  W&D        both, logits added                Ftrl on the wide part, Adagrad on the deep  loss SUM
"""
import math

from .engine import DeepFM, OptimizerSpec
from .estimator import Estimator
from .feature_column import EmbeddingColumn, NumericColumn
from .model import run_batch


def _linear_lr(n_columns):
    return min(0.2, 1.0 / math.sqrt(max(n_columns, 1)))     # canned/linear.py _get_default_optimizer


def _embedding_size(columns, default=4):
    """Width of the fused table: the widest embedding column, in whole float4s (a narrower column — and a width that
    is no multiple of 4 — uses its first `dimension` columns: engine.DeepFM field_dims)."""
    dims = [c.dimension for c in columns if isinstance(c, EmbeddingColumn)]
    return (max(dims) + 3) // 4 * 4 if dims else default


def _name(c):
    return (c.categorical_column if isinstance(c, EmbeddingColumn) else c).name


def _subsets(plan, lin, dnn, lin_num, dnn_num):
    """engine.DeepFM's column-subset arguments for the plan's field order: per categorical field its embedding dimension
    (0: not in dnn_feature_columns) and whether linear_feature_columns has it; the same two flags per numeric column."""
    dim = {_name(c): c.dimension for c in dnn if isinstance(c, EmbeddingColumn)}
    wide = {_name(c) for c in lin}
    deep_n, wide_n = {c.name for c in dnn_num}, {c.name for c in lin_num}
    return dict(field_dims=[dim.get(c.name, 0) for c in plan.categorical],
                wide_fields=[c.name in wide for c in plan.categorical],
                deep_numeric=[c.name in deep_n for c in plan.numeric] if plan.numeric else None,
                wide_numeric=[c.name in wide_n for c in plan.numeric] if plan.numeric else None)


def _by_plan(plan, columns):
    """the embedding columns in the plan's field order"""
    by = {_name(c): c for c in columns}
    return [by[c.name] for c in plan.categorical]


def _split(columns):
    """(categorical / embedding columns, numeric columns) of a feature-column list"""
    cols = list(columns or [])
    return [c for c in cols if not isinstance(c, NumericColumn)], [c for c in cols if isinstance(c, NumericColumn)]


class LinearClassifier(Estimator):
    def __init__(self, feature_columns, model_dir=None, config=None, optimizer=None):
        cat, num = _split(feature_columns)
        opt = optimizer or OptimizerSpec("Ftrl", _linear_lr(len(cat) + len(num)))

        def model_fn(features, labels, mode, params):
            return run_batch(features, labels, mode, params, lambda plan, dev, shard=None: DeepFM(
                plan.vocab_sizes, n_numeric=len(plan.numeric), numeric="raw", use_linear=True, use_mf=False,
                use_dnn=False, optimizer=opt, reduction="sum", device=dev, shard=shard))
        super().__init__(model_fn, model_dir, config, {"categorical_columns": cat, "numeric_columns": num,
                                                       "tf_model": "linear"})


class DNNClassifier(Estimator):
    def __init__(self, hidden_units, feature_columns, model_dir=None, dropout=None, config=None, optimizer=None):
        cat, num = _split(feature_columns)
        E = _embedding_size(cat)
        opt = optimizer or OptimizerSpec("Adagrad", 0.05)
        hidden = list(hidden_units)

        def model_fn(features, labels, mode, params):
            return run_batch(features, labels, mode, params, lambda plan, dev, shard=None: DeepFM(
                plan.vocab_sizes, n_numeric=len(plan.numeric), numeric="raw", embedding_size=E, hidden_units=hidden,
                use_linear=False, use_mf=False, use_dnn=True, dropout=dropout or 0.0, optimizer=opt, reduction="sum",
                device=dev, shard=shard, field_dims=[getattr(c, "dimension", E) for c in _by_plan(plan, cat)]))
        super().__init__(model_fn, model_dir, config, {"categorical_columns": cat, "numeric_columns": num,
                                                       "tf_model": "dnn"})


class DNNLinearCombinedClassifier(Estimator):
    def __init__(self, model_dir=None, linear_feature_columns=None, dnn_feature_columns=None, dnn_hidden_units=None,
                 dnn_dropout=None, config=None, linear_optimizer=None, dnn_optimizer=None):
        lin_all = list(linear_feature_columns or [])
        dnn_all = list(dnn_feature_columns or [])
        if not lin_all and not dnn_all:
            raise ValueError("Either linear_feature_columns or dnn_feature_columns must be defined.")
        lin, lin_num = _split(lin_all)
        dnn, dnn_num = _split(dnn_all)
        for c in dnn:
            if not isinstance(c, EmbeddingColumn):
                raise ValueError("dnn_feature_columns takes embedding_column()s and numeric_column()s; wrap %r" % (_name(c),))
        # the two lists are independent (linear_deep.py:32-39 happens to pass the same columns twice): the fused table
        # holds the union of the categorical columns, each part reads its own subset (engine.DeepFM field_dims / ...)
        union = {_name(c): c for c in lin}
        union.update({_name(c): c for c in dnn})
        union_num = {c.name: c for c in lin_num + dnn_num}
        E = _embedding_size(dnn)
        l_opt = linear_optimizer or OptimizerSpec("Ftrl", _linear_lr(len(lin_all)))
        d_opt = dnn_optimizer or OptimizerSpec("Adagrad", 0.05)
        hidden = list(dnn_hidden_units or [])

        def model_fn(features, labels, mode, params):
            return run_batch(features, labels, mode, params, lambda plan, dev, shard=None: DeepFM(
                plan.vocab_sizes, n_numeric=len(plan.numeric), numeric="raw", embedding_size=E, hidden_units=hidden,
                use_linear=bool(lin_all), use_mf=False, use_dnn=bool(dnn_all), dropout=dnn_dropout or 0.0, optimizer=d_opt,
                linear_optimizer=l_opt if (lin_all and dnn_all) else None, reduction="sum", device=dev, shard=shard,
                **_subsets(plan, lin if lin_all else dnn, dnn, lin_num if lin_all else dnn_num, dnn_num))
                if dnn_all else DeepFM(
                plan.vocab_sizes, n_numeric=len(plan.numeric), numeric="raw", use_linear=True, use_mf=False, use_dnn=False,
                optimizer=l_opt, reduction="sum", device=dev, shard=shard))
        super().__init__(model_fn, model_dir, config, {"categorical_columns": list(union.values()),
                                                       "numeric_columns": list(union_num.values()),
                                                       "tf_model": "dnn_linear_combined"})
