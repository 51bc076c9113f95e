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
    dims = {c.dimension for c in columns if isinstance(c, EmbeddingColumn)}
    if len(dims) > 1:
        raise NotImplementedError("all embedding columns must share one dimension on the HIP path")
    return dims.pop() if dims else default


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
                device=dev, shard=shard))
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
        names = lambda cs: sorted((c.categorical_column if isinstance(c, EmbeddingColumn) else c).name for c in cs)
        if lin_all and dnn_all and (names(lin) != names(dnn) or names(lin_num) != names(dnn_num)):
            raise NotImplementedError("the HIP path shares one fused table and one numeric input: wide and deep parts "
                                      "must use the same columns (as trainers/linear_deep.py does)")
        E = _embedding_size(dnn)
        l_opt = linear_optimizer or OptimizerSpec("Ftrl", _linear_lr(len(lin_all)))
        d_opt = dnn_optimizer or OptimizerSpec("Adagrad", 0.05)
        hidden = list(dnn_hidden_units or [])

        def model_fn(features, labels, mode, params):
            return run_batch(features, labels, mode, params, lambda plan, dev, shard=None: DeepFM(
                plan.vocab_sizes, n_numeric=len(plan.numeric), numeric="raw", embedding_size=E, hidden_units=hidden,
                use_linear=bool(lin_all), use_mf=False, use_dnn=bool(dnn_all), dropout=dnn_dropout or 0.0, optimizer=d_opt,
                linear_optimizer=l_opt if (lin_all and dnn_all) else None, reduction="sum", device=dev, shard=shard)
                if dnn_all else DeepFM(
                plan.vocab_sizes, n_numeric=len(plan.numeric), numeric="raw", use_linear=True, use_mf=False, use_dnn=False,
                optimizer=l_opt, reduction="sum", device=dev, shard=shard))
        super().__init__(model_fn, model_dir, config, {"categorical_columns": lin or dnn, "numeric_columns": lin_num or dnn_num,
                                                       "tf_model": "dnn_linear_combined"})
