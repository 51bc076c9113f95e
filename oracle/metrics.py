"""Streaming eval metrics of the binary classification head.  Oracle: tests only.

Restates what ``head.create_estimator_spec`` reports in EVAL mode (reference
``trainers/deep_fm.py:118-125``) and what the unused helper ``get_binary_metric_ops``
(``trainers/model_utils.py:39-54``) spells out: accuracy, auc (ROC), auc_precision_recall,
average_loss — with ``tf.metrics.auc``'s 200-threshold trapezoidal approximation (SURVEY A.5,
recalled from TF 1.12 ``python/ops/metrics_impl.py``; PARITY UNPINNED).
"""
import numpy as np

NUM_THRESHOLDS = 200
_KEPS = 1e-7
_EPS = 1.0e-6


def auc_thresholds(num=NUM_THRESHOLDS):
    t = [(i + 1) * 1.0 / (num - 1) for i in range(num - 2)]
    return np.array([0.0 - _KEPS] + t + [1.0 + _KEPS], np.float32)  # compared against fp32 predictions


class BinaryMetrics:
    def __init__(self):
        n = NUM_THRESHOLDS
        self.tp = np.zeros(n, np.int64)
        self.fp = np.zeros(n, np.int64)
        self.tn = np.zeros(n, np.int64)
        self.fn = np.zeros(n, np.int64)
        self.n = 0
        self.n_pos = 0
        self.n_correct = 0
        self.tp5 = self.fp5 = self.fn5 = 0
        self.loss_sum = 0.0
        self.pred_sum = 0.0

    def update(self, logits, labels, per_example_loss=None):
        x = logits.astype(np.float32)
        y = labels.astype(bool)
        e = np.exp(-np.abs(x))
        p = np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(np.float32)
        th = auc_thresholds()
        pos = p[None, :] > th[:, None]                 # [T, B]
        self.tp += (pos & y[None]).sum(1)
        self.fp += (pos & ~y[None]).sum(1)
        self.fn += (~pos & y[None]).sum(1)
        self.tn += (~pos & ~y[None]).sum(1)
        cls = p > 0.5                                   # model_utils.py:12
        self.n += len(x)
        self.n_pos += int(y.sum())
        self.n_correct += int((cls == y).sum())
        self.tp5 += int((cls & y).sum())
        self.fp5 += int((cls & ~y).sum())
        self.fn5 += int((~cls & y).sum())
        if per_example_loss is None:
            xd = x.astype(np.float64)
            per_example_loss = np.maximum(xd, 0) - xd * y + np.log1p(np.exp(-np.abs(xd)))
        self.loss_sum += float(np.sum(per_example_loss, dtype=np.float64))
        self.pred_sum += float(p.astype(np.float64).sum())

    @staticmethod
    def _auc(tp, fp, tn, fn, curve):
        tp, fp, tn, fn = (a.astype(np.float64) for a in (tp, fp, tn, fn))
        rec = tp / (tp + fn + _EPS)
        if curve == "ROC":
            x, y = fp / (fp + tn + _EPS), rec
        else:
            x, y = rec, (tp + _EPS) / (tp + fp + _EPS)
        return float(np.sum((x[:-1] - x[1:]) * (y[:-1] + y[1:]) / 2.0))

    def result(self):
        n = max(self.n, 1)
        lm = self.n_pos / n
        return {
            "accuracy": self.n_correct / n,
            "accuracy_baseline": max(lm, 1 - lm),
            "auc": self._auc(self.tp, self.fp, self.tn, self.fn, "ROC"),
            "auc_precision_recall": self._auc(self.tp, self.fp, self.tn, self.fn, "PR"),
            "average_loss": self.loss_sum / n,
            "label/mean": lm,
            "prediction/mean": self.pred_sum / n,
            "precision": self.tp5 / max(self.tp5 + self.fp5, 1),
            "recall": self.tp5 / max(self.tp5 + self.fn5, 1),
        }
