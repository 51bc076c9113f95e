"""Host-side finish of the streaming eval metrics (reference: the EVAL branch of
``head.create_estimator_spec``, trainers/deep_fm.py:118-125, and get_binary_metric_ops,
trainers/model_utils.py:39-54).  The device side (mi_eval_accumulate) only counts; the 200-threshold
trapezoidal AUC of tf.metrics.auc is a few hundred flops and is done here in fp64."""
import numpy as np

NUM_THRESHOLDS = 200
_EPS = 1.0e-6


def confusion_from_hist(hist):
    """hist [2, 201]: hist[y, k] = examples of label y whose sigmoid exceeds exactly k thresholds.
    prediction is positive at threshold j iff k > j."""
    h = np.asarray(hist, np.int64).reshape(2, NUM_THRESHOLDS + 1)
    # tail[j] = sum_{k > j} h[k]
    tail = np.cumsum(h[:, ::-1], 1)[:, ::-1]
    tot = h.sum(1)
    pos_gt = tail[:, 1:]                 # [2, 200]: count with k > j, j = 0..199
    tp, fp = pos_gt[1], pos_gt[0]
    fn, tn = tot[1] - tp, tot[0] - fp
    return tp, fp, tn, fn


def _auc(tp, fp, tn, fn, curve):
    tp, fp, tn, fn = (a.astype(np.float64) for a in (tp, fp, tn, fn))
    rec = tp / (tp + fn + _EPS)
    if curve == "ROC":
        x, y = fp / (fp + tn + _EPS), rec
    else:
        x, y = rec, (tp + _EPS) / (tp + fp + _EPS)
    return float(np.sum((x[:-1] - x[1:]) * (y[:-1] + y[1:]) / 2.0))


def metrics_from_counters(hist, counts, sums):
    tp, fp, tn, fn = confusion_from_hist(hist)
    n = max(int(counts[0]), 1)
    n_pos, n_correct = int(counts[1]), int(counts[3])
    tp5, fp5, fn5 = int(counts[4]), int(counts[5]), int(counts[6])
    lm = n_pos / n
    return {
        "accuracy": n_correct / n,
        "accuracy_baseline": max(lm, 1 - lm),
        "auc": _auc(tp, fp, tn, fn, "ROC"),
        "auc_precision_recall": _auc(tp, fp, tn, fn, "PR"),
        "average_loss": float(sums[0]) / n,
        "label/mean": lm,
        "prediction/mean": float(sums[1]) / n,
        "precision": tp5 / max(tp5 + fp5, 1),
        "recall": tp5 / max(tp5 + fn5, 1),
    }


def histogram_limits():
    """TensorFlow's default histogram bucket limits (core/lib/histogram/histogram.cc InitDefaultBucketsInner,
    what tf.summary.histogram of layer_summary — model_utils.py:6 — bins with): 1e-12 * 1.1^k below 1e20,
    DBL_MAX, their negatives and 0; ascending, 1,551 values."""
    pos, v = [], 1.0e-12
    while v < 1.0e20:
        pos.append(v)
        v *= 1.1
    pos.append(np.finfo(np.float64).max)
    return np.asarray([-x for x in reversed(pos)] + [0.0] + pos, np.float64)


def histogram_proto(limits, counts, sums, vmin, vmax):
    """The fields of a tensorflow.HistogramProto, empty buckets dropped (bucket b holds limit[b-1] <= x < limit[b])"""
    counts = np.asarray(counts, np.int64)
    nz = np.flatnonzero(counts)
    lim = np.append(limits, np.finfo(np.float64).max)
    return {"min": float(vmin), "max": float(vmax), "num": int(counts.sum()), "sum": float(sums[0]), "sum_squares": float(sums[1]),
            "bucket_limit": [float(lim[b]) for b in nz], "bucket": [int(counts[b]) for b in nz]}
