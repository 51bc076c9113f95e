"""Counterpart of the reference's ``trainers/model_utils.py`` (same names, same dict keys).

``layer_summary`` and ``get_optimizer`` are what ``deep_fm.model_fn`` imports (deep_fm.py:8); the
other helpers are dead code in the reference but spell out the prediction / loss / metric
contract, so they are kept as thin views over the engine's kernels."""
import torch

from mi355x_rec.engine import HipKernels, OptimizerSpec
from mi355x_rec.metrics import metrics_from_counters

_OPTIMIZERS = ("Adagrad", "Adam", "Ftrl", "RMSProp", "SGD")


def layer_summary(value):
    """zero fraction + histogram of an activation tensor (tf.summary.scalar("fraction_of_zero_values", zero_fraction(x))
    and tf.summary.histogram("activation", x) in the reference, model_utils.py:4-6; TensorFlow's default bucket limits)."""
    from mi355x_rec.metrics import histogram_limits, histogram_proto
    k = HipKernels()
    x = value.contiguous().view(-1)
    out = torch.empty(4, device=x.device)
    ws = torch.empty(k.query("mi_layer_stats_workspace_bytes", x.numel()) + 256, dtype=torch.uint8, device=x.device)
    k.mi_layer_stats(x, x.numel(), out, ws, ws.numel())
    z, mn, mx, mean = out.tolist()
    res = {"fraction_of_zero_values": z, "min": mn, "max": mx, "mean": mean}
    if getattr(k, "mi_layer_histogram", None) is not None and x.device.type == "cuda":
        lim = torch.from_numpy(histogram_limits()).to(x.device)
        counts = torch.zeros(lim.numel() + 1, dtype=torch.int64, device=x.device)
        sums = torch.zeros(2, dtype=torch.float64, device=x.device)
        k.mi_layer_histogram(x, x.numel(), lim, lim.numel(), counts, sums)
        res["activation"] = histogram_proto(lim.cpu().numpy(), counts.cpu().numpy(), sums.cpu().numpy(), mn, mx)
    return res


def get_binary_predictions(logits):
    x = logits.reshape(-1).contiguous()
    logistic = torch.empty_like(x)
    cls = torch.empty(x.numel(), dtype=torch.int64, device=x.device)
    HipKernels().mi_binary_predictions(x, None, x.numel(), logistic, None, cls, None)
    logistic = logistic.reshape(logits.shape)
    return {"logits": logits, "logistic": logistic, "probabilities": logistic,
            "class_id": cls.reshape(logits.shape).to(torch.int32)}


def get_binary_losses(labels, predictions):
    x = predictions["logits"].reshape(-1).contiguous()
    y = labels.reshape(-1).to(torch.uint8).contiguous()
    unreduced = torch.empty_like(x)
    HipKernels().mi_binary_predictions(x, y, x.numel(), None, None, None, unreduced)
    # (the two reductions of model_utils.py:28-29; the per-example values come from the library)
    return {"unreduced_loss": unreduced.reshape(-1, 1), "average_loss": unreduced.mean(), "loss": unreduced.sum()}


def get_binary_metric_ops(labels, predictions, losses):
    k = HipKernels()
    x = predictions["logits"].reshape(-1).contiguous()
    y = labels.reshape(-1).to(torch.uint8).contiguous()
    hist = torch.zeros(2 * 201, dtype=torch.int64, device=x.device)
    counts = torch.zeros(8, dtype=torch.int64, device=x.device)
    sums = torch.zeros(4, dtype=torch.float64, device=x.device)
    k.mi_eval_accumulate(x, y, x.numel(), hist, counts, sums)
    m = metrics_from_counters(hist.cpu().numpy(), counts.cpu().numpy(), sums.cpu().numpy())
    return {key: m[key] for key in ("accuracy", "auc", "auc_precision_recall", "average_loss")}


def get_optimizer(optimizer_name="Adam", learning_rate=0.001):
    if optimizer_name not in _OPTIMIZERS:
        raise KeyError(optimizer_name)
    return OptimizerSpec(optimizer_name, learning_rate)


class TrainOp:
    """What ``optimizer.minimize(loss, global_step)`` is here: there is no graph, so the op is a callable bound to
    an optimizer spec; ``op(engine, ids, labels, x_num=None)`` runs one train step — forward, head, backward and
    that optimizer's apply on every variable, global_step += 1 — on an engine built with that spec (model_fn's
    TRAIN branch does exactly this through engine.DeepFM.train_step)."""

    def __init__(self, loss, optimizer):
        self.loss, self.optimizer = loss, optimizer

    def __call__(self, engine, ids, labels, x_num=None):
        o, e = self.optimizer, engine.opt
        if (o.name, o.lr) != (e.name, e.lr):
            raise ValueError("train op for %s(%g) applied to an engine built with %s(%g)" % (o.name, o.lr, e.name, e.lr))
        return engine.train_step(ids, labels, x_num)


def get_train_op(loss, optimizer):
    """model_utils.py:69-72: ``optimizer.minimize(loss, global_step=tf.train.get_global_step())``"""
    if not isinstance(optimizer, OptimizerSpec):
        raise TypeError("optimizer must come from get_optimizer()")
    return TrainOp(loss, optimizer)
