"""Glue between the Estimator-style surface and the engine: runs one batch in TRAIN / EVAL / PREDICT
mode and returns an EstimatorSpec.  Shared by ``trainers.deep_fm.model_fn`` and the canned
classifiers (``canned.py``).  Reference: the tail of model_fn, ``trainers/deep_fm.py:117-125``
(``head.create_estimator_spec``; the prediction / metric keys are those of SURVEY A.5)."""
import numpy as np
import torch

from . import _lib
from .estimator import EstimatorSpec, ModeKeys
from .feature_column import FieldPlan
from .metrics import metrics_from_counters


GRAPH_AUTO_MAX_BATCH = 1024      # from here on nothing in a step is launch-bound (bench.py: the replay is 1-6 % behind eager at B = 65536)


def binary_predictions(logits, kernels):
    """logits [B] -> the head's PREDICT dict (same keys as the TF head): mi_binary_predictions."""
    x = logits.reshape(-1).contiguous()
    B = x.numel()
    p = torch.empty(B, 1, dtype=torch.float32, device=x.device)
    prob = torch.empty(B, 2, dtype=torch.float32, device=x.device)
    cls = torch.empty(B, 1, dtype=torch.int64, device=x.device)
    kernels.mi_binary_predictions(x, None, B, p, prob, cls, None)
    return {"logits": x.reshape(-1, 1), "logistic": p, "probabilities": prob, "class_ids": cls, "classes": cls}


class _EvalCounters:
    def __init__(self, device):
        self.hist = torch.zeros(2 * 201, dtype=torch.int64, device=device)
        self.counts = torch.zeros(8, dtype=torch.int64, device=device)
        self.sums = torch.zeros(4, dtype=torch.float64, device=device)
        self.batch_losses = []

    def reset(self):
        self.hist.zero_(); self.counts.zero_(); self.sums.zero_()
        self.batch_losses = []


def run_batch(features, labels, mode, params, make_engine):
    """make_engine(plan, device) -> engine.DeepFM; called once, the result lives in params['_store']."""
    store = params.setdefault("_store", {})
    if "engine" not in store:
        plan = FieldPlan(params.get("categorical_columns", []), params.get("numeric_columns", []))
        device = params.get("device", "cuda")
        store["plan"] = plan
        shard = params.get("_shard")           # parallel.RowShard of a multi-GPU launch (trainers/_cli.py), else None
        eng = store["engine"] = make_engine(plan, device, shard) if shard is not None else make_engine(plan, device)
        gen = torch.Generator(device=eng.device)
        gen.manual_seed(int(params.get("seed", 0)) + (0 if shard is None else 7919 * shard.rank))
        eng.init_variables(gen)                # TF initialisers; a checkpoint restore overrides
        if shard is not None:
            from .parallel import broadcast_dense
            broadcast_dense(eng)               # the replicated MLP starts identical on every rank
        store["counters"] = None
    if mode == "_build":
        return None
    plan, eng = store["plan"], store["engine"]
    dev = eng.device
    ahead = params.get("_ahead")
    if ahead is not None and mode == ModeKeys.TRAIN and ahead.get("features") is features:
        # small batches (Estimator.train groups them): the id transforms of a whole group of batches were done in one
        # call and copied to the device once; this batch is rows lo..hi of the group
        g = ahead["group"]
        if "ids" not in g:
            ids_np, x_np = plan.transform(g["features"])
            g["ids"] = torch.from_numpy(ids_np).to(dev)
            g["x"] = torch.from_numpy(x_np).to(dev) if x_np is not None else None
            g["y"] = torch.from_numpy(np.ascontiguousarray(np.asarray(g["labels"]).reshape(-1)).astype(np.uint8)).to(dev)
        lo, hi = ahead["rows"]
        ids, y = g["ids"][lo:hi], g["y"][lo:hi]
        x = g["x"][lo:hi] if g["x"] is not None else None
    else:
        def stage(f, l):
            ids_np, x_np = plan.transform(f)
            y_ = None
            if l is not None:
                y_ = torch.from_numpy(np.ascontiguousarray(np.asarray(l).reshape(-1)).astype(np.uint8)).to(dev)
            return (torch.from_numpy(ids_np).to(dev), torch.from_numpy(x_np).to(dev) if x_np is not None else None, y_)
        staged = store.pop("staged", None)          # this batch, transformed and copied a step ago (Estimator._with_lookahead)
        if staged is not None and staged[0] is features and mode == ModeKeys.TRAIN:
            ids, x, y = staged[1]
        else:
            ids, x, y = stage(features, labels)

    # multi-GPU: a rank's loss is its SHARE of the global-batch mean (already divided by the global batch); times
    # world = the mean over its own examples — what is logged, and, with every rank evaluating the same batches,
    # the eval loss
    rescale = (lambda l: l * float(eng.shard.world)) if (eng.shard is not None and eng.reduction == "mean") else (lambda l: l)
    if mode == ModeKeys.TRAIN:
        # the whole step as ONE hipGraph launch — bit for bit the eager step (tests/test_hip_model.py) — where a step is
        # launch-bound: "auto" (the default) takes it for batches of <= GRAPH_AUTO_MAX_BATCH examples on a single GPU
        # (trainers.deep_fm at the reference's defaults, B = 32: 2-2.5x the eager rate), "on" / True always, "off" never
        hg = params.get("hip_graph", "auto")
        graph = hg in (True, "on") or (hg == "auto" and ids.shape[0] <= GRAPH_AUTO_MAX_BATCH)
        if getattr(eng, "summaries_next", False) and ids.shape[0] >= getattr(eng, "TOP_FUSED_MIN_BATCH", 1 << 62):
            graph = False         # (a captured large-batch step keeps the last hidden layer on the chip: this one is looked at)
        if graph and eng.device.type == "cuda" and hasattr(eng, "graph_ok") and eng.graph_ok():
            loss, logits = eng.graph_train_step(ids, y, x)
        else:
            # the next batch, if the train loop holds it (Estimator._with_lookahead: large batches, one GPU): transformed and
            # copied now — the GPU still runs the previous step — and announced to the engine
            la, nxt_ids = params.get("_lookahead"), None
            if la is not None and ahead is None and eng.shard is None:
                nxt = stage(la["features"], la["labels"])
                store["staged"] = (la["features"], nxt)
                if nxt[0].shape == ids.shape:
                    nxt_ids = nxt[0]
            loss, logits = eng.train_step(ids, y, x, next_ids=nxt_ids) if nxt_ids is not None else eng.train_step(ids, y, x)
        return EstimatorSpec(mode, predictions=None, loss=rescale(loss), train_op=eng.step)
    if mode == ModeKeys.EVAL:
        loss, logits = eng.loss(ids, y, x)
        loss = rescale(loss)
        if store["counters"] is None:
            store["counters"] = _EvalCounters(dev)
        ctr = store["counters"]
        if store.pop("metrics_reset", False):
            ctr.reset()
        eng.k.mi_eval_accumulate(logits, y, ids.shape[0], ctr.hist, ctr.counts, ctr.sums)
        ctr.batch_losses.append(loss.clone())

        def result():
            out = metrics_from_counters(ctr.hist.cpu().numpy(), ctr.counts.cpu().numpy(), ctr.sums.cpu().numpy())
            out["loss"] = float(torch.stack(ctr.batch_losses).mean())     # tf.metrics.mean over batch losses
            return out
        store["metrics_result"] = result
        return EstimatorSpec(mode, predictions=binary_predictions(logits, eng.k), loss=loss, eval_metric_ops=result)
    if mode == ModeKeys.PREDICT:
        logits = eng.predict_logits(ids, x)
        pr = binary_predictions(logits.clone(), eng.k)
        return EstimatorSpec(mode, predictions=pr, export_outputs={"predict": pr})
    raise ValueError("unknown mode %r" % (mode,))
