"""A minimal Estimator loop with the knobs the reference configures (``trainers/conf_utils.py:6-34``)
and the call shapes its trainers use (``tf.estimator.Estimator(model_fn, model_dir, config, params)``
+ ``train_and_evaluate``, ``trainers/deep_fm.py:153-178``).  Not a GPU target:
plain Python around the engine.

Differences from TensorFlow that matter to a user switching over:
  * model_fn is called once per batch and EXECUTES the step (there is no graph); variables live
    in ``params["_store"]``, which the Estimator owns and checkpoints;
  * checkpoints are ``model.ckpt-<step>.pt`` (torch.save of engine.state_dict()), newest
    ``keep_checkpoint_max`` kept, listed in ``checkpoint.json``; a TF-checkpoint importer would map
    the variable names of SURVEY A.8;
  * multi-GPU runs launch one process per GPU (torch.distributed.run; trainers/_cli.py reads RANK /
    WORLD_SIZE) instead of TF_CONFIG parameter servers (distributed.md:58-82): synchronous steps, row-
    sharded tables, per-rank checkpoint files.
"""
import collections
import glob
import json
import os
import time

import numpy as np
import torch


class ModeKeys:
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"


EstimatorSpec = collections.namedtuple("EstimatorSpec", "mode predictions loss train_op eval_metric_ops export_outputs")
EstimatorSpec.__new__.__defaults__ = (None,) * 5
TrainSpec = collections.namedtuple("TrainSpec", "input_fn max_steps")
EvalSpec = collections.namedtuple("EvalSpec", "input_fn steps exporters start_delay_secs throttle_secs")
ServingInputReceiver = collections.namedtuple("ServingInputReceiver", "features receiver_tensors")


class RunConfig:
    def __init__(self, model_dir=None, save_checkpoints_secs=600, keep_checkpoint_max=5, save_summary_steps=100,
                 log_step_count_steps=100, device="cuda", clock_sync_steps=10):
        self.model_dir = model_dir
        self.save_checkpoints_secs = save_checkpoints_secs
        self.keep_checkpoint_max = keep_checkpoint_max
        self.save_summary_steps = save_summary_steps
        self.log_step_count_steps = log_step_count_steps
        self.device = device
        # multi-GPU: wall-clock decisions (checkpoint now?) are taken by rank 0 and broadcast, every this many steps
        self.clock_sync_steps = clock_sync_steps


def _agree(shard, value):
    """Rank 0's integer `value` on every rank (one tiny broadcast).  Every wall-clock decision of a multi-GPU run goes
    through this: the ranks' clocks differ by far more than a step, and a rank that starts an evaluation (a collective
    in the row-sharded engine) while the others start the next train step mismatches the collectives of the two."""
    if shard is None or shard.world == 1:
        return int(value)
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(shard.group) == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, 0, group=shard.group)
    return int(t.item())


def _all_equal(shard, value):
    """Does every rank hold the same integer?  (min == max over the ranks)"""
    if shard is None or shard.world == 1:
        return True
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(shard.group) == "nccl" else "cpu"
    t = torch.tensor([int(value), -int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=shard.group)
    return int(t[0].item()) == -int(t[1].item())


class LatestExporter:
    """Writes the newest model state + serving signature under <model_dir>/export/<name>/<ts>/,
    keeping `exports_to_keep` (conf_utils.py:20-24)."""

    def __init__(self, name, serving_input_receiver_fn, exports_to_keep=5):
        self.name, self.fn, self.keep = name, serving_input_receiver_fn, exports_to_keep

    def export(self, estimator, export_dir):
        """Single GPU: variables.pt.  N GPUs (every rank calls this): the chief picks the directory and writes the
        signature, whose "sharding" entry says how to put the model together again — row r of the stacked tables lives
        in variables.rank<r % world>.pt at index r // world; the dense variables are replicated in every file."""
        shard = estimator._shard
        ts = 0
        if estimator.is_chief:
            ts = int(time.time())
            while os.path.exists(os.path.join(export_dir, self.name, str(ts))):
                ts += 1
            os.makedirs(os.path.join(export_dir, self.name, str(ts)))
        ts = _agree(shard, ts)                       # (also orders the chief's makedirs before the other ranks' writes)
        out = os.path.join(export_dir, self.name, str(ts))
        recv = self.fn()
        torch.save(estimator._engine().state_dict(), os.path.join(out, "variables%s.pt" % estimator._rank_tag))
        if estimator.is_chief:
            sig = {"receiver_tensors": {k: str(v) for k, v in recv.receiver_tensors.items()},
                   "outputs": ["logits", "logistic", "probabilities", "class_ids", "classes"],
                   "global_step": estimator.global_step}
            if shard is not None:
                sig["sharding"] = {"world": shard.world, "files": ["variables.rank%d.pt" % r for r in range(shard.world)],
                                   "rule": "row r of table / lin_w / slots: file r % world, index r // world; dense replicated"}
            with open(os.path.join(out, "signature.json"), "w") as f:
                json.dump(sig, f, indent=1)
            old = sorted(glob.glob(os.path.join(export_dir, self.name, "*")))
            for d in old[:-self.keep]:
                for fn in glob.glob(os.path.join(d, "*")):
                    os.remove(fn)
                os.rmdir(d)
        return out


class Estimator:
    def __init__(self, model_fn, model_dir=None, config=None, params=None, warm_start_from=None):
        """warm_start_from: .npz of TensorFlow-named variables (mi355x_rec/tf_names.py; the dump of a
        TF-1.12 checkpoint of the reference), applied when model_dir holds no checkpoint — the role of
        tf.estimator.Estimator(warm_start_from=...)."""
        self.model_fn = model_fn
        self.warm_start_from = warm_start_from
        self.config = config or RunConfig()
        self.model_dir = model_dir or self.config.model_dir or "checkpoints/model"
        self.params = dict(params or {})
        self.params.setdefault("_store", {})
        self.params.setdefault("device", self.config.device)
        self._restored = False

    # -- variable store -----------------------------------------------------------------
    def _engine(self):
        return self.params["_store"].get("engine")

    @property
    def _shard(self):
        return self.params.get("_shard")

    @property
    def _rank_tag(self):
        """multi-GPU: every rank keeps its own shard of the tables (+ the replicated dense variables)"""
        return "" if self._shard is None else ".rank%d" % self._shard.rank

    @property
    def is_chief(self):
        return self._shard is None or self._shard.rank == 0

    @property
    def global_step(self):
        e = self._engine()
        return e.step if e is not None else 0

    def latest_checkpoint(self):
        idx = os.path.join(self.model_dir, "checkpoint%s.json" % self._rank_tag)
        if not os.path.exists(idx):
            return None
        with open(idx) as f:
            files = json.load(f)["all_model_checkpoint_paths"]
        return os.path.join(self.model_dir, files[-1]) if files else None

    def _maybe_restore(self):
        if self._restored or self._engine() is None:
            return
        self._restored = True
        ck = self.latest_checkpoint()
        # (multi-GPU: every rank restores its own shard; a rank without its file, or with an older one, would
        # otherwise resume at a different step than the others)
        if not _all_equal(self._shard, 1 if ck else 0):
            raise RuntimeError("restore: some ranks found a checkpoint in %s and some did not (this rank: %s)" % (self.model_dir, ck))
        if ck:
            self._engine().load_state_dict(torch.load(ck, weights_only=True))
            if not _all_equal(self._shard, self.global_step):
                raise RuntimeError("restore: the ranks' newest checkpoints are of different steps (this rank: %s, step %d)" %
                                   (ck, self.global_step))
            if self.is_chief:
                print("INFO: restored %s (global_step %d)" % (ck, self.global_step))
        elif self.warm_start_from:
            import numpy as np
            from . import tf_bundle, tf_names
            plan = self.params["_store"]["plan"]
            cols, nums = [c.name for c in plan.categorical], [c.name for c in plan.numeric]
            model = self.params.get("tf_model", "deep_fm")
            if tf_bundle.is_bundle(self.warm_start_from):
                # a TensorFlow checkpoint itself (model_dir or prefix): only the model's variables are read — not the
                # optimizer slots, global_step, beta powers that the bundle also holds
                nm = tf_names._names_for(self._engine(), model, cols, nums, sharded_ok=True)
                flat = []

                def walk(v):
                    if isinstance(v, str):
                        flat.append(v)
                    elif isinstance(v, (list, tuple)):
                        for q in v:
                            walk(q)
                walk(list(nm.values()))
                have = tf_bundle.list_variables(self.warm_start_from)
                arrays = tf_bundle.read_bundle(self.warm_start_from, names=[n for n in flat if n in have])
            else:
                with np.load(self.warm_start_from, allow_pickle=False) as z:
                    arrays = dict(z)
            names = tf_names.import_variables(self._engine(), arrays, cols, model=model, numeric_names=nums)
            print("INFO: warm-started %d variables from %s" % (len(names), self.warm_start_from))

    def save_checkpoint(self):
        os.makedirs(self.model_dir, exist_ok=True)
        name = "model.ckpt-%d%s.pt" % (self.global_step, self._rank_tag)
        torch.save(self._engine().state_dict(), os.path.join(self.model_dir, name))
        idx = os.path.join(self.model_dir, "checkpoint%s.json" % self._rank_tag)
        files = []
        if os.path.exists(idx):
            with open(idx) as f:
                files = json.load(f)["all_model_checkpoint_paths"]
        files = [f for f in files if f != name] + [name]
        for old in files[:-self.config.keep_checkpoint_max]:
            p = os.path.join(self.model_dir, old)
            if os.path.exists(p):
                os.remove(p)
        files = files[-self.config.keep_checkpoint_max:]
        with open(idx, "w") as f:
            json.dump({"model_checkpoint_path": name, "all_model_checkpoint_paths": files}, f)
        return os.path.join(self.model_dir, name)

    def _write_summaries(self, loss):
        """Every save_summary_steps: loss + the layer_summary statistics (model_utils.py:4-6; TensorBoard
        events in the reference) appended as one JSON line to <model_dir>/summaries.jsonl."""
        eng = self._engine()
        if eng is None or not hasattr(eng, "layer_summaries") or not self.is_chief:
            return                                   # (multi-GPU: the chief's batch share; one writer per file)
        os.makedirs(self.model_dir, exist_ok=True)
        rec = {"global_step": self.global_step, "loss": loss, "layers": eng.layer_summaries()}
        with open(os.path.join(self.model_dir, "summaries.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")

    # -- modes ----------------------------------------------------------------------------
    def _first_call(self, features, labels, mode):
        """Build the variables (first model_fn call) and restore the latest checkpoint before any step."""
        if self._engine() is None:
            self.model_fn(features, labels, "_build", self.params)
        self._maybe_restore()

    def train(self, input_fn, steps=None, max_steps=None, on_checkpoint=None):
        t_ckpt = t_log = time.time()
        n_log = 0
        done = 0
        loss = None
        for features, labels in self._with_lookahead(self._grouped(input_fn())):
            self._first_call(features, labels, ModeKeys.TRAIN)
            if max_steps is not None and self.global_step >= max_steps:
                break
            if steps is not None and done >= steps:
                break
            eng = self._engine()
            if eng is not None and hasattr(eng, "summaries_next"):
                # (the step whose layer_summary statistics are recorded keeps every layer's output in memory: engine._top_fusable)
                n_sum = self.config.save_summary_steps
                eng.summaries_next = bool(n_sum and (self.global_step + 1) % n_sum == 0)
            spec = self.model_fn(features, labels, ModeKeys.TRAIN, self.params)
            loss = spec.loss
            done += 1
            n_log += 1
            if self.global_step % self.config.log_step_count_steps == 0 and self.is_chief:
                now = time.time()
                print("INFO: loss = %.6f, step = %d (%.1f global_step/sec)" %
                      (float(loss), self.global_step, n_log / max(now - t_log, 1e-9)))
                t_log, n_log = now, 0
            if self.config.save_summary_steps and self.global_step % self.config.save_summary_steps == 0:
                self._write_summaries(float(loss))
            if self.config.save_checkpoints_secs and self._checkpoint_due(t_ckpt):
                self.save_checkpoint()
                t_ckpt = time.time()
                if on_checkpoint:
                    on_checkpoint()
        if self._engine() is not None and done:
            self.save_checkpoint()
            if on_checkpoint:
                on_checkpoint()
        return self

    LOOKAHEAD_MIN_BATCH = 4096    # from this batch size on the train loop holds the NEXT batch too (see _with_lookahead)

    def _with_lookahead(self, it):
        """The training batches of `it`, unchanged — but from LOOKAHEAD_MIN_BATCH examples on (one GPU) the loop reads one batch
        ahead and leaves the next batch in params["_lookahead"]: run_batch then transforms its ids and copies them to the device
        while the GPU still runs the previous step, and announces them to the engine (train_step(next_ids=...)), whose sort of
        the next batch then runs beside this step's catch-up instead of at the head of the next step — what a prefetching
        input pipeline gives the reference (tf.data, ml_100k.py:42-61).  Results are those of the plain loop, bit for bit
        (tests/test_trainers.py).  Row-sharded training keeps the plain loop: every rank would have to announce alike."""
        it = iter(it)
        cur = next(it, None)
        if cur is None:
            return
        try:
            B = len(cur[1])
        except TypeError:
            B = 0
        if B < self.LOOKAHEAD_MIN_BATCH or self._shard is not None or not isinstance(cur[0], dict):
            self.params.pop("_lookahead", None)
            yield cur
            yield from it
            return
        while cur is not None:
            nxt = next(it, None)
            self.params["_lookahead"] = None if nxt is None else {"features": nxt[0], "labels": nxt[1]}
            yield cur
            cur = nxt
        self.params.pop("_lookahead", None)

    GROUP_ROWS = 2048         # small batches: this many examples' id transforms in one call (see _grouped)

    def _grouped(self, it):
        """The training batches of `it`, unchanged — but batches of a few dozen examples (the reference's default is 32)
        are drawn GROUP_ROWS examples at a time, and the group's concatenated columns ride along in params["_ahead"]: the
        engine's run_batch then transforms the ids of the whole group in one call and copies them to the device once (the
        per-call overhead of 26 column transforms and three host-to-device copies is what a 32-example step costs on the
        host; the transforms are stateless, so the order of results is that of the batches).  A model_fn that does not look
        at params["_ahead"] sees exactly what it always saw."""
        import itertools
        it = iter(it)
        first = next(it, None)
        if first is None:
            return
        try:
            B = len(first[1])
        except TypeError:
            B = 0
        k = self.GROUP_ROWS // B if 0 < B <= self.GROUP_ROWS // 4 else 1
        it = itertools.chain([first], it)
        if k <= 1 or not isinstance(first[0], dict):
            self.params.pop("_ahead", None)
            yield from it
            return
        while True:
            grp = list(itertools.islice(it, k))
            if not grp:
                break
            feats = {key: np.concatenate([np.asarray(f[key]) for f, _ in grp]) for key in grp[0][0]}
            group = {"features": feats, "labels": np.concatenate([np.asarray(l).reshape(-1) for _, l in grp])}
            lo = 0
            for f, l in grp:
                n = len(l)
                self.params["_ahead"] = {"features": f, "group": group, "rows": (lo, lo + n)}
                lo += n
                yield f, l
        self.params.pop("_ahead", None)

    def _checkpoint_due(self, t_last):
        """save_checkpoints_secs have passed.  One process: its own clock, every step.  N processes: rank 0's clock,
        looked at every clock_sync_steps-th global step (the same steps on every rank: synchronous training) and
        broadcast — the checkpoint, and the evaluation that follows it, then start at the same step everywhere."""
        due = time.time() - t_last >= self.config.save_checkpoints_secs
        if self._shard is None or self._shard.world == 1:
            return due
        if self.global_step % max(1, int(self.config.clock_sync_steps)):
            return False
        return bool(_agree(self._shard, due))

    def evaluate(self, input_fn, steps=None):
        store = self.params["_store"]
        n = 0
        for features, labels in input_fn():
            self._first_call(features, labels, ModeKeys.EVAL)
            if n == 0:
                store["metrics_reset"] = True
            self.model_fn(features, labels, ModeKeys.EVAL, self.params)
            n += 1
            if steps is not None and n >= steps:
                break
        out = store["metrics_result"]() if n else {}
        out["global_step"] = self.global_step
        if self.is_chief:
            print("INFO: Saving dict for global step %d: %s" % (self.global_step, ", ".join(
            "%s = %.6g" % (k, v) for k, v in sorted(out.items()))))
        return out

    def predict(self, input_fn):
        for batch in input_fn():
            features = batch[0] if isinstance(batch, tuple) else batch
            self._first_call(features, None, ModeKeys.PREDICT)
            spec = self.model_fn(features, None, ModeKeys.PREDICT, self.params)
            pr = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in spec.predictions.items()}
            for i in range(len(pr["logits"])):
                yield {k: v[i] for k, v in pr.items()}


def train_and_evaluate(estimator, train_spec, eval_spec):
    """Local-mode tf.estimator.train_and_evaluate: train to max_steps; after every checkpoint
    (every save_checkpoints_secs and at the end) evaluate on the whole eval input and export."""
    t_start = time.time()
    state = {"last_eval": None, "evaluated_step": None}

    def after_checkpoint(final=False):
        # EvalSpec.start_delay_secs / throttle_secs (conf_utils.py:27-34): no evaluation before
        # start_delay_secs of training, none sooner than throttle_secs after the previous one started;
        # the checkpoint written when training ends is always evaluated (as TF's local loop does).
        # Multi-GPU: every rank gets here at the same global step (the checkpoint decision is collective), and rank 0's
        # clock decides for all of them — evaluate() is a collective in the row-sharded engine.
        now = time.time()
        go = True
        if not final:
            if now - t_start < (eval_spec.start_delay_secs or 0):
                go = False
            elif state["last_eval"] is not None and now - state["last_eval"] < (eval_spec.throttle_secs or 0):
                go = False
        if state["evaluated_step"] == estimator.global_step:
            go = False
        if not _agree(estimator._shard, go):
            return
        state["last_eval"], state["evaluated_step"] = now, estimator.global_step
        estimator.evaluate(eval_spec.input_fn, steps=eval_spec.steps)
        exporters = eval_spec.exporters
        if exporters is not None:                             # (multi-GPU: every rank writes its shard, the chief the manifest)
            for ex in (exporters if isinstance(exporters, (list, tuple)) else [exporters]):
                ex.export(estimator, os.path.join(estimator.model_dir, "export"))
    estimator.train(train_spec.input_fn, max_steps=train_spec.max_steps, on_checkpoint=after_checkpoint)
    after_checkpoint(final=True)
    return estimator
