"""The four factory functions every trainer calls to configure its Estimator run.

Same names and arguments as the reference module of this name (trainers/conf_utils.py:6-34 there),
so `trainers.*` and user scripts keep working; the objects returned are this package's
(`mi355x_rec.estimator`).  All timing derives from one number: a checkpoint — and therefore an
evaluation plus an export — every EVAL_INTERVAL seconds, at most `KEEP` checkpoints on disk.
"""
from mi355x_rec import estimator as _est

EVAL_INTERVAL = 60
KEEP = 5


def get_run_config():
    """Checkpoint cadence of the run (the reference saves every minute and keeps five)."""
    cfg = _est.RunConfig()
    cfg.save_checkpoints_secs = EVAL_INTERVAL
    cfg.keep_checkpoint_max = KEEP
    return cfg


def get_train_spec(input_fn, train_steps):
    """Train until the global step reaches `train_steps`."""
    return _est.TrainSpec(input_fn, train_steps)


def get_exporter(serving_input_fn):
    """Export the latest model after each evaluation, under <job_dir>/export/exporter."""
    return _est.LatestExporter("exporter", serving_input_fn)


def get_eval_spec(input_fn, exporter):
    """Evaluate on the whole eval input (no step limit) after each checkpoint."""
    whole_input, first_eval_after = None, min(EVAL_INTERVAL, 120)
    return _est.EvalSpec(input_fn, whole_input, exporter, first_eval_after, EVAL_INTERVAL)
