"""Run configuration (counterpart of the reference's ``trainers/conf_utils.py``)."""
from mi355x_rec.estimator import EvalSpec, LatestExporter, RunConfig, TrainSpec

EVAL_INTERVAL = 60  # seconds between checkpoints = between evaluations


def get_run_config():
    return RunConfig(save_checkpoints_secs=EVAL_INTERVAL, keep_checkpoint_max=5)


def get_train_spec(input_fn, train_steps):
    return TrainSpec(input_fn=input_fn, max_steps=train_steps)


def get_exporter(serving_input_fn):
    return LatestExporter(name="exporter", serving_input_receiver_fn=serving_input_fn)


def get_eval_spec(input_fn, exporter):
    # steps=None: until the eval input is exhausted
    return EvalSpec(input_fn=input_fn, steps=None, exporters=exporter, start_delay_secs=min(EVAL_INTERVAL, 120),
                    throttle_secs=EVAL_INTERVAL)
