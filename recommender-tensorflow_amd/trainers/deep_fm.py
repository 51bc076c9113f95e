"""DeepFM trainer: ``model_fn(features, labels, mode, params)`` and the CLI.

Counterpart of the reference's ``trainers/deep_fm.py``: model_fn reads the same params keys with
the same defaults (:13-26) and raises the same two ValueErrors (:31-34); instead of building a TF
graph it binds (once) an ``mi355x_rec.engine.DeepFM`` — linear + FM + DNN logits summed, sigmoid
cross-entropy head with the mean reduction of tf.contrib's binary_classification_head — and runs
the batch on it.  Divergence kept on purpose: ``--exclude-linear/-mf/-dnn``
work here, whereas trailing commas at :135-137 make them no-ops in the reference."""
from mi355x_rec.engine import DeepFM
from mi355x_rec.estimator import Estimator
from mi355x_rec.model import run_batch
from trainers import _cli
from trainers.model_utils import get_optimizer


def model_fn(features, labels, mode, params):
    cat = params.get("categorical_columns", [])
    num = params.get("numeric_columns", [])
    flags = [params.get(k, True) for k in ("use_linear", "use_mf", "use_dnn")]
    if len(cat) + len(num) == 0:
        raise ValueError("At least 1 feature column of categorical_columns or numeric_columns must be specified.")
    if not any(flags):
        raise ValueError("At least 1 of linear, mf or dnn component must be used.")
    activation = params.get("activation", "relu")      # deep_fm.py:22 (a callable there): "relu" | "sigmoid" | "tanh" | None,
                                                       # or a callable with one of those names

    def make(plan, device, shard=None):
        opt = get_optimizer(params.get("optimizer", "Adam"), params.get("learning_rate", 0.001))
        return DeepFM(plan.vocab_sizes, n_numeric=len(plan.numeric), embedding_size=params.get("embedding_size", 4),
                      hidden_units=params.get("hidden_units", [16, 16]), use_linear=flags[0], use_mf=flags[1],
                      use_dnn=flags[2], dropout=params.get("dropout", 0), optimizer=opt, reduction="mean",
                      device=device, seed=params.get("seed", 0), shard=shard, activation=activation,
                      catchup=params.get("catchup", "bounded"))

    return run_batch(features, labels, mode, params, make)


def train_and_evaluate(args):
    def make_estimator(columns, config):
        return Estimator(model_fn=model_fn, model_dir=args.job_dir, config=config, params={
            "categorical_columns": columns["linear"],
            "use_linear": not args.exclude_linear, "use_mf": not args.exclude_mf, "use_dnn": not args.exclude_dnn,
            "embedding_size": args.embedding_size, "hidden_units": args.hidden_units, "dropout": args.dropout,
        })
    return _cli.run(args, make_estimator)


if __name__ == "__main__":
    train_and_evaluate(_cli.make_parser("deep_fm", ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units",
                                                    "dropout")).parse_args())
