"""DNN classifier trainer — counterpart of the reference's ``trainers/deep.py`` (canned
``tf.estimator.DNNClassifier`` over the embedding columns: Adagrad(0.05), dropout in TRAIN,
sum-reduced loss)."""
from mi355x_rec.canned import DNNClassifier
from trainers import _cli


def train_and_evaluate(args):
    return _cli.run(args, lambda columns, config: DNNClassifier(
        hidden_units=args.hidden_units, feature_columns=columns["deep"], model_dir=args.job_dir,
        dropout=args.dropout, config=config))


if __name__ == "__main__":
    train_and_evaluate(_cli.make_parser("deep", ("hidden_units", "dropout")).parse_args())
