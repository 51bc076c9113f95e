"""Wide & Deep trainer — counterpart of the reference's ``trainers/linear_deep.py`` (canned
``tf.estimator.DNNLinearCombinedClassifier``: wide logits + deep logits, Ftrl on the wide part and
Adagrad on the deep part in one step)."""
from mi355x_rec.canned import DNNLinearCombinedClassifier
from trainers import _cli


def train_and_evaluate(args):
    return _cli.run(args, lambda columns, config: DNNLinearCombinedClassifier(
        model_dir=args.job_dir, linear_feature_columns=columns["linear"], dnn_feature_columns=columns["deep"],
        dnn_hidden_units=args.hidden_units, dnn_dropout=args.dropout, config=config))


if __name__ == "__main__":
    train_and_evaluate(_cli.make_parser("linear_deep", ("hidden_units", "dropout")).parse_args())
