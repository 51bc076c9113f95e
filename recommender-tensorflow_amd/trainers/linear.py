"""Linear (wide) classifier trainer — counterpart of the reference's ``trainers/linear.py``
(canned ``tf.estimator.LinearClassifier``: Ftrl, sum-reduced sigmoid cross-entropy).
``--embedding-size`` is accepted and irrelevant here, as in the reference (Appendix C.3)."""
from mi355x_rec.canned import LinearClassifier
from trainers import _cli


def train_and_evaluate(args):
    return _cli.run(args, lambda columns, config: LinearClassifier(
        feature_columns=columns["linear"], model_dir=args.job_dir, config=config))


if __name__ == "__main__":
    train_and_evaluate(_cli.make_parser("linear").parse_args())
