"""Drop-in counterparts of the reference's ``trainers`` package: same module paths, CLI flags,
defaults and function names, running on the MI355X engine (``mi355x_rec``) instead of TensorFlow.
Run from this directory's parent, e.g. ``python -m trainers.deep_fm --train-steps 2000``."""
