"""TF-1.12 optimizer update rules in numpy.  Oracle: tests only.

Restates what ``get_optimizer`` (reference ``trainers/model_utils.py:57-66``) hands to
``head.create_estimator_spec`` (``trainers/deep_fm.py:117-125``): tf.train.{Adam, Adagrad, Ftrl,
RMSProp, GradientDescent}Optimizer(learning_rate=lr) with TF-1.12 defaults, and the canned
estimators' defaults.  TensorFlow itself is absent (un-vendored dependency,
``environment.yml:10``): the op order below is recalled from TF 1.12's
``python/training/{adam,adagrad,ftrl,rmsprop}.py`` and ``core/kernels/training_ops.cc`` —
PARITY UNPINNED, see ``oracle/__init__.py``.

All arithmetic is done in the dtype of the arrays passed in (float32 reproduces TF's sequence of
roundings; float64 is the reference for tolerance tests).  No fused multiply-adds: numpy does
one rounding per operation, as Eigen's CPU expressions do.
"""
import numpy as np

ADAM, ADAGRAD, FTRL, RMSPROP, SGD = "Adam", "Adagrad", "Ftrl", "RMSProp", "SGD"
NAMES = (ADAM, ADAGRAD, FTRL, RMSPROP, SGD)


class Hyper:
    """Constructor arguments with TF-1.12 defaults."""

    def __init__(self, name=ADAM, lr=0.001, beta1=0.9, beta2=0.999, epsilon=None, decay=0.9,
                 momentum=0.0, lr_power=-0.5, initial_accumulator_value=0.1, l1=0.0, l2=0.0):
        if name not in NAMES:
            raise KeyError(name)  # model_utils.py:65 raises KeyError for unknown names
        self.name = name
        self.lr = lr
        self.beta1, self.beta2 = beta1, beta2
        self.epsilon = epsilon if epsilon is not None else (1e-8 if name == ADAM else 1e-10)
        self.decay, self.momentum = decay, momentum
        self.lr_power = lr_power
        self.initial_accumulator_value = initial_accumulator_value
        self.l1, self.l2 = l1, l2


def slot_init(hp, like):
    """(slot0, slot1) for a variable shaped like ``like``."""
    z = np.zeros_like(like)
    if hp.name == ADAM:
        return z.copy(), z.copy()                                     # m, v
    if hp.name == ADAGRAD:
        return np.full_like(like, hp.initial_accumulator_value), z    # accum
    if hp.name == FTRL:
        return np.full_like(like, hp.initial_accumulator_value), z.copy()  # accum, linear
    if hp.name == RMSPROP:
        return np.ones_like(like), z.copy()                           # ms (init 1.0), mom
    return z, z


class AdamPowers:
    """beta1_power / beta2_power non-slot variables: start at beta, multiplied by beta
    after every apply, all in the variable dtype."""

    def __init__(self, hp, dtype):
        self.dt = np.dtype(dtype).type
        self.b1 = self.dt(hp.beta1)
        self.b2 = self.dt(hp.beta2)
        self.b1p = self.dt(hp.beta1)
        self.b2p = self.dt(hp.beta2)

    def lr_t(self, lr):
        one = self.dt(1)
        return self.dt(lr) * np.sqrt(one - self.b2p) / (one - self.b1p)

    def finish(self):
        self.b1p = self.dt(self.b1p * self.b1)
        self.b2p = self.dt(self.b2p * self.b2)


def dense_apply(hp, var, s0, s1, g, lr_t=None):
    """In-place fused Apply* on a dense variable (training_ops.cc functors)."""
    dt = var.dtype.type
    if hp.name == ADAM:
        b1, b2, eps = dt(hp.beta1), dt(hp.beta2), dt(hp.epsilon)
        s0 += (g - s0) * (dt(1) - b1)
        s1 += (g * g - s1) * (dt(1) - b2)
        var -= (s0 * dt(lr_t)) / (np.sqrt(s1) + eps)
    elif hp.name == ADAGRAD:
        s0 += g * g
        var -= g * dt(hp.lr) * (dt(1) / np.sqrt(s0))
    elif hp.name == FTRL:
        _ftrl(hp, var, s0, s1, g)
    elif hp.name == RMSPROP:
        s0 += (g * g - s0) * (dt(1) - dt(hp.decay))
        s1[...] = s1 * dt(hp.momentum) + (g * dt(hp.lr)) / np.sqrt(s0 + dt(hp.epsilon))
        var -= s1
    else:
        var -= g * dt(hp.lr)


def _ftrl(hp, var, accum, linear, g):
    dt = var.dtype.type
    lr = dt(hp.lr)
    if hp.lr_power != -0.5:
        raise NotImplementedError("only lr_power=-0.5 (TF default) is restated")
    new_accum = accum + g * g
    linear += g - (np.sqrt(new_accum) - np.sqrt(accum)) / lr * var
    adj = np.clip(linear, -dt(hp.l1), dt(hp.l1))
    var[...] = (adj - linear) / (np.sqrt(new_accum) / lr + dt(2) * dt(hp.l2))
    accum[...] = new_accum


def dedup(indices, values):
    """_deduplicate_indexed_slices: unique + unsorted_segment_sum.  Rows are summed
    in order of occurrence (CPU kernel order); unique ids returned sorted ascending — the row
    order is immaterial because every optimizer acts row-wise."""
    order = np.argsort(indices, kind="stable")
    si = indices[order]
    starts = np.flatnonzero(np.r_[True, si[1:] != si[:-1]])
    uniq = si[starts]
    out = np.zeros((len(uniq),) + values.shape[1:], values.dtype)
    ends = np.r_[starts[1:], len(si)]
    maxlen = int((ends - starts).max()) if len(starts) else 0
    for k in range(maxlen):               # k-th occurrence of every row, in occurrence order
        sel = starts + k < ends
        out[sel] += values[order[(starts + k)[sel]]]
    return uniq, out


def sparse_apply(hp, var, s0, s1, indices, values, lr_t=None):
    """_apply_sparse_duplicate_indices on a [rows, width] variable; ``values`` one row per index."""
    dt = var.dtype.type
    uniq, g = dedup(indices, values)
    if hp.name == ADAM:
        # adam.py _apply_sparse_shared: assign(m, m*beta1) on ALL rows, scatter_add, same for v,
        # then var -= lr*m/(sqrt(v)+eps) on ALL rows.
        b1, b2, eps = dt(hp.beta1), dt(hp.beta2), dt(hp.epsilon)
        s0 *= b1
        s0[uniq] += g * (dt(1) - b1)
        s1 *= b2
        s1[uniq] += (g * g) * (dt(1) - b2)
        var -= (dt(lr_t) * s0) / (np.sqrt(s1) + eps)
        return uniq
    v, a, b = var[uniq], s0[uniq], s1[uniq]
    if hp.name == ADAGRAD:
        a += g * g
        v -= g * dt(hp.lr) * (dt(1) / np.sqrt(a))
    elif hp.name == FTRL:
        _ftrl(hp, v, a, b, g)
    elif hp.name == RMSPROP:
        a += (g * g - a) * (dt(1) - dt(hp.decay))
        b[...] = b * dt(hp.momentum) + (g * dt(hp.lr)) / np.sqrt(a + dt(hp.epsilon))
        v -= b
    else:
        v -= g * dt(hp.lr)
    var[uniq] = v
    if hp.name != SGD:
        s0[uniq] = a
    if hp.name in (FTRL, RMSPROP):
        s1[uniq] = b
    return uniq
