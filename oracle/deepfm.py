"""DeepFM / Wide&Deep forward, backward and train step in numpy.  Oracle: tests only.

Follows the reference graph ``trainers/deep_fm.py:36-125`` line by line (citations on each block)
plus the TensorFlow-1.12 semantics of SURVEY Appendix A for everything the reference leaves to
``tf.feature_column`` / ``tf.layers`` / the estimator head.  Deliberately un-fused: one table per
field, a materialised [B, d, E] tensor, separate reductions — the way the TF graph runs on CPU —
so that it also serves as the timed ``cpu_baseline`` of bench.py.  PARITY UNPINNED (see
``oracle/__init__.py``).

dtype: every array in ``Params`` decides the arithmetic (float32 = TF's precision, float64 =
tolerance reference).
"""
import numpy as np

from . import optimizers as opt


class Params:
    """Model variables (names of SURVEY A.8 in comments).  Fields are in SORTED column-name order
    (SURVEY A.2); field f owns rows [field_off[f], field_off[f+1]) of the per-field tables."""

    def __init__(self, emb, lin_w, lin_bias, mlp, num_emb=None, lin_num=None):
        self.emb = emb            # list of [V_f, E]   input_layer/<col>_embedding/embedding_weights
        self.lin_w = lin_w        # list of [V_f]      linear/linear_model/<col>/weights
        self.lin_bias = lin_bias  # [1]                linear/linear_model/bias_weights
        self.mlp = mlp            # list of (kernel [in,out], bias [out]); last = logits layer
        self.num_emb = num_emb    # [n_d, E]           input_layer/numeric_embeddings
        self.lin_num = lin_num    # [n_d]              linear_model weights of numeric columns

    @property
    def dtype(self):
        return self.mlp[0][0].dtype if self.mlp else self.emb[0].dtype

    def astype(self, dt):
        c = lambda a: None if a is None else a.astype(dt)
        return Params([c(a) for a in self.emb], [c(a) for a in self.lin_w], c(self.lin_bias),
                      [(c(k), c(b)) for k, b in self.mlp], c(self.num_emb), c(self.lin_num))

    def dense_list(self):
        """Dense variables: MLP kernels / biases, linear bias, [numeric_embeddings], numeric linear
        weights (num_emb is None for the canned estimators' raw numeric columns)."""
        out = []
        for k, b in self.mlp:
            out += [k, b]
        out.append(self.lin_bias)
        if self.num_emb is not None:
            out.append(self.num_emb)
        if self.lin_num is not None:
            out.append(self.lin_num)
        return out

    def wide_flags(self):
        """Per entry of dense_list(): True for the variables under TF's ``linear`` scope."""
        out = [False] * (2 * len(self.mlp)) + [True]
        if self.num_emb is not None:
            out.append(False)
        if self.lin_num is not None:
            out.append(True)
        return out


def init_params(rng, vocab_sizes, E, hidden_units, n_numeric=0, dtype=np.float32, lin_scale=0.0,
                use_dnn=True, numeric="embed", field_dims=None, wide_fields=None, deep_numeric=None):
    """TF initialisers: embeddings truncated_normal(0, 1/sqrt(E)) at 2 sigma,
    linear weights / biases zero, dense kernels glorot-uniform.  ``lin_scale`` > 0 replaces the
    zero linear init by N(0, lin_scale) so parity tests exercise that path with non-trivial data.
    (TF's Philox stream is not reproducible; tests inject identical weights on both sides.)
    field_dims (canned DNNLinearCombinedClassifier with per-column embedding dimensions, or a column that only the
    wide part uses: dimension 0): per-field embedding widths instead of one E; wide_fields: per-field flags, False = the
    column is not in linear_feature_columns (its linear weights stay zero and are never touched); deep_numeric: per raw
    numeric column, False = not in dnn_feature_columns (kernel_0 has no row for it)."""
    def trunc_normal(shape, std):
        x = rng.standard_normal(shape)
        bad = np.abs(x) > 2
        while bad.any():
            x[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(x) > 2
        return (x * std).astype(dtype)

    def glorot(fan_in, fan_out, shape):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, shape).astype(dtype)

    dims = [E] * len(vocab_sizes) if field_dims is None else [int(d) for d in field_dims]
    emb = [trunc_normal((v, dd), 1.0 / np.sqrt(max(dd, 1))) for v, dd in zip(vocab_sizes, dims)]
    lin_w = [(rng.standard_normal(v) * lin_scale).astype(dtype) for v in vocab_sizes]
    if wide_fields is not None:
        lin_w = [w if on else np.zeros_like(w) for w, on in zip(lin_w, wide_fields)]
    lin_bias = np.zeros(1, dtype)
    d = len(vocab_sizes) + (n_numeric if numeric == "embed" else 0)
    mlp = []
    if use_dnn:
        n_raw = (n_numeric if deep_numeric is None else int(np.sum(deep_numeric))) if numeric == "raw" else 0
        fan = (d * E if field_dims is None else sum(dims)) + n_raw
        for h in list(hidden_units) + [1]:
            mlp.append((glorot(fan, h, (fan, h)), np.zeros(h, dtype)))
            fan = h
    num_emb = glorot(n_numeric, E, (n_numeric, E)) if (n_numeric and numeric == "embed") else None  # deep_fm.py:64
    lin_num = (rng.standard_normal(n_numeric) * lin_scale).astype(dtype) if n_numeric else None
    return Params(emb, lin_w, lin_bias, mlp, num_emb, lin_num)


ACTIVATIONS = {   # params["activation"] (deep_fm.py:22): name -> (f, f' expressed through the output y)
    "relu": (lambda v: np.maximum(v, 0), lambda y: (y > 0).astype(y.dtype)),
    "sigmoid": (lambda v: 1 / (1 + np.exp(-v)), lambda y: y * (1 - y)),
    "tanh": (np.tanh, lambda y: 1 - y * y),
    None: (lambda v: v, lambda y: np.ones_like(y)),
}


def forward(p, ids, x_num=None, use_linear=True, use_mf=True, use_dnn=True, dropout_masks=None, numeric="embed",
            keep_prob=1.0, activation="relu", relu_masks=None, wide_fields=None, deep_numeric=None, wide_numeric=None):
    """model_fn forward (deep_fm.py:36-115).  ids [B,F] per-field local ids; x_num [B,n_d].
    dropout_masks: per hidden layer, a [B,h] array of {0, 1} keep flags (TRAIN) or None; tf.layers.dropout
    (deep_fm.py:102-103) is tf.nn.dropout: div(x, keep_prob) * mask — a division, not a multiplication by 1/keep.
    numeric: "embed" = DeepFM's numeric_embeddings (deep_fm.py:62-73: x[B,n_d,1] * V[1,n_d,E]);
    "raw" = the canned estimators' input_layer, where a numeric_column contributes its value itself to
    the concat (trainers/linear_deep.py:32-39 with numeric columns in dnn_feature_columns) —
    no FM term exists there.  The engine keeps numeric columns after the categorical block in both
    forms; TF's input_layer interleaves them by column name, which only permutes kernel_0's rows.
    relu_masks (tests only; relu only): per hidden layer a [B,h] bool array that REPLACES the sign test of relu —
    unit (b, j) passes its pre-activation iff relu_masks[i][b, j].  A pre-activation that is 0 to within the rounding
    of its dot product takes either side depending on summation order; a test that compares multi-step trajectories
    hands the oracle the device's decisions (after checking that they differ from its own only on such units), so
    that both follow the same branch.  c["pre"] holds every hidden layer's pre-activations.
    Returns a cache with logits [B] and every intermediate the backward needs."""
    dt = p.dtype
    B, F = ids.shape
    if numeric == "raw" and use_mf:
        raise ValueError("raw numeric columns belong to the canned estimators, which have no FM term")
    c = {"ids": ids, "x_num": x_num, "flags": (use_linear, use_mf, use_dnn), "numeric": numeric, "keep_prob": keep_prob,
         "activation": activation, "wide_fields": wide_fields, "widths": [a.shape[1] for a in p.emb],
         "deep_numeric": deep_numeric, "wide_numeric": wide_numeric}
    # the canned DNNLinearCombinedClassifier's general form: every column has its own embedding width, and
    # the wide and the deep part each read their own subset of the columns
    c["general"] = len(set(c["widths"])) > 1 or deep_numeric is not None or wide_numeric is not None
    if c["general"] and (use_mf or (x_num is not None and numeric == "embed")):
        raise ValueError("per-column widths / column subsets belong to the canned estimators (no FM term, raw numeric columns)")
    logits = np.zeros(B, dt)                                      # deep_fm.py:36
    if use_linear:                                                # deep_fm.py:37-44 linear_model
        lin = np.zeros(B, dt)
        for f in range(F):                                        # add_n over sorted columns
            if wide_fields is None or wide_fields[f]:             # (a column of dnn_feature_columns only: no linear weight)
                lin = lin + p.lin_w[f][ids[:, f]]
        if x_num is not None:
            for j in range(x_num.shape[1]):
                if wide_numeric is None or wide_numeric[j]:
                    lin = lin + x_num[:, j] * p.lin_num[j]
        lin = lin + p.lin_bias[0]                                 # bias_add
        c["lin"] = lin
        logits = logits + lin
    if use_mf or use_dnn:                                         # deep_fm.py:47-73 input layer
        parts = [p.emb[f][ids[:, f]] for f in range(F)]           # mean of one id = the row
        E = parts[0].shape[1] if parts else p.num_emb.shape[1]    # numeric columns only: deep_fm.py:57-70
        if x_num is not None and numeric == "embed":              # deep_fm.py:62-70
            parts += [x_num[:, j:j + 1] * p.num_emb[j][None, :] for j in range(x_num.shape[1])]
        d = len(parts)
        if x_num is not None and numeric == "raw":                # canned input_layer: the value itself
            parts.append((x_num if deep_numeric is None else x_num[:, np.asarray(deep_numeric, bool)]).astype(dt))
        concat = np.concatenate(parts, 1)                         # [B, d*E (+ n_d raw)]
        c["concat"] = concat
    if use_mf:                                                    # deep_fm.py:76-90
        mat = concat.reshape(B, d, E)                             # :79
        s = mat.sum(1)
        sum_square = s * s                                        # :81
        square_sum = (mat * mat).sum(1)                           # :83
        fm = dt.type(0.5) * (sum_square - square_sum).sum(1)      # :87
        c["sumv"], c["fm"] = s, fm
        logits = logits + fm
    if use_dnn:                                                   # deep_fm.py:93-111
        net = concat
        acts = []
        pres = []
        nh = len(p.mlp) - 1
        for i in range(nh):
            k, b = p.mlp[i]
            pre = net @ k + b
            pres.append(pre)
            if relu_masks is not None and relu_masks[i] is not None:
                assert activation == "relu"
                net = np.where(relu_masks[i], pre, 0).astype(dt)
            else:
                net = ACTIVATIONS[activation][0](pre).astype(dt)   # tf.layers.dense(activation) :100
            if dropout_masks is not None and dropout_masks[i] is not None:
                net = (net / dt.type(keep_prob)) * dropout_masks[i].astype(dt)   # tf.layers.dropout :102-103
            acts.append(net)
        k, b = p.mlp[nh]
        dnn = (net @ k + b)[:, 0]                                 # :108
        c["acts"], c["dnn"], c["pre"], c["relu_masks"] = acts, dnn, pres, relu_masks
        logits = logits + dnn
    c["logits"] = logits
    return c


def head(logits, labels, reduction="mean", global_batch=None):
    """binary_classification_head (deep_fm.py:118-125).  reduction "mean" =
    contrib head (SUM_OVER_BATCH_SIZE), "sum" = canned estimators.  Returns loss, d_logits,
    per-example loss and sigmoid."""
    dt = logits.dtype
    y = labels.astype(dt)
    x = logits
    per = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    e = np.exp(-np.abs(x))
    sig = np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(dt)
    n = global_batch if global_batch is not None else len(x)
    scale = dt.type(1.0 / n) if reduction == "mean" else dt.type(1)
    loss = (per * scale).sum(dtype=dt)
    return loss, (sig - y) * scale, per, sig


def predictions(logits):
    """model_utils.py:9-20 get_binary_predictions (probabilities == logistic there)."""
    e = np.exp(-np.abs(logits))
    sig = np.where(logits >= 0, 1 / (1 + e), e / (1 + e)).astype(logits.dtype)
    return {"logits": logits, "logistic": sig, "probabilities": sig,
            "class_id": (sig > 0.5).astype(np.int32)}


def backward(p, c, d_logits, dropout_masks=None):
    """Gradients of sum(d_logits * logits).  Returns (dense grads aligned with Params.dense_list(),
    per-entry sparse grads: d_rows [B,F,E] for the embedding rows, d_lin [B,F] for linear rows)."""
    use_linear, use_mf, use_dnn = c["flags"]
    ids, x_num = c["ids"], c["x_num"]
    raw = c.get("numeric", "embed") == "raw" and x_num is not None
    B, F = ids.shape
    dt = p.dtype
    g_mlp = []
    d_concat = None
    if use_dnn:
        nh = len(p.mlp) - 1
        k, b = p.mlp[nh]
        top = c["acts"][-1] if nh else c["concat"]
        g_last = (top.T @ d_logits[:, None], d_logits.sum(keepdims=True))
        d_net = d_logits[:, None] * k[:, 0][None, :]
        g_hidden = []
        for i in range(nh - 1, -1, -1):
            a = c["acts"][i]                 # post-relu, post-dropout activation
            dropped = dropout_masks is not None and dropout_masks[i] is not None
            if dropped:
                d_net = (d_net * dropout_masks[i].astype(dt)) / dt.type(c["keep_prob"])
            # relu: a > 0  <=>  (pre-activation > 0 and the unit was kept); dropped units already got 0.
            # other activations: f'(pre) through the un-dropped output a * keep
            act = c.get("activation", "relu")
            y = a * dt.type(c["keep_prob"]) if (dropped and act != "relu") else a
            rm = c.get("relu_masks")
            if rm is not None and rm[i] is not None:
                d_pre = d_net * rm[i].astype(dt)          # (a dropped unit's d_net is already 0)
            else:
                d_pre = d_net * ACTIVATIONS[act][1](y).astype(dt)
            inp = c["acts"][i - 1] if i else c["concat"]
            k_i = p.mlp[i][0]
            g_hidden.append((inp.T @ d_pre, d_pre.sum(0)))
            d_net = d_pre @ k_i.T
        g_mlp = list(reversed(g_hidden)) + [g_last]
        d_concat = d_net
    widths = c.get("widths") or []
    if c.get("general"):
        # per-field embedding widths / column subsets (canned estimators: no FM term, raw numeric columns): slices of d_concat
        dense = []
        for gk, gb in g_mlp:
            dense += [gk, gb]
        dense.append(d_logits.sum(keepdims=True) if use_linear else np.zeros(1, dt))
        if x_num is not None:
            g_lnum = ((d_logits[:, None] * x_num).sum(0) if use_linear else np.zeros_like(p.lin_num)).astype(dt)
            if c.get("wide_numeric") is not None:
                g_lnum[~np.asarray(c["wide_numeric"], bool)] = 0
            dense.append(g_lnum)
        offs = np.concatenate([[0], np.cumsum(widths)])
        d_rows = [d_concat[:, offs[f]:offs[f + 1]] if d_concat is not None else np.zeros((B, widths[f]), dt) for f in range(F)]
        d_lin = np.broadcast_to(d_logits[:, None], (B, F)).copy() if use_linear else None
        if d_lin is not None and c.get("wide_fields") is not None:
            d_lin[:, ~np.asarray(c["wide_fields"], bool)] = 0
        return dense, d_rows, d_lin
    n_emb_num = 0 if (x_num is None or raw) else x_num.shape[1]    # numeric columns that own an embedding
    d = F + n_emb_num
    E = (c["concat"].shape[1] - (x_num.shape[1] if raw else 0)) // d if (use_mf or use_dnn) else 0
    g_v = None
    if use_mf or use_dnn:
        g_v = np.zeros((B, d, E), dt)
        if d_concat is not None:
            g_v += d_concat[:, :d * E].reshape(B, d, E)
        if use_mf:                            # d fm / d v_j = S - v_j
            mat = c["concat"].reshape(B, d, E)
            g_v += d_logits[:, None, None] * (c["sumv"][:, None, :] - mat)
    dense = []
    for gk, gb in g_mlp:
        dense += [gk, gb]
    dense.append(d_logits.sum(keepdims=True) if use_linear else np.zeros(1, dt))
    if x_num is not None:
        if raw:
            g_num = None                      # a raw numeric column has no variable of its own in the deep part
        else:
            g_num = np.einsum("bj,bje->je", x_num, g_v[:, F:, :]) if g_v is not None else np.zeros_like(p.num_emb)
        g_lnum = (d_logits[:, None] * x_num).sum(0) if use_linear else np.zeros_like(p.lin_num)
        dense += ([] if raw else [g_num.astype(dt)]) + [g_lnum.astype(dt)]
    d_rows = g_v[:, :F, :] if g_v is not None else None
    d_lin = np.broadcast_to(d_logits[:, None], (B, F)).copy() if use_linear else None
    if d_lin is not None and c.get("wide_fields") is not None:
        d_lin[:, ~np.asarray(c["wide_fields"], bool)] = 0
    return dense, d_rows, d_lin


class TrainState:
    """Optimizer slots for every variable + global step.

    lin_hp: the canned DNNLinearCombinedClassifier's second optimizer (reference
    trainers/linear_deep.py:32-39).  TF 1.12's ``_dnn_linear_combined_model_fn`` builds ONE forward
    and ONE loss, then ``dnn_optimizer.minimize(loss, var_list=<variables under the dnn scope>)`` and
    ``linear_optimizer.minimize(loss, var_list=<variables under the linear scope>)``, groups the two
    and increments global_step once.  Linear scope = linear_model weights of every column (categorical
    [vocab,1] tables, numeric [1,1] weights) + bias_weights; dnn scope = embeddings, hidden layers,
    logits layer (and, for a DeepFM given two optimizers, numeric_embeddings: an input-layer variable).
    Both gradients are taken at the same pre-update variables."""

    def __init__(self, p, hp, lin_hp=None):
        self.hp = hp
        self.lin_hp = lin_hp if lin_hp is not None else hp
        self.step = 0
        self.emb = [opt.slot_init(hp, a) for a in p.emb]
        self.lin = [opt.slot_init(self.lin_hp, a) for a in p.lin_w]
        self.wide = p.wide_flags()           # per dense variable: does it belong to the linear scope
        self.dense = [opt.slot_init(self.lin_hp if w else hp, a) for a, w in zip(p.dense_list(), self.wide)]
        self.powers = opt.AdamPowers(hp, p.dtype) if hp.name == opt.ADAM else None
        self.lin_powers = self.powers if lin_hp is None else (
            opt.AdamPowers(lin_hp, p.dtype) if lin_hp.name == opt.ADAM else None)


def train_step(p, st, ids, labels, x_num=None, use_linear=True, use_mf=True, use_dnn=True,
               reduction="mean", dropout_masks=None, global_batch=None, numeric="embed", keep_prob=1.0, activation="relu",
               relu_masks=None, wide_fields=None, deep_numeric=None, wide_numeric=None):
    """One optimizer.minimize(loss) (deep_fm.py:119-125 TRAIN branch): forward, head, backward,
    apply_gradients (dense vars: fused Apply*, embedding / linear tables: sparse apply with
    duplicate-summing), beta powers / global_step update.  Returns (loss, logits)."""
    c = forward(p, ids, x_num, use_linear, use_mf, use_dnn, dropout_masks, numeric, keep_prob, activation, relu_masks, wide_fields,
                deep_numeric, wide_numeric)
    loss, d_logits, _, _ = head(c["logits"], labels, reduction, global_batch)
    dense_g, d_rows, d_lin = backward(p, c, d_logits, dropout_masks)
    apply_gradients(p, st, ids, dense_g, d_rows, d_lin, wide_fields)
    return loss, c["logits"]


def apply_gradients(p, st, ids, dense_g, d_rows, d_lin, wide_fields=None):
    hp, lhp = st.hp, st.lin_hp
    lr_t = st.powers.lr_t(hp.lr) if st.powers else None
    lin_lr_t = st.lin_powers.lr_t(lhp.lr) if st.lin_powers else None
    F = ids.shape[1]
    for var, (s0, s1), g, wide in zip(p.dense_list(), st.dense, dense_g, st.wide):
        if wide:
            opt.dense_apply(lhp, var, s0, s1, g.reshape(var.shape).astype(var.dtype), lin_lr_t)
        else:
            opt.dense_apply(hp, var, s0, s1, g.reshape(var.shape).astype(var.dtype), lr_t)
    for f in range(F):
        if d_rows is not None and p.emb[f].shape[1]:              # (width 0: the column is not in the deep part)
            g_f = d_rows[f] if isinstance(d_rows, list) else d_rows[:, f, :]
            opt.sparse_apply(hp, p.emb[f], st.emb[f][0], st.emb[f][1], ids[:, f], g_f, lr_t)
        if d_lin is not None and (wide_fields is None or wide_fields[f]):
            w = p.lin_w[f][:, None]
            s0, s1 = st.lin[f][0][:, None], st.lin[f][1][:, None]
            opt.sparse_apply(lhp, w, s0, s1, ids[:, f], d_lin[:, f:f + 1], lin_lr_t)
    if st.powers:
        st.powers.finish()
    if st.lin_powers is not None and st.lin_powers is not st.powers:
        st.lin_powers.finish()
    st.step += 1


def fm_pairwise(mat):
    """sum_{i<j} <v_i, v_j> — the identity deep_fm.py:81-87 implements (known-answer check)."""
    B, d, E = mat.shape
    out = np.zeros(B, np.float64)
    m = mat.astype(np.float64)
    for i in range(d):
        for j in range(i + 1, d):
            out += (m[:, i] * m[:, j]).sum(1)
    return out


def layer_summary(x):
    """model_utils.py:4-6: tf.nn.zero_fraction + the histogram's range."""
    return {"fraction_of_zero_values": float((x == 0).mean()), "min": float(x.min()),
            "max": float(x.max()), "mean": float(x.mean())}
