"""TensorFlow-1.12 variable names for the engine's parameters.

The reference's checkpoints are TF tensor bundles written by ``tf.estimator`` (conf_utils.py:6-10).
Reading that format needs TensorFlow; naming does not.  Someone who has a TF-1.12 checkpoint dumps
it once with TF itself

    reader = tf.train.load_checkpoint(model_dir)
    np.savez("vars.npz", **{n: reader.get_tensor(n) for n in reader.get_variable_to_shape_map()})

and ``import_variables`` maps those names onto the fused table / flat dense buffer;
``export_variables`` writes the same names back (so a run here can be inspected or resumed there
with ``tf.train.init_from_checkpoint``-style assignment).  Names follow the scopes of
trainers/deep_fm.py:38,48,94,99,107 for ``model="deep_fm"`` and the canned estimators' for
``"linear" | "dnn" | "dnn_linear_combined"`` (trainers/linear.py:30, deep.py:32, linear_deep.py:32).
Optimizer slots are not part of the mapping (a warm start re-creates them, as TF's
``warm_start_from`` does).  A row-sharded engine (N GPUs) imports its own rows of every table (row r of the stacked
tables lives on rank r % world at index r // world); exporting needs the whole table and is single-GPU only.
"""
import numpy as np

_MODELS = ("deep_fm", "linear", "dnn", "dnn_linear_combined")


def variable_names(model, column_names, n_hidden, n_numeric=0, numeric_names=None):
    """-> dict with keys 'emb' [F names], 'lin_w' [F names], 'lin_bias', 'mlp' [(kernel, bias)] (hidden
    layers then logits), 'num_emb' (DeepFM's numeric_embeddings), 'lin_num' (one name for all numeric
    linear weights, or — with numeric_names — a list of per-column linear_model weights [1,1])."""
    if model not in _MODELS:
        raise ValueError("model must be one of %s" % (_MODELS,))
    cols = list(column_names)
    if model == "deep_fm":
        emb = ["input_layer/input_layer/%s_embedding/embedding_weights" % c for c in cols]
        lin = ["linear/linear_model/%s/weights" % c for c in cols]
        mlp = [("dnn/dnn/hiddenlayer_%d/dense/kernel" % i, "dnn/dnn/hiddenlayer_%d/dense/bias" % i) for i in range(n_hidden)]
        mlp.append(("dnn/dnn/logits/dense/kernel", "dnn/dnn/logits/dense/bias"))
    else:
        emb = ["dnn/input_from_feature_columns/input_layer/%s_embedding/embedding_weights" % c for c in cols]
        lin = ["linear/linear_model/%s/weights" % c for c in cols]
        mlp = [("dnn/hiddenlayer_%d/kernel" % i, "dnn/hiddenlayer_%d/bias" % i) for i in range(n_hidden)]
        mlp.append(("dnn/logits/kernel", "dnn/logits/bias"))
    lin_num = None
    if n_numeric:
        lin_num = (["linear/linear_model/%s/weights" % c for c in numeric_names] if numeric_names
                   else "linear/linear_model/numeric/weights")
    return {"emb": emb, "lin_w": lin, "lin_bias": "linear/linear_model/bias_weights", "mlp": mlp,
            "num_emb": "input_layer/numeric_embeddings" if (n_numeric and model == "deep_fm") else None,
            "lin_num": lin_num}


def kernel0_rows(column_names, numeric_names, E):
    """Canned estimators with raw numeric columns: TF's input_layer orders the concat by column name over
    ``<categorical>_embedding`` (E columns each) and the numeric keys (1 column each); the engine keeps
    the numeric columns after the categorical block.  Returns perm with
    tf_kernel0[i] == engine_kernel0[perm[i]].  E: one width, or a list of per-column widths (0 = the column is not in
    dnn_feature_columns); numeric_names: the numeric columns the deep part reads."""
    dims = list(E) if isinstance(E, (list, tuple)) else [E] * len(column_names)
    starts = np.concatenate([[0], np.cumsum(dims)]).astype(int)
    items = [(c + "_embedding", int(starts[f]), dims[f]) for f, c in enumerate(column_names) if dims[f]]
    items += [(n, int(starts[-1]) + j, 1) for j, n in enumerate(numeric_names)]
    perm = []
    for _, start, width in sorted(items):
        perm.extend(range(start, start + width))
    return np.asarray(perm, np.int64)


def _names_for(m, model, column_names, numeric_names=None, sharded_ok=False):
    if m.shard is not None and not sharded_ok:
        raise ValueError("TF-named export works on single-GPU engines (a sharded engine holds table slices)")
    if len(column_names) != m.F:
        raise ValueError("%d column names for %d categorical fields" % (len(column_names), m.F))
    if m.raw_numeric and (numeric_names is None or len(numeric_names) != m.n_numeric):
        raise ValueError("raw numeric columns need their %d names (kernel row order, linear_model weights)" % m.n_numeric)
    return variable_names(model, column_names, len(m.layers) - 1 if m.use_dnn else 0, m.n_numeric,
                          numeric_names if m.raw_numeric else None)


def _subsets(m, numeric_names):
    """(per-column embedding widths, names of the numeric columns the deep part reads, flags of the numeric columns the
    wide part reads) — the whole lists unless the engine was built with column subsets (engine.DeepFM field_dims,
    wide_fields, deep_numeric, wide_numeric)."""
    dims = getattr(m, "field_dims", None) or [m.E] * m.F
    names = list(numeric_names or [])
    dn, wn = getattr(m, "deep_numeric", None), getattr(m, "wide_numeric", None)
    deep_names = [n for j, n in enumerate(names) if dn is None or dn[j]]
    wide_num = [True] * len(names) if wn is None else list(wn)
    return dims, deep_names, wide_num


def export_variables(m, column_names, model="deep_fm", numeric_names=None):
    """Engine -> {TF variable name: ndarray} with TF's shapes (linear weights [vocab, 1], bias [1],
    numeric embeddings [1, n_d, E] as deep_fm.py:64 creates them).  A column that only one of the wide / deep parts
    reads has only that part's variable, an embedding column its own dimension."""
    nm = _names_for(m, model, column_names, numeric_names)
    dims, deep_names, wide_num = _subsets(m, numeric_names)
    g = m.export_numpy()
    out = {}
    if g.get("emb") is not None:
        out.update({n: a for n, a in zip(nm["emb"], g["emb"]) if a.shape[1]})
    if g.get("lin_w") is not None:
        out.update({n: a.reshape(-1, 1) for n, a in zip(nm["lin_w"], g["lin_w"]) if a is not None})
        out[nm["lin_bias"]] = g["lin_bias"].reshape(1)
    if m.use_dnn:
        for i, ((kn, bn), (k, b)) in enumerate(zip(nm["mlp"], g["mlp"])):
            if i == 0 and m.raw_numeric:
                k = k[kernel0_rows(column_names, deep_names, dims)]
            out[kn], out[bn] = k, b
    if m.n_numeric:
        if nm["num_emb"] is not None:
            out[nm["num_emb"]] = g["num_emb"].reshape(1, m.n_numeric, m.E)
        if "lin_num" in g:
            if isinstance(nm["lin_num"], list):
                out.update({n: g["lin_num"][j].reshape(1, 1) for j, n in enumerate(nm["lin_num"]) if wide_num[j]})
            else:
                out[nm["lin_num"]] = g["lin_num"].reshape(-1, 1)
    return out


def import_variables(m, arrays, column_names, model="deep_fm", strict=True, numeric_names=None):
    """{TF variable name: ndarray} (e.g. ``dict(np.load("vars.npz"))``) -> engine variables.  Every
    variable the engine has must be present with TF's shape (strict) or keeps its value (not strict);
    returns the list of names that were loaded.  Optimizer slots and Adam row stamps are reset."""
    import torch
    nm = _names_for(m, model, column_names, numeric_names, sharded_ok=True)
    dims, deep_names, wide_num = _subsets(m, numeric_names)
    wide_f = getattr(m, "wide_fields", None)
    loaded = []
    off = m.field_off_host
    rank, world = (0, 1) if m.shard is None else (m.shard.rank, m.shard.world)

    def put_rows(dst, f, a):
        """rows of field f (a: the whole variable) into the engine's table: all of them, or this rank's"""
        if world == 1:
            dst[off[f]:off[f + 1]].copy_(a)
            return
        g = np.arange(int(off[f]), int(off[f + 1]), dtype=np.int64)
        mine = g[g % world == rank]
        if len(mine):
            dst[torch.from_numpy(mine // world).to(m.device)] = a[torch.from_numpy(mine - int(off[f])).to(m.device)]

    def get(name, shape):
        if name not in arrays:
            if strict:
                raise KeyError("variable %r not in the checkpoint dump (has: %s ...)" % (name, sorted(arrays)[:4]))
            return None
        a = np.asarray(arrays[name], dtype=np.float32)
        if a.size != int(np.prod(shape)):
            raise ValueError("variable %r has shape %s, the model needs %s" % (name, a.shape, tuple(shape)))
        loaded.append(name)
        return torch.from_numpy(np.ascontiguousarray(a.reshape(shape))).to(m.device)

    for f in range(m.F):
        v = int(off[f + 1] - off[f])
        if m.table is not None and dims[f]:
            a = get(nm["emb"][f], (v, dims[f]))
            if a is not None:
                if dims[f] < m.E:                                    # a narrower column: its first dims[f] columns
                    a = torch.nn.functional.pad(a, (0, m.E - dims[f]))
                put_rows(m.table, f, a)
        if m.lin_w is not None and (wide_f is None or wide_f[f]):
            a = get(nm["lin_w"][f], (v,))
            if a is not None:
                put_rows(m.lin_w, f, a)
    if m.lin_w is not None:
        a = get(nm["lin_bias"], (1,))
        if a is not None:
            m.dense[m.lin_bias_off:m.lin_bias_off + 1].copy_(a)
    if m.use_dnn:
        for i, (kn, bn) in enumerate(nm["mlp"]):
            _, _, fan, h = m.layers[i]
            k0 = getattr(m, "_k0_rows", None) if i == 0 else None   # (the logical input layer's rows of the stored kernel)
            rows = len(k0) if k0 is not None else (m.D_in if i == 0 else fan)
            a, b = get(kn, (rows, h)), get(bn, (h,))
            if a is not None:
                if i == 0 and m.raw_numeric:
                    inv = torch.from_numpy(np.argsort(kernel0_rows(column_names, deep_names, dims))).to(m.device)
                    a = a[inv]
                if k0 is not None:
                    m.kernel(0)[torch.from_numpy(k0).to(m.device)] = a
                else:
                    m.kernel(i)[:rows].copy_(a)
            if b is not None:
                m.bias(i).copy_(b)
    if m.n_numeric:
        if nm["num_emb"] is not None:
            a = get(nm["num_emb"], (m.n_numeric, m.E))
            if a is not None:
                m._seg(m.dense, m.num_emb_off, (m.n_numeric, m.E)).copy_(a)
        if m.lin_num_off is not None:
            seg = m._seg(m.dense, m.lin_num_off, (m.n_numeric,))
            if isinstance(nm["lin_num"], list):
                for j, n in enumerate(nm["lin_num"]):
                    a = get(n, (1,)) if wide_num[j] else None
                    if a is not None:
                        seg[j:j + 1].copy_(a)
            else:
                a = get(nm["lin_num"], (m.n_numeric,))
                if a is not None:
                    seg.copy_(a)
    m.reset_optimizer_state()
    return loaded
