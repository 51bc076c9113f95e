"""DeepFM / Wide&Deep training engine: the host side of the hot path.

Owns the model variables (PyTorch tensors = device memory only) and drives one train / eval /
predict step as a fixed sequence of libmi355x_rec.so launches on torch's current HIP stream.
It replaces what ``model_fn`` builds and TensorFlow executes in the reference
(``trainers/deep_fm.py:36-125``): the feature-column lookups, FM term, MLP, head and the
optimizer apply.  No autograd, no torch math on the data path: torch allocates buffers and
(for N > 1 GPUs) runs the RCCL collectives.

Variable layout in HBM
  table   [R, E] f32   all embedding tables stacked row-major; field f owns rows
                       [field_off[f], field_off[f+1])  (fields in sorted column-name order,
                       SURVEY A.2) -- one coalesced 4E-byte read per (example, field)
  lin_w   [R]    f32   linear_model weights, same row numbering
  t_s0/t_s1, l_s0/l_s1 optimizer slots shaped like table / lin_w
  last_step [R]  i32   Adam only: step at which a row was last brought up to date
  dense   [P]    f32   every dense variable back to back (16-float aligned segments):
                       kernel_0, bias_0, ..., kernel_logits, bias_logits, linear bias,
                       numeric_embeddings, numeric linear weights; d_s0/d_s1/d_grad mirror it
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import OptHparams, check, ptr

OPT_KINDS = {"Adam": 0, "Adagrad": 1, "Ftrl": 2, "RMSProp": 3, "SGD": 4}


def _align(n, a=16):
    return (n + a - 1) // a * a


class OptimizerSpec:
    """Constructor arguments of tf.train.<name>Optimizer with TF-1.12 defaults
    (reference trainers/model_utils.py:57-66)."""

    def __init__(self, name="Adam", learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=None,
                 decay=0.9, momentum=0.0, lr_power=-0.5, initial_accumulator_value=0.1,
                 l1=0.0, l2=0.0):
        if name not in OPT_KINDS:
            raise KeyError(name)          # the reference's dict lookup raises KeyError too
        self.name = name
        self.kind = OPT_KINDS[name]
        self.lr = float(learning_rate)
        self.beta1, self.beta2 = float(beta1), float(beta2)
        self.epsilon = float(epsilon) if epsilon is not None else (1e-8 if name == "Adam" else 1e-10)
        self.decay, self.momentum = float(decay), float(momentum)
        self.lr_power = float(lr_power)
        self.initial_accumulator_value = float(initial_accumulator_value)
        self.l1, self.l2 = float(l1), float(l2)

    @property
    def slot_init(self):
        """(slot0 fill, slot1 fill) or None when the slot does not exist."""
        return {"Adam": (0.0, 0.0), "Adagrad": (self.initial_accumulator_value, None),
                "Ftrl": (self.initial_accumulator_value, 0.0), "RMSProp": (1.0, 0.0),
                "SGD": (None, None)}[self.name]

    def hparams(self, lr_t=0.0):
        return OptHparams(self.kind, self.lr, self.beta1, self.beta2, self.epsilon, float(lr_t),
                          self.decay, self.momentum, self.lr_power, self.l1, self.l2)


class AdamSchedule:
    """beta1_power / beta2_power bookkeeping in fp32 exactly as TF keeps them, and the
    device table lr_t[s] that mi_sparse_catchup replays."""

    def __init__(self, spec, device, capacity=1 << 16):
        self.spec = spec
        self.device = device
        self.b1 = np.float32(spec.beta1)
        self.b2 = np.float32(spec.beta2)
        self.b1p = np.float32(spec.beta1)   # power that step 1 will see
        self.b2p = np.float32(spec.beta2)
        self.host = np.zeros(1, np.float32)  # index 0 unused: steps are 1-based
        self.table = None
        self._extend(capacity)

    def _extend(self, capacity):
        one = np.float32(1)
        lr = np.float32(self.spec.lr)
        vals = []
        b1p, b2p = self.b1p, self.b2p
        for _ in range(len(self.host), capacity + 1):
            vals.append(lr * np.sqrt(one - b2p) / (one - b1p))
            b1p = np.float32(b1p * self.b1)
            b2p = np.float32(b2p * self.b2)
        self.b1p, self.b2p = b1p, b2p
        self.host = np.concatenate([self.host, np.asarray(vals, np.float32)])
        self.table = torch.from_numpy(self.host.copy()).to(self.device)

    def lr_t(self, step):
        if step >= len(self.host):
            self._extend(2 * len(self.host))
        return float(self.host[step])


class DeepFM:
    """model_fn-shaped model (reference trainers/deep_fm.py:11-125).

    vocab_sizes: rows per categorical field, already in sorted column-name order.
    reduction: "mean" (contrib head, DeepFM) or "sum" (canned estimators), SURVEY A.5.
    linear_optimizer: if given, the wide part (lin_w + linear bias [+ numeric linear weights]) uses
    it and everything else uses ``optimizer`` (DNNLinearCombinedClassifier)."""

    def __init__(self, vocab_sizes, n_numeric=0, embedding_size=4, hidden_units=(16, 16),
                 use_linear=True, use_mf=True, use_dnn=True, dropout=0.0, optimizer=None,
                 linear_optimizer=None, reduction="mean", device="cuda", seed=0, shard=None):
        if len(vocab_sizes) + n_numeric == 0:
            raise ValueError("At least 1 feature column of categorical_columns or numeric_columns "
                             "must be specified.")            # deep_fm.py:31-32
        if not (use_linear or use_mf or use_dnn):
            raise ValueError("At least 1 of linear, mf or dnn component must be used.")  # :33-34
        if len(vocab_sizes) == 0:
            raise NotImplementedError("numeric-only models are not supported by the HIP path")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.vocab_sizes = [int(v) for v in vocab_sizes]
        self.F = len(self.vocab_sizes)
        self.n_numeric = int(n_numeric)
        self.E = int(embedding_size)
        self.hidden = [int(h) for h in hidden_units] if use_dnn else []
        self.use_linear, self.use_mf, self.use_dnn = bool(use_linear), bool(use_mf), bool(use_dnn)
        self.use_emb = self.use_mf or self.use_dnn
        self.dropout = float(dropout)
        self.reduction = reduction
        self.opt = optimizer or OptimizerSpec()
        self.lin_opt = linear_optimizer
        self.seed = int(seed)
        self.shard = shard                      # parallel.RowShard or None
        self.step = 0
        if self.use_emb and (self.E % 4 or not 4 <= self.E <= 256):
            raise ValueError("embedding_size must be a multiple of 4 in [4, 256] on the HIP path")

        off = np.zeros(self.F + 1, np.int64)
        off[1:] = np.cumsum(self.vocab_sizes)
        self.field_off_host = off
        self.R = int(off[-1])
        if self.R >= 2 ** 31:
            raise ValueError("total rows must fit int32")
        dev = self.device
        self.field_off = torch.from_numpy(off[:-1].copy()).to(dev)
        self.R_local = self.R if shard is None else shard.local_rows(self.R)

        f32 = dict(dtype=torch.float32, device=dev)
        self.table = torch.zeros(self.R_local, self.E, **f32) if self.use_emb else None
        self.lin_w = torch.zeros(self.R_local, **f32) if self.use_linear else None
        sparse_lin_opt = self.lin_opt or self.opt
        self.t_s0, self.t_s1 = self._slots(self.table, self.opt)
        self.l_s0, self.l_s1 = self._slots(self.lin_w, sparse_lin_opt)
        self.adam_rows = self.opt.name == "Adam" or sparse_lin_opt.name == "Adam"
        self.last_step = torch.zeros(self.R_local, dtype=torch.int32, device=dev) if self.adam_rows else None
        self.sched = AdamSchedule(self.opt if self.opt.name == "Adam" else sparse_lin_opt, dev) \
            if (self.opt.name == "Adam" or sparse_lin_opt.name == "Adam") else None

        # dense variables: one flat buffer
        self.D = (self.F + self.n_numeric) * self.E if self.use_emb else 0
        self.layers = []                         # (kernel_off, bias_off, fan_in, fan_out)
        segs = []
        o = 0
        if self.use_dnn:
            fan = self.D
            for h in self.hidden + [1]:
                k_off = o; o = _align(o + fan * h)
                b_off = o; o = _align(o + h)
                self.layers.append((k_off, b_off, fan, h))
                fan = h
        self.dnn_end = o
        self.lin_bias_off = o; o = _align(o + 1)
        self.num_emb_off = self.lin_num_off = None
        if self.n_numeric:
            self.num_emb_off = o; o = _align(o + self.n_numeric * self.E)
            self.lin_num_off = o; o = _align(o + self.n_numeric)
        self.P = o
        self.dense = torch.zeros(self.P, **f32)
        self.d_grad = torch.zeros(self.P, **f32)
        self.d_s0, self.d_s1 = self._slots(self.dense, self.opt)
        if self.lin_opt is not None:
            # wide-part dense variables (linear bias, numeric linear weights) follow linear_optimizer
            self.dl_s0, self.dl_s1 = self._slots(self.dense, self.lin_opt)
        self._ws = {}
        self.timers = None

    # ------------------------------------------------------------------ variables
    def _slots(self, like, spec):
        if like is None:
            return None, None
        a, b = spec.slot_init
        s0 = torch.full_like(like, a) if a is not None else None
        s1 = torch.full_like(like, b) if b is not None else None
        return s0, s1

    def _seg(self, buf, off, shape):
        n = int(np.prod(shape))
        return buf[off:off + n].view(*shape)

    def kernel(self, i, buf=None):
        k_off, _, fan, h = self.layers[i]
        return self._seg(self.dense if buf is None else buf, k_off, (fan, h))

    def bias(self, i, buf=None):
        _, b_off, _, h = self.layers[i]
        return self._seg(self.dense if buf is None else buf, b_off, (h,))

    def init_variables(self, generator=None, lin_scale=0.0):
        """TF initialisers: truncated_normal(0, 1/sqrt(E)) embeddings, zero linear
        weights / biases, glorot-uniform kernels.  torch's generator, not TF's Philox stream."""
        g = generator
        if self.table is not None:
            torch.nn.init.trunc_normal_(self.table, 0.0, 1.0 / math.sqrt(self.E), -2.0 / math.sqrt(self.E),
                                        2.0 / math.sqrt(self.E), generator=g)
        if self.lin_w is not None and lin_scale:
            self.lin_w.normal_(0.0, lin_scale, generator=g)
        for i, (_, _, fan, h) in enumerate(self.layers):
            lim = math.sqrt(6.0 / (fan + h))
            self.kernel(i).uniform_(-lim, lim, generator=g)
        if self.n_numeric:
            lim = math.sqrt(6.0 / (self.n_numeric + self.E))
            self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E)).uniform_(-lim, lim, generator=g)

    def load_oracle_params(self, p):
        """Copy an ``oracle.deepfm.Params`` (numpy) into the device buffers (tests / smoke)."""
        assert self.shard is None
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        if self.table is not None:
            self.table.copy_(t(np.concatenate(p.emb, 0)))
        if self.lin_w is not None:
            self.lin_w.copy_(t(np.concatenate(p.lin_w, 0)))
        for i in range(len(self.layers)):
            self.kernel(i).copy_(t(p.mlp[i][0]))
            self.bias(i).copy_(t(p.mlp[i][1]))
        self.dense[self.lin_bias_off] = float(p.lin_bias[0])
        if self.n_numeric:
            self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E)).copy_(t(p.num_emb))
            self._seg(self.dense, self.lin_num_off, (self.n_numeric,)).copy_(t(p.lin_num))

    def export_numpy(self):
        """Variables as numpy arrays in the oracle's structure (after bringing Adam rows up to date)."""
        self.finalize_rows()
        off = self.field_off_host
        sp = lambda a: [a[off[f]:off[f + 1]].cpu().numpy() for f in range(self.F)] if a is not None else None
        out = {"emb": sp(self.table), "lin_w": sp(self.lin_w),
               "mlp": [(self.kernel(i).cpu().numpy(), self.bias(i).cpu().numpy()) for i in range(len(self.layers))],
               "lin_bias": self.dense[self.lin_bias_off:self.lin_bias_off + 1].cpu().numpy()}
        if self.n_numeric:
            out["num_emb"] = self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E)).cpu().numpy()
            out["lin_num"] = self._seg(self.dense, self.lin_num_off, (self.n_numeric,)).cpu().numpy()
        return out

    # ------------------------------------------------------------------ helpers
    def _buf(self, name, shape, dtype=torch.float32):
        n = int(np.prod(shape))
        cur = self._ws.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            cur = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._ws[name] = cur
        return cur[:n].view(*shape)

    def _bytes(self, name, nbytes):
        n = (int(nbytes) + 255) // 256 * 256 + 256
        cur = self._ws.get(name)
        if cur is None or cur.numel() < n:
            cur = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws[name] = cur
        return cur

    def _run(self, fn, *args):
        """Launch one C-ABI entry on the current stream; with self.timers set, bracket it with HIP
        events on that stream (bench.py reads per-kernel durations from them)."""
        if self.timers is None:
            rc = fn(*args)
        else:
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            self.timers.setdefault(fn.__name__, []).append((s, e))
        check(rc, fn.__name__)

    def _layer_seed(self, layer):
        return (self.seed * 0x9E3779B97F4A7C15 + (self.step + 1) * 1000003 + layer * 7919) & (2 ** 64 - 1)

    # ------------------------------------------------------------------ forward
    def _forward(self, ids, x_num, train, st):
        """ids [B,F] int32 device; returns logits components, caches activations for backward."""
        L = self.lib
        B = ids.shape[0]
        c = {"B": B}
        lin = fm = dnn = None
        concat = sumv = None
        ld = self.D
        if self.use_emb:
            concat = self._buf("concat", (B, ld))
            sumv = self._buf("sumv", (B, self.E)) if self.use_mf else None
            fm = self._buf("fm", (B,)) if self.use_mf else None
        lin = self._buf("lin", (B,)) if self.use_linear else None
        self._run(L.mi_embed_fm_linear_fwd, ptr(self.table), ptr(self.lin_w), ptr(self.field_off), ptr(ids),
                                       B, self.F, self.E, ptr(concat), ld, ptr(sumv), ptr(fm), ptr(lin), st)
        if self.n_numeric:
            V = self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E))
            wn = self._seg(self.dense, self.lin_num_off, (self.n_numeric,)) if self.use_linear else None
            if self.use_emb:
                self._run(L.mi_numeric_embed_fwd, ptr(x_num), ptr(V), ptr(wn), B, self.n_numeric, self.E,
                                             ptr(concat), ld, self.F * self.E, ptr(sumv), ptr(fm), ptr(lin), st)
            else:
                raise NotImplementedError("numeric columns need the embedding path (use_mf or use_dnn)")
        acts = []
        if self.use_dnn:
            x, ldx = concat, ld
            keep = 1.0 - self.dropout if (train and self.dropout > 0) else 1.0
            nh = len(self.layers) - 1
            for i, (k_off, b_off, fan, h) in enumerate(self.layers):
                last = i == nh
                y = self._buf("act%d" % i, (B, h))
                self._run(L.mi_dense_fwd, ptr(x), ldx, ptr(self.kernel(i)), ptr(self.bias(i)), ptr(y), h, B, h, fan,
                                     0 if last else 1, 1.0 if last else keep, self._layer_seed(i), st)
                acts.append(y)
                x, ldx = y, h
            dnn = acts[-1].view(B)
            c["keep"] = keep
        c.update(concat=concat, sumv=sumv, fm=fm, lin=lin, dnn=dnn, acts=acts, ids=ids, x_num=x_num)
        return c

    def _head(self, c, labels, st, want_grad, global_batch=None):
        L = self.lib
        B = c["B"]
        logits = self._buf("logits", (B,))
        loss = self._buf("loss", (1,)) if labels is not None else None
        dlogit = self._buf("dlogit", (B,)) if want_grad else None
        n = global_batch if global_batch is not None else B
        scale = np.float32(1.0 / n) if self.reduction == "mean" else np.float32(1.0)
        ws = self._bytes("head_ws", L.mi_head_workspace_bytes(B))
        lb = self.dense[self.lin_bias_off:] if self.use_linear else None
        dsum = self.d_grad[self.lin_bias_off:] if (want_grad and self.use_linear) else None
        self._run(L.mi_sigmoid_ce_head, ptr(c["lin"]), ptr(lb), ptr(c["fm"]), ptr(c["dnn"]), ptr(labels), B,
                  float(scale), ptr(logits), ptr(loss), ptr(dlogit), ptr(dsum), ptr(ws), ws.numel(), st)
        return logits, loss, dlogit

    # ------------------------------------------------------------------ public steps
    def _prep(self, ids, labels, x_num):
        if ids.dtype != torch.int32 or not ids.is_contiguous() or ids.dim() != 2 or ids.shape[1] != self.F:
            raise ValueError("ids must be a contiguous int32 [B, %d] tensor" % self.F)
        if ids.device != self.device and ids.device.type != self.device.type:
            raise ValueError("ids must live on %s" % self.device)
        if labels is not None and (labels.dtype != torch.uint8 or labels.shape != (ids.shape[0],)):
            raise ValueError("labels must be uint8 [B]")
        if self.n_numeric:
            if x_num is None or x_num.shape != (ids.shape[0], self.n_numeric) or x_num.dtype != torch.float32:
                raise ValueError("x_num must be float32 [B, %d]" % self.n_numeric)
            if not x_num.is_contiguous():
                raise ValueError("x_num must be contiguous")
        elif x_num is not None:
            raise ValueError("model has no numeric columns")

    def predict_logits(self, ids, x_num=None):
        """PREDICT / EVAL forward (no dropout).  Returns logits [B] (device)."""
        self._prep(ids, None, x_num)
        self.finalize_rows()
        st = _lib.cur_stream()
        c = self._forward(ids, x_num, False, st)
        logits, _, _ = self._head(c, None, st, False)
        return logits

    def loss(self, ids, labels, x_num=None):
        """EVAL forward: (loss [1], logits [B]) without touching any variable."""
        self._prep(ids, labels, x_num)
        self.finalize_rows()
        st = _lib.cur_stream()
        c = self._forward(ids, x_num, False, st)
        logits, loss, _ = self._head(c, labels, st, False)
        return loss, logits

    def finalize_rows(self):
        """Bring every Adam row up to date (all-rows mi_sparse_catchup).  No-op when nothing is stale."""
        if not self.adam_rows or self.step == 0 or getattr(self, "_final_step", -1) == self.step:
            return
        st = _lib.cur_stream()
        self._catchup(None, None, self.R_local, st)
        self._final_step = self.step

    def _catchup(self, uniq, num_uniq, n_max, st):
        s = self.sched.spec
        t_adam = self.opt.name == "Adam" and self.table is not None
        l_adam = (self.lin_opt or self.opt).name == "Adam" and self.lin_w is not None
        if not (t_adam or l_adam):
            return
        self.sched.lr_t(self.step)  # make sure the table covers step
        self._run(self.lib.mi_sparse_catchup, ptr(self.table if t_adam else None), ptr(self.t_s0 if t_adam else None),
                                         ptr(self.t_s1 if t_adam else None), ptr(self.lin_w if l_adam else None),
                                         ptr(self.l_s0 if l_adam else None), ptr(self.l_s1 if l_adam else None),
                                         ptr(self.last_step), ptr(uniq), ptr(num_uniq), n_max, self.E, self.step,
                                         ptr(self.sched.table), s.beta1, s.beta2, s.epsilon, st)

    def train_step(self, ids, labels, x_num=None):
        """One optimizer.minimize(loss): returns (loss [1], logits [B]) device tensors, no host sync."""
        self._prep(ids, labels, x_num)
        if self.shard is not None:
            from . import parallel
            return parallel.sharded_train_step(self, ids, labels, x_num)
        L = self.lib
        st = _lib.cur_stream()
        B = ids.shape[0]
        n = B * self.F
        i32 = torch.int32
        # (1) which rows does this batch touch: sort + unique (TF: unique/unsorted_segment_sum)
        rows = self._buf("rows", (n,), i32)
        self._run(L.mi_global_rows, ptr(ids), ptr(self.field_off), B, self.F, ptr(rows), st)
        sorted_entry = self._buf("sorted_entry", (n,), i32)
        uniq = self._buf("uniq", (n,), i32)
        seg = self._buf("seg", (n + 1,), i32)
        num_uniq = self._buf("num_uniq", (1,), i32)
        ws = self._bytes("sort_ws", L.mi_sort_unique_workspace_bytes(n))
        self._run(L.mi_sort_unique_rows, ptr(rows), n, self.R, ptr(sorted_entry), ptr(uniq), ptr(seg), ptr(num_uniq),
                                    ptr(ws), ws.numel(), st)
        # (2) TF Adam moved these rows on every step they sat out: replay that now
        if self.adam_rows and self.step > 0:
            self._catchup(uniq, num_uniq, n, st)
        # (3) forward + head
        c = self._forward(ids, x_num, True, st)
        logits, loss, dlogit = self._head(c, labels, st, True)
        # (4) backward through the MLP
        d_concat = self._backward_dense(c, dlogit, st)
        # (5) per-entry row gradients, then the sparse apply on unique rows
        d_rows = self._buf("d_rows", (n, self.E)) if self.use_emb else None
        d_lin = self._buf("d_lin", (n,)) if self.use_linear else None
        self._run(L.mi_embed_fm_linear_bwd, ptr(d_concat), self.D, ptr(c["concat"]), self.D, ptr(c["sumv"]),
                                       ptr(dlogit if self.use_mf else None),
                                       ptr(dlogit if self.use_linear else None), None, B, self.F, self.E,
                                       ptr(d_rows), ptr(d_lin), st)
        self._apply(uniq, seg, sorted_entry, num_uniq, n, d_rows, d_lin, st)
        return loss, logits

    def _backward_dense(self, c, dlogit, st):
        """Fills self.d_grad (dense gradients) and returns d_concat [B, D] (or None)."""
        L = self.lib
        B = c["B"]
        d_concat = None
        if self.use_dnn:
            nh = len(self.layers) - 1
            keep = c["keep"]
            dy, lddy = dlogit, 1
            wsz = max(L.mi_dense_bwd_weight_workspace_bytes(B, h, fan) for (_, _, fan, h) in self.layers)
            ws = self._bytes("wgrad_ws", wsz)
            for i in range(nh, -1, -1):
                k_off, b_off, fan, h = self.layers[i]
                x = c["acts"][i - 1] if i else c["concat"]
                ldx = self.layers[i - 1][3] if i else self.D
                self._run(L.mi_dense_bwd_weight, ptr(x), ldx, ptr(dy), lddy, ptr(self.kernel(i, self.d_grad)),
                                            ptr(self.bias(i, self.d_grad)), B, h, fan, ptr(ws), ws.numel(), st)
                dx = self._buf("dact%d" % i, (B, fan))
                self._run(L.mi_dense_bwd_data, ptr(dy), lddy, ptr(self.kernel(i)), ptr(x if i else None), ldx,
                                          ptr(dx), fan, B, h, fan, keep if i else 1.0, st)
                dy, lddy = dx, fan
            d_concat = dy
        # d loss / d linear bias = sum_b dlogit was written into d_grad by the head kernel
        if self.n_numeric:
            ws = self._bytes("num_ws", L.mi_numeric_embed_bwd_workspace_bytes(B, self.n_numeric, self.E))
            self._run(L.mi_numeric_embed_bwd, ptr(c["x_num"]), ptr(d_concat), self.D, ptr(c["concat"]), self.D,
                                         self.F * self.E, ptr(c["sumv"]), ptr(dlogit if self.use_mf else None),
                                         ptr(dlogit if self.use_linear else None), B, self.n_numeric, self.E,
                                         ptr(self.d_grad[self.num_emb_off:]),
                                         ptr(self.d_grad[self.lin_num_off:] if self.use_linear else None),
                                         ptr(ws), ws.numel(), st)
        return d_concat

    def _apply(self, uniq, seg, sorted_entry, num_uniq, n_max, d_rows, d_lin, st):
        """apply_gradients: dense Apply*, sparse apply on the unique rows, step += 1."""
        L = self.lib
        step = self.step + 1
        lr_t = self.sched.lr_t(step) if self.sched else 0.0
        hp = self.opt.hparams(lr_t)
        if self.lin_opt is None:
            if self.P:
                self._run(L.mi_dense_apply, ptr(self.dense), ptr(self.d_s0), ptr(self.d_s1), ptr(self.d_grad), self.P,
                                       C.byref(hp), st)
            sparse_hp = [(True, True, hp)]
        else:
            lhp = self.lin_opt.hparams(lr_t)
            if self.dnn_end:
                self._run(L.mi_dense_apply, ptr(self.dense), ptr(self.d_s0), ptr(self.d_s1), ptr(self.d_grad),
                                       self.dnn_end, C.byref(hp), st)
            o = self.dnn_end
            sl = lambda t: t[o:] if t is not None else None
            self._run(L.mi_dense_apply, ptr(self.dense[o:]), ptr(sl(self.dl_s0)), ptr(sl(self.dl_s1)),
                                   ptr(self.d_grad[o:]), self.P - o, C.byref(lhp), st)
            sparse_hp = [(True, False, hp), (False, True, lhp)]
        for do_table, do_lin, h in sparse_hp:
            tb = self.table if (do_table and self.use_emb) else None
            lw = self.lin_w if (do_lin and self.use_linear) else None
            if tb is None and lw is None:
                continue
            self._run(L.mi_sparse_apply, ptr(tb), ptr(self.t_s0 if tb is not None else None),
                                    ptr(self.t_s1 if tb is not None else None), ptr(lw),
                                    ptr(self.l_s0 if lw is not None else None),
                                    ptr(self.l_s1 if lw is not None else None), ptr(self.last_step),
                                    ptr(uniq), ptr(seg), ptr(sorted_entry), ptr(num_uniq), n_max,
                                    ptr(d_rows if tb is not None else None), ptr(d_lin if lw is not None else None),
                                    self.E, step, C.byref(h), st)
        self.step = step

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self):
        self.finalize_rows()
        sd = {"step": self.step, "dense": self.dense, "d_s0": self.d_s0, "d_s1": self.d_s1,
              "table": self.table, "lin_w": self.lin_w, "t_s0": self.t_s0, "t_s1": self.t_s1,
              "l_s0": self.l_s0, "l_s1": self.l_s1, "last_step": self.last_step}
        if self.lin_opt is not None:
            sd.update(dl_s0=self.dl_s0, dl_s1=self.dl_s1)
        return {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items() if v is not None}

    def load_state_dict(self, sd):
        self.step = int(sd["step"])
        for k, v in sd.items():
            if k == "step":
                continue
            getattr(self, k).copy_(v.to(self.device))
        self._final_step = self.step
