"""DeepFM / Wide&Deep training engine: the host side of the hot path.

Owns the model variables (PyTorch tensors = device memory only) and drives one train / eval /
predict step as a fixed sequence of libmi355x_rec.so launches on torch's current HIP stream.
It replaces what ``model_fn`` builds and TensorFlow executes in the reference
(``trainers/deep_fm.py:36-125``): the feature-column lookups, FM term, MLP, head and the
optimizer apply.  No autograd, no torch math on the data path: torch allocates buffers and
(for N > 1 GPUs, ``parallel.py``) runs the RCCL collectives.

Variable layout in HBM
  t_rec   [R, 3E] f32  one record [w | slot0 | slot1] per embedding row (fewer slots: narrower), all tables stacked;
                       field f owns rows [field_off[f], field_off[f+1])  (fields in sorted column-name order,
                       SURVEY A.2).  table, t_s0, t_s1 are strided [R, E] views; kernels take the record stride ts
  table   [R, E] view  the weights: one coalesced 4E-byte read per (example, field)
  lin_state [R, 4] f32 the wide part's per-row record {weight, slot0, slot1, Adam stamp}: lin_w, l_s0, l_s1 and
                       last_step are strided views of it (one memory sector per row)
  t_s0/t_s1            optimizer slots: views of t_rec shaped like table
  last_step [R]  i32   Adam only: step at which a row was last brought up to date (a view of lin_state, or a
                       plain array when there is no wide part)
  dense   [P]    f32   every dense variable back to back (16-float aligned segments): first TF's
                       "dnn" scope — kernel_0, bias_0, ..., kernel_logits, bias_logits,
                       numeric_embeddings — then, from wide_off on, the dense part of its "linear" scope —
                       linear bias, numeric linear weights; d_s0/d_s1/d_grad mirror it
With a RowShard (N > 1 GPUs) table / lin_w / slots / last_step hold only the rows r with
r % world == rank, stored at r // world; the dense buffer is replicated.
"""
import ctypes as C
import json
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import OptHparams, check, ptr

OPT_KINDS = {"Adam": 0, "Adagrad": 1, "Ftrl": 2, "RMSProp": 3, "SGD": 4}


def _align(n, a=16):
    return (n + a - 1) // a * a


class HipKernels:
    """The device entry points of libmi355x_rec.so, called with torch tensors.

    ``k.mi_xxx(a, b, ...)`` turns tensors into device pointers, structs into byref, appends torch's
    current HIP stream and raises MiError on a non-zero status.  With ``timers`` set, each launch is
    bracketed by HIP events on that stream (bench.py reads per-entry durations from them).

    Test seam only: the CPU (gloo) tests of the multi-rank exchange plumbing pass the engine an
    object with the same method names implemented in numpy (tests/cpu_kernels.py).  The shipped
    path always constructs this class, and this class cannot be constructed without the library.
    """

    supports_planes = True        # the pre-split 16-bit operand path of the MLP (csrc/gemm_pl.hip)

    def __init__(self):
        self.lib = _lib.load()
        self.timers = None
        self.timer_only = None        # a set of timer keys: bracket only these launches (an event pair costs ~10 us of stream time)

    def query(self, name, *args):
        """Host-side queries (``*_workspace_bytes``): no stream, returns the value."""
        return getattr(self.lib, name)(*args)

    def tagged(self, name, tag):
        """The entry `name`, timed under `name + tag` (two uses of one entry that bench.py prices apart)."""
        key = name + tag
        if key not in self.__dict__:
            self.__dict__[key] = self._bind(name, key)
        return self.__dict__[key]

    def __getattr__(self, name):
        if not name.startswith("mi_"):
            raise AttributeError(name)
        call = self.__dict__[name] = self._bind(name, name)
        return call

    def _bind(self, name, key):
        fn = getattr(self.lib, name)

        def call(*args):
            a = [ptr(x) if isinstance(x, torch.Tensor) else (C.byref(x) if isinstance(x, C.Structure) else x)
                 for x in args]
            a.append(_lib.cur_stream())
            if self.timers is None or (self.timer_only is not None and key not in self.timer_only):
                rc = fn(*a)
            else:
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                rc = fn(*a)
                e.record()
                self.timers.setdefault(key, []).append((s, e))
            check(rc, name)

        return call


class PlaneBuf:
    """Device memory of one mi_planes_t matrix (k-block major: [K/16 blocks][rows][64 B]) + one int32
    exponent per row."""

    def __init__(self, rows, K, device):
        self.rows, self.K = int(rows), int(K)
        self.blk_stride = 64 * max(self.rows, 1)
        self.data = torch.empty(_align(self.K, 16) // 16 * self.blk_stride, dtype=torch.uint8, device=device)
        self.exp = torch.empty(max(self.rows, 1), dtype=torch.int32, device=device)
        self.struct = _lib.Planes(self.data.data_ptr(), self.exp.data_ptr(), self.blk_stride)


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
        self.gen = 0                         # bumped whenever `table` moves to new device memory (captured graphs hold its address)
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
        self.gen += 1

    def lr_t(self, step):
        while step >= len(self.host):                     # also after restoring a checkpoint far into a run
            self._extend(max(2 * len(self.host), step + 1))
        return float(self.host[step])



_SIDE_STREAMS = {}        # (device index, priority) -> the side streams found so far, in the order engines ask for them
_SPIN = {}                # device index -> spin cycles for ~0.2 ms


def _streams_overlap(a, b, cyc):
    """Do kernels enqueued on streams a and b run side by side?  A long spin on a, a short one on b: on ONE hardware queue the
    short one ends after the long one."""
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        e0.record()
        torch.cuda._sleep(cyc)
        e1.record()
    with torch.cuda.stream(b):
        torch.cuda._sleep(max(cyc // 8, 1))
        e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e2) < 0.6 * e0.elapsed_time(e1)


def _tested_side_stream(device, prio, k):
    key = (device.index if device.index is not None else torch.cuda.current_device(), prio)
    found = _SIDE_STREAMS.setdefault(key, [])
    if k < len(found):
        return found[k]
    with torch.cuda.device(key[0]):                             # (the spin kernels and the synchronisations: on THAT device)
        return _find_side_stream(device, prio, k, key, found)


def _find_side_stream(device, prio, k, key, found):
    if not hasattr(torch.cuda, "_sleep") or torch.cuda.is_current_stream_capturing():
        found.append(torch.cuda.Stream(device=device, priority=prio))
        return found[-1]
    cyc = _SPIN.get(key[0])
    if cyc is None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)                                 # (the first launch of the spin kernel is not timed)
        torch.cuda.synchronize()
        a.record(); torch.cuda._sleep(1_000_000); b.record(); torch.cuda.synchronize()
        cyc = _SPIN[key[0]] = max(int(1_000_000 * 0.2 / max(a.elapsed_time(b), 1e-3)), 1000)
    main = torch.cuda.current_stream(device)
    others = [s for (d, _), ss in _SIDE_STREAMS.items() if d == key[0] for s in ss]
    seen, fallback = set(), None
    while len(found) <= k:
        pick = None
        for _ in range(32):                                     # once around the pool
            s = torch.cuda.Stream(device=device, priority=prio)
            if s.cuda_stream in seen or any(s.cuda_stream == o.cuda_stream for o in others + [main]):
                continue
            seen.add(s.cuda_stream)
            if not _streams_overlap(main, s, cyc):
                continue
            if fallback is None:
                fallback = s                                    # beside the step's stream at least
            if all(_streams_overlap(o, s, cyc) for o in others):
                pick = s
                break
        pick = pick or fallback or torch.cuda.Stream(device=device, priority=prio)
        found.append(pick)
        others.append(pick)
        fallback = None
    return found[k]

class DeepFM:
    """model_fn-shaped model (reference trainers/deep_fm.py:11-125).

    vocab_sizes: rows per categorical field, already in sorted column-name order.
    reduction: "mean" (contrib head, DeepFM) or "sum" (canned estimators), SURVEY A.5.
    linear_optimizer: if given, the wide part (lin_w + linear bias [+ numeric linear weights]) uses
    it and everything else uses ``optimizer`` (DNNLinearCombinedClassifier).
    numeric: what a numeric column feeds the deep part with — "embed": DeepFM's numeric_embeddings
    (deep_fm.py:62-73, x[b,j] * V[j,:], also seen by the FM term); "raw": the value itself, appended
    to the concat after the embedding columns, as the canned estimators' input_layer does (TF orders
    the concat by column name; here numeric columns follow the categorical block, which permutes
    kernel_0's rows only — tf_names maps them).
    shard: parallel.RowShard for N > 1 GPUs (row-sharded tables, data-parallel MLP).
    field_dims / wide_fields (the canned DNNLinearCombinedClassifier with dnn_feature_columns != linear_feature_columns,
    linear_deep.py:32-39 / SURVEY A.7; one GPU or row-sharded over N): per-field embedding dimensions (<= embedding_size; 0 = the column is
    not in the deep part) and per-field flags "the column has a linear weight".  The fused [R, E] table keeps one E: a
    narrower column uses the first field_dims[f] of them and its other columns — with the matching rows of kernel_0 — are
    zero and stay zero under every optimizer here (their gradients are products with those zeros), which is the smaller
    embedding exactly; the wide part runs on the wide columns' ids only.
    catchup: how the steps a row sat out under TF Adam's dense-equivalent sparse update are replayed when
    the row is next read — "bounded" (the default: what bench.py's `value` is timed in and what the trainers' CLIs run; inside
    the north star's 1e-5 on logits / loss) or "exact" (--catchup exact: TensorFlow's bits, ~8 % slower at
    config 3).  "exact": TF's fp32 op sequence, the sweep's bits; "bounded": the same m chain and numerators
    with sqrt(v_j) ~ sqrtf(v_0) beta2^(j/2) and a 1-ulp reciprocal: every variable within 3 ulp + 2e-6 of the movement the
    replay covers (98.7 % of them within 1e-7 relative of the sweep, 96.7 % bit-identical; include/mi355x_rec.h,
    MI_CATCHUP_BOUNDED), a third of the instructions."""

    def __init__(self, vocab_sizes, n_numeric=0, embedding_size=4, hidden_units=(16, 16),
                 use_linear=True, use_mf=True, use_dnn=True, dropout=0.0, optimizer=None,
                 linear_optimizer=None, reduction="mean", device="cuda", seed=0, shard=None,
                 gemm="f16x2", numeric="embed", activation="relu", catchup="bounded", field_dims=None, wide_fields=None,
                 deep_numeric=None, wide_numeric=None, _kernels=None):
        if len(vocab_sizes) + n_numeric == 0:
            raise ValueError("At least 1 feature column of categorical_columns or numeric_columns "
                             "must be specified.")            # deep_fm.py:31-32
        if not (use_linear or use_mf or use_dnn):
            raise ValueError("At least 1 of linear, mf or dnn component must be used.")  # :33-34
        # params["activation"] of model_fn (deep_fm.py:22; the reference passes a TF callable, default tf.nn.relu)
        acts = {"relu": 1, None: 0, "identity": 0, "linear": 0, "sigmoid": 2, "tanh": 3}
        name = activation if (activation is None or isinstance(activation, str)) else getattr(activation, "__name__", None)
        if name not in acts:
            raise NotImplementedError("activation %r: the GEMM epilogues implement relu, sigmoid, tanh and identity" % (activation,))
        self.act = acts[name]
        if numeric not in ("embed", "raw"):
            raise ValueError("numeric must be 'embed' or 'raw'")
        if numeric == "raw" and use_mf:
            raise ValueError("raw numeric columns belong to the canned estimators, which have no FM term")
        if catchup not in ("exact", "bounded"):
            raise ValueError("catchup must be 'exact' or 'bounded'")
        self.field_dims = None if field_dims is None else [int(d) for d in field_dims]
        self.wide_fields = None if (wide_fields is None or all(wide_fields)) else [bool(w) for w in wide_fields]
        if self.field_dims is not None and all(d == int(embedding_size) for d in self.field_dims):
            self.field_dims = None
        if self.field_dims is not None or self.wide_fields is not None:
            if use_mf or (n_numeric and numeric == "embed"):
                raise ValueError("field_dims / wide_fields belong to the canned estimators (no FM term, raw numeric columns)")
            for v in (self.field_dims, self.wide_fields):
                if v is not None and len(v) != len(vocab_sizes):
                    raise ValueError("field_dims / wide_fields need one entry per categorical field")
            if self.field_dims is not None and any(d < 0 or d > int(embedding_size) for d in self.field_dims):
                raise ValueError("field_dims must lie in [0, embedding_size]")
        flags = lambda v: None if (v is None or all(v)) else [bool(x) for x in v]
        self.deep_numeric, self.wide_numeric = flags(deep_numeric), flags(wide_numeric)
        if self.deep_numeric is not None or self.wide_numeric is not None:
            if numeric != "raw":
                raise ValueError("deep_numeric / wide_numeric belong to the canned estimators (raw numeric columns)")
            for v in (self.deep_numeric, self.wide_numeric):
                if v is not None and len(v) != int(n_numeric):
                    raise ValueError("deep_numeric / wide_numeric need one entry per numeric column")
        self.catchup = catchup
        self.k = _kernels if _kernels is not None else HipKernels()
        self.device = torch.device(device)
        self.vocab_sizes = [int(v) for v in vocab_sizes]
        self.F = len(self.vocab_sizes)
        self.max_vocab = max(self.vocab_sizes) if self.vocab_sizes else 1
        self.n_numeric = int(n_numeric)
        self.numeric = numeric if self.n_numeric else "embed"
        self.raw_numeric = self.numeric == "raw"
        self.E = int(embedding_size)
        self.hidden = [int(h) for h in hidden_units] if use_dnn else []
        self.use_linear, self.use_mf, self.use_dnn = bool(use_linear), bool(use_mf), bool(use_dnn)
        self.use_emb = self.use_mf or self.use_dnn
        self.dropout = float(dropout)
        self.reduction = reduction
        self.opt = optimizer or OptimizerSpec()
        self.lin_opt = linear_optimizer
        self.seed = int(seed)
        self.shard = shard
        self.step = 0
        if self.use_emb and (self.E % 4 or not 4 <= self.E <= 256):
            raise ValueError("embedding_size must be a multiple of 4 in [4, 256] on the HIP path")
        if self.n_numeric and not self.raw_numeric and not self.use_emb:
            raise NotImplementedError("numeric embeddings need the embedding path (use_mf or use_dnn)")

        off = np.zeros(self.F + 1, np.int64)
        off[1:] = np.cumsum(self.vocab_sizes)
        self.field_off_host = off
        self.R = int(off[-1])
        if self.R >= 2 ** 31:
            raise ValueError("total rows must fit int32")
        dev = self.device
        self.field_off = torch.from_numpy(off[:-1].copy()).to(dev)
        self.R_local = self.R if shard is None else shard.local_rows(self.R)
        self.wide_idx = self.wide_off_t = None                # the wide part's columns of ids / their field offsets
        self.Fw = self.F
        if self.wide_fields is not None:
            wi = [f for f, on in enumerate(self.wide_fields) if on]
            self.Fw = len(wi)
            self.wide_idx = torch.tensor(wi, dtype=torch.int64, device=dev)
            self.wide_off_t = torch.from_numpy(off[:-1][wi].copy()).to(dev)
            self.wide_max_vocab = max([self.vocab_sizes[f] for f in wi] or [1])

        f32 = dict(dtype=torch.float32, device=dev)
        # A table row and its optimizer slots are ONE record [w | slot0 | slot1] of (1 + slots) * E floats: table, t_s0 and
        # t_s1 are strided views of it and every kernel that walks rows takes the record stride self.ts (include/
        # mi355x_rec.h: table_stride).  The sparse apply and the catch-up, which read and write a row's whole state, then
        # visit one contiguous 12 E-byte run per row and direction instead of three 4 E-byte runs in three arrays.
        sparse_lin_opt = self.lin_opt or self.opt
        self.table = self.t_s0 = self.t_s1 = self.t_rec = None
        self.ts = self.E
        if self.use_emb and self.F:
            a, b = self.opt.slot_init
            nsl = (a is not None) + (b is not None)
            if self.ROW_RECORDS:
                self.ts = (1 + nsl) * self.E
                self.t_rec = torch.zeros(self.R_local, self.ts, **f32)
                self.table = self.t_rec[:, :self.E]
                if a is not None:
                    self.t_s0 = self.t_rec[:, self.E:2 * self.E]
                    self.t_s0.fill_(a)
                if b is not None:
                    self.t_s1 = self.t_rec[:, 2 * self.E:3 * self.E]
                    self.t_s1.fill_(b)
            else:                                            # (A/B runs: three [R, E] arrays, the layout of rounds 1-3)
                self.table = torch.zeros(self.R_local, self.E, **f32)
                self.t_s0, self.t_s1 = self._slots(self.table, self.opt)
        t_adam = self.opt.name == "Adam" and self.table is not None
        l_adam = sparse_lin_opt.name == "Adam" and self.use_linear and self.F > 0
        self.adam_rows = t_adam or l_adam
        # The wide part's per-row state is ONE 16-byte record {weight, slot0, slot1, Adam stamp}: lin_w, l_s0,
        # l_s1 and last_step are strided views of it (self.ls = 4 elements), so the catch-up and the apply
        # touch one memory sector per row instead of four.  Without a wide part the stamps are a plain array.
        self.lin_state = self.lin_w = self.l_s0 = self.l_s1 = self.last_step = None
        self.ls = 1
        if self.use_linear and self.F:
            self.lin_state = torch.zeros(self.R_local, 4, **f32)
            self.ls = 4
            self.lin_w = self.lin_state[:, 0]
            a, b = sparse_lin_opt.slot_init
            if a is not None:
                self.l_s0 = self.lin_state[:, 1]
                self.l_s0.fill_(a)
            if b is not None:
                self.l_s1 = self.lin_state[:, 2]
                self.l_s1.fill_(b)
            if self.adam_rows:
                self.last_step = self.lin_state.view(torch.int32)[:, 3]
        elif self.adam_rows:
            self.last_step = torch.zeros(self.R_local, dtype=torch.int32, device=dev)
        if shard is not None and self.F == 0:
            raise NotImplementedError("a model without categorical columns has nothing to shard")
        # lr_t schedules (TF keeps beta powers per optimizer): one for `optimizer`, one for `linear_optimizer` — the same
        # object when the two Adams agree (or there is only one), two tables when they differ (round 2 refused that)
        same = lambda a, b: (a.lr, a.beta1, a.beta2, a.epsilon) == (b.lr, b.beta1, b.beta2, b.epsilon)
        self.sched = AdamSchedule(self.opt, dev) if self.opt.name == "Adam" else None
        if sparse_lin_opt.name != "Adam":
            self.lin_sched = None
        elif self.sched is not None and same(sparse_lin_opt, self.opt):
            self.lin_sched = self.sched
        else:
            self.lin_sched = AdamSchedule(sparse_lin_opt, dev)
        if self.sched is None and self.lin_sched is not None and self.lin_opt is None:
            self.sched = self.lin_sched

        # dense variables: one flat buffer
        # D_in: logical width of the MLP input; D: its width in memory (raw numeric columns: padded with
        # zero columns to a multiple of 32, so that layer 1 keeps whole k-tiles; the matching rows of
        # kernel_0 are zero, get zero gradients and stay zero under every optimizer here)
        n_emb = self.F + (0 if self.raw_numeric else self.n_numeric)
        self.D_emb = n_emb * self.E if self.use_emb else 0
        self.D_in = self.D_emb + (self.n_numeric if (self.raw_numeric and self.use_dnn) else 0)
        # rows of kernel_0 (as stored) that the LOGICAL input layer owns: all of them, or — with per-field embedding
        # dimensions — the first field_dims[f] of every field's E rows (+ the raw numeric columns the deep part reads)
        self._k0_rows = None
        if self.use_dnn and (self.field_dims is not None or self.deep_numeric is not None):
            rows = [f * self.E + j for f, d in enumerate(self.field_dims or [self.E] * self.F) for j in range(d)]
            if self.raw_numeric:
                rows += [self.D_emb + j for j in range(self.n_numeric) if self.deep_numeric is None or self.deep_numeric[j]]
            self._k0_rows = np.asarray(rows, np.int64)
            self.D_logical = len(rows)
        # (wide input layers — config 4 at config 3's sizes, 1664 + 13 columns: to a multiple of 128, the k-tile of the planes
        # weight gradient; with the gather writing the numeric columns into the planes itself, see pl_numeric below)
        d_align = 128 if (self.D_in >= 512 and use_dnn and len(hidden_units) > 0 and int(hidden_units[0]) % 128 == 0) else 32
        self.D = _align(self.D_in, d_align) if (self.raw_numeric and self.use_dnn) else self.D_in
        self.layers = []                         # (kernel_off, bias_off, fan_in as stored, fan_out)
        o = 0
        if self.use_dnn:
            fan = self.D
            for h in self.hidden + [1]:
                k_off = o; o = _align(o + fan * h)
                b_off = o; o = _align(o + h)
                self.layers.append((k_off, b_off, fan, h))
                fan = h
        self.dnn_end = o                         # end of the MLP parameter block
        self.num_emb_off = self.lin_num_off = None
        if self.n_numeric and not self.raw_numeric:
            self.num_emb_off = o; o = _align(o + self.n_numeric * self.E)
        self.wide_off = o                        # [0, wide_off): TF's "dnn" scope; [wide_off, P): "linear" scope
        self.lin_bias_off = o; o = _align(o + 1)
        if self.n_numeric and self.use_linear:
            self.lin_num_off = o; o = _align(o + self.n_numeric)
        self.P = o
        self.dense = torch.zeros(self.P, **f32)
        self.d_grad = torch.zeros(self.P, **f32)
        self.d_s0, self.d_s1 = self._slots(self.dense, self.opt)
        # slots of the flat buffer that belong to no variable of the model and must stay 0: kernel_0's rows of a raw
        # numeric column the deep part does not read (their gradient x^T dY is not zero by itself) and the linear
        # weight of one the wide part does not read.  Their gradients are cleared before the dense apply.
        frozen = []
        if self.deep_numeric is not None and self.use_dnn:
            k_off, _, _, h0 = self.layers[0]
            for j, on in enumerate(self.deep_numeric):
                if not on:
                    frozen += list(range(k_off + (self.D_emb + j) * h0, k_off + (self.D_emb + j + 1) * h0))
        if self.wide_numeric is not None and self.lin_num_off is not None:
            frozen += [self.lin_num_off + j for j, on in enumerate(self.wide_numeric) if not on]
        self._frozen = torch.tensor(frozen, dtype=torch.int64, device=dev) if frozen else None
        if self.lin_opt is not None:
            # wide-part dense variables (linear bias, numeric linear weights) follow linear_optimizer
            self.dl_s0, self.dl_s1 = self._slots(self.dense, self.lin_opt)
        self._ws = {}
        self._alloc_gen = 0                      # bumped whenever a workspace / planes buffer moves (see graph_train_step)
        self._final_step = 0
        self._presorted = None
        # Without numeric columns layer 1 of the MLP reads its input straight from the embedding table
        # (multi-GPU: from the receive buffer of the row exchange) as a gathered GEMM operand and the
        # concat [B, F*E] is never materialised.
        self.gather_mlp = self.use_dnn and self.n_numeric == 0 and self.F > 0
        # Matrix-pipe path of the MLP GEMMs (all: fp32 in, fp32 accumulate, fp32-level error):
        #   "f16x2"  operands as fp16 high + low parts, three products per k-step: forward and data gradient
        #            on pre-split planes with one exponent per ROW (self.planes, below; layers whose widths
        #            are not multiples of 16 fall back to bf16x3), the weight gradient on the same planes with
        #            every example brought to matrix-wide scales (abs-max vectors the producers emit:
        #            self._amax) or, for ragged shapes, on fp32 copies split with one exponent per matrix;
        #   "bf16x3" three bf16 parts, six products, no scales;   "fp32"  fp32-input MFMA.
        if gemm not in ("f16x2", "bf16x3", "fp32"):
            raise ValueError("gemm must be 'f16x2', 'bf16x3' or 'fp32'")
        self.gemm = gemm
        self._amax = torch.zeros(_lib.AMAX_SLOTS * (2 * len(self.layers) + 4), dtype=torch.float32, device=self.device)
        self._amax_idx = {}
        # Planes path (f16x2 only): forward and data-gradient GEMMs read operands that their producers left
        # as fp16 high/low planes with a per-row exponent (no split arithmetic in the GEMM loop, LDS-DMA
        # staging, 512-column tiles).  Every hidden layer's widths must be multiples of 16.  The weight gradient
        # reads the same planes where whole 128 x 128 x 32 tiles fit (_wgrad_planes_ok: one power of two per
        # example brings its rows to matrix-wide scales), and the fp32 copies are then not written at all;
        # otherwise it runs on fp32 copies (gemm.hip, matrix-wide scales).
        self.planes = (gemm == "f16x2" and self.use_dnn and len(self.hidden) > 0 and self.act == 1 and
                       getattr(self.k, "supports_planes", False) and self.D % 16 == 0 and
                       all(h % 16 == 0 for h in self.hidden))
        self._pl = {}
        self._acts_in_planes = set()      # hidden layers whose last training output exists as planes only
        self.summaries_next = False       # set before a train step whose layer_summaries() will be recorded (Estimator.train)
        # layer 1's operand: the gather kernel writes it as planes when the embedding size allows; otherwise
        # the concat is materialised in fp32 (as with numeric columns) and split
        self.pl_gather_ok = self.E % 16 == 0 and self.E >= 32 and self.F <= 48
        if self.planes and self.gather_mlp and not self.pl_gather_ok:
            self.gather_mlp = False
        # raw numeric columns (canned Wide&Deep, config 4) on the planes path: the gather writes them into the planes as
        # the concat's tail, under the example's exponent — no fp32 concat, no abs-max pass over it, no split
        # (config 4 at config 3's sizes: 0.36 + 0.12 ms of a 3.0 ms step, and the fp32 concat's 0.47 GB written)
        self.pl_numeric = bool(self.planes and self.raw_numeric and self.use_dnn and self.F > 0 and self.pl_gather_ok and
                               0 < self.D - self.F * self.E <= 4 * self.E and getattr(self.k, "supports_planes", False))

    # ------------------------------------------------------------------ variables
    def _slots(self, like, spec):
        if like is None:
            return None, None
        a, b = spec.slot_init
        s0 = torch.full_like(like, a) if a is not None else None
        s1 = torch.full_like(like, b) if b is not None else None
        return s0, s1

    def reset_optimizer_state(self):
        """Slots back to their TF initial values, global step 0 (after a warm start from variables only)."""
        lin_spec = self.lin_opt or self.opt
        for t, spec in ((self.t_s0, self.opt), (self.d_s0, self.opt), (self.l_s0, lin_spec),
                        (getattr(self, "dl_s0", None), self.lin_opt)):
            if t is not None:
                t.fill_(spec.slot_init[0])
        for t, spec in ((self.t_s1, self.opt), (self.d_s1, self.opt), (self.l_s1, lin_spec),
                        (getattr(self, "dl_s1", None), self.lin_opt)):
            if t is not None:
                t.fill_(spec.slot_init[1])
        if self.last_step is not None:
            self.last_step.zero_()
        self.step = 0
        self._final_step = 0

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
        weights / biases, glorot-uniform kernels.  torch's generator, not TF's Philox stream.
        Dense variables must be identical on every rank: seed the generator identically or
        broadcast afterwards (parallel.broadcast_dense)."""
        g = generator
        if self.table is not None:
            s = 1.0 / math.sqrt(self.E)
            torch.nn.init.trunc_normal_(self.table, 0.0, s, -2.0 * s, 2.0 * s, generator=g)
            if self.field_dims is not None:          # a column of dimension d: N(0, 1/sqrt(d)) in its d columns, 0 in the rest
                for f, d in enumerate(self.field_dims):
                    blk = self.table[slice(*self._field_rows(f))]
                    if d:
                        blk[:, :d].mul_(math.sqrt(self.E / d))
                    blk[:, d:].zero_()
        if self.lin_w is not None and lin_scale:
            self.lin_w.normal_(0.0, lin_scale, generator=g)
            if self.wide_fields is not None:
                for f, on in enumerate(self.wide_fields):
                    if not on:
                        self.lin_w[slice(*self._field_rows(f))].zero_()
        for i, (_, _, fan, h) in enumerate(self.layers):
            fan_in = (self.D_logical if self._k0_rows is not None else self.D_in) if i == 0 else fan
            lim = math.sqrt(6.0 / (fan_in + h))
            self.kernel(i).uniform_(-lim, lim, generator=g)
            if i == 0 and self._k0_rows is not None:
                keep = torch.zeros(fan, dtype=torch.bool, device=self.device)
                keep[torch.from_numpy(self._k0_rows).to(self.device)] = True
                self.kernel(0)[~keep] = 0.0              # rows of the columns a narrower embedding does not have, and of the pad
            elif i == 0 and fan_in < fan:
                self.kernel(0)[fan_in:].zero_()            # rows of the zero pad columns
        if self.num_emb_off is not None:
            lim = math.sqrt(6.0 / (self.n_numeric + self.E))
            self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E)).uniform_(-lim, lim, generator=g)
        if self._frozen is not None:
            self.dense.index_fill_(0, self._frozen, 0.0)

    def _field_rows(self, f):
        """[lo, hi): the rows of field f in this rank's tables (row r of the model lives on rank r % world at r // world)"""
        off = self.field_off_host
        if self.shard is None:
            return int(off[f]), int(off[f + 1])
        r, w = self.shard.rank, self.shard.world
        return (int(off[f]) - r + w - 1) // w, (int(off[f + 1]) - r + w - 1) // w

    def _my_rows(self, a):
        """rows of a [R, ...] array that live on this rank, in local order"""
        return a if self.shard is None else a[self.shard.rank::self.shard.world]

    def load_oracle_params(self, p):
        """Copy an ``oracle.deepfm.Params`` (numpy) into the device buffers (tests / smoke)."""
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        if self.table is not None:
            emb = p.emb
            if self.field_dims is not None:              # narrower columns: zero-padded to E
                emb = [np.pad(a, ((0, 0), (0, self.E - a.shape[1]))) for a in p.emb]
            self.table.copy_(t(self._my_rows(np.concatenate(emb, 0))))
        if self.lin_w is not None:
            self.lin_w.copy_(t(self._my_rows(np.concatenate(p.lin_w, 0))))
        for i in range(len(self.layers)):
            k = self.kernel(i)
            k.zero_()
            if i == 0 and self._k0_rows is not None:
                k[torch.from_numpy(self._k0_rows).to(self.device)] = t(p.mlp[0][0])
            else:
                k[:p.mlp[i][0].shape[0]].copy_(t(p.mlp[i][0]))
            self.bias(i).copy_(t(p.mlp[i][1]))
        self.dense[self.lin_bias_off] = float(p.lin_bias[0])
        if self.num_emb_off is not None:
            self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E)).copy_(t(p.num_emb))
        if self.lin_num_off is not None:
            self._seg(self.dense, self.lin_num_off, (self.n_numeric,)).copy_(t(p.lin_num))
        if self._frozen is not None:
            self.dense.index_fill_(0, self._frozen, 0.0)

    def export_numpy(self):
        """Variables as numpy arrays (after bringing Adam rows up to date).  Sparse variables are
        returned per field for a single GPU, as the local shard ("table", "lin_w_local") otherwise."""
        self.finalize_rows()
        rows = lambda i: self.D_in if i == 0 else self.layers[i][2]          # without the zero pad rows

        def kern(i):
            if i == 0 and self._k0_rows is not None:                         # the logical input layer's rows only
                return self.kernel(0)[torch.from_numpy(self._k0_rows).to(self.device)].cpu().numpy()
            return self.kernel(i)[:rows(i)].cpu().numpy()
        out = {"mlp": [(kern(i), self.bias(i).cpu().numpy()) for i in range(len(self.layers))],
               "lin_bias": self.dense[self.lin_bias_off:self.lin_bias_off + 1].cpu().numpy()}
        if self.shard is None:
            off = self.field_off_host
            sp = lambda a: [a[off[f]:off[f + 1]].cpu().numpy() for f in range(self.F)] if a is not None else None
            out.update(emb=sp(self.table), lin_w=sp(self.lin_w))
            if self.wide_fields is not None and out["lin_w"] is not None:
                # a column outside linear_feature_columns owns no linear weight: its slots of lin_w are never read (the
                # sparse apply does write them — one kernel serves every row of the batch — and nothing looks)
                out["lin_w"] = [a if on else None for a, on in zip(out["lin_w"], self.wide_fields)]
            if self.field_dims is not None and out["emb"] is not None:
                out["emb_padding_max_abs"] = max([float(np.abs(a[:, d:]).max()) if a[:, d:].size else 0.0
                                                  for a, d in zip(out["emb"], self.field_dims)])
                out["emb"] = [a[:, :d] for a, d in zip(out["emb"], self.field_dims)]
        else:
            out.update(table=None if self.table is None else self.table.cpu().numpy(),
                       lin_w_local=None if self.lin_w is None else self.lin_w.cpu().numpy())
        if self.num_emb_off is not None:
            out["num_emb"] = self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E)).cpu().numpy()
        if self.lin_num_off is not None:
            out["lin_num"] = self._seg(self.dense, self.lin_num_off, (self.n_numeric,)).cpu().numpy()
        return out

    # ------------------------------------------------------------------ helpers
    def _buf(self, name, shape, dtype=torch.float32):
        n = int(np.prod(shape))
        cur = self._ws.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            cur = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._ws[name] = cur
            self._alloc_gen += 1
        return cur[:n].view(*shape)

    def _bytes(self, name, nbytes):
        n = (int(nbytes) + 255) // 256 * 256 + 256
        cur = self._ws.get(name)
        if cur is None or cur.numel() < n:
            cur = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws[name] = cur
            self._alloc_gen += 1
        return cur

    def _planes(self, name, rows, K):
        cur = self._pl.get(name)
        if cur is None or cur.rows < rows or cur.K != K:
            cur = self._pl[name] = PlaneBuf(rows, K, self.device)
            self._alloc_gen += 1
        return cur.struct

    def _av(self, name):
        """The abs-max vector called `name` (a slice of self._amax; zeroed by _forward each step)."""
        i = self._amax_idx.setdefault(name, len(self._amax_idx))
        return self._amax[i * _lib.AMAX_SLOTS:(i + 1) * _lib.AMAX_SLOTS]

    def _ga(self, a, b, out):
        """mi_gemm_amax_t for one GEMM call: names of the operands' / result's abs-max vectors.
        None unless gemm == 'f16x2' (the library then takes the bf16x3 or fp32 path)."""
        if self.gemm != "f16x2":
            return None
        p = lambda n: None if n is None else ptr(self._av(n))
        return _lib.GemmAmax(p(a), p(b), p(out))

    def _split_weights(self, train, amax=None):
        """Planes of every hidden layer's kernel in one launch per step: transposed (rows = output units) for
        the forward pass, as stored (rows = inputs) for the data gradient; one exponent for the whole
        parameter block, from its abs-max."""
        hidden = self.layers[:-1]
        # once per step, not once per forward: the chunks of a pipelined multi-GPU step see the same weights (the dense
        # variables change in _apply, which moves self.step, or through torch, which moves the tensor's version)
        stamp = (self.step, self.dense._version)
        done = getattr(self, "_wsplit", None)
        if done is not None and done[:2] == stamp and (done[2] or not train) and not getattr(self, "_capturing", False):
            ev = getattr(self, "_wsplit_event", None)
            if ev is not None:            # launched ahead on the side stream (_split_weights_ahead): the GEMMs wait for it here
                torch.cuda.current_stream().wait_event(ev)
                self._wsplit_event = None
            return
        self._wsplit = stamp + (bool(train),)
        key = "wjobs_train" if train else "wjobs_eval"
        jobs = self._ws.get(key)
        if jobs is None:
            # one launch takes MI_MAX_WEIGHT_JOBS layers; a deeper MLP takes several (round 2 refused it)
            jobs = []
            for lo in range(0, len(hidden), _lib.MAX_WEIGHT_JOBS):
                part = hidden[lo:lo + _lib.MAX_WEIGHT_JOBS]
                arr = (_lib.WeightJob * len(part))()
                for j, (k_off, _, fan, h) in enumerate(part):
                    i = lo + j
                    arr[j].offset, arr[j].K, arr[j].N = k_off, fan, h
                    arr[j].wt = self._planes("wt%d" % i, h, fan)
                    if train:
                        arr[j].w = self._planes("w%d" % i, fan, h)
                jobs.append(arr)
            self._ws[key] = jobs
        amax = self._av("w") if amax is None else amax          # (an abs-max vector zeroed by the caller's stream)
        self.k.mi_absmax(self.dense, self.dnn_end, amax)
        for arr in jobs:
            self.k.mi_split_weights(self.dense, arr, len(arr), amax)

    def _split_weights_ahead(self):
        """The weight planes of a train step, started on a side stream at the HEAD of the step: they depend on the dense
        variables only (final since the previous step's apply), while the step's first half — sort, catch-up, gather —
        does not touch the MLP.  Two small latency-bound launches (abs-max, split: ~40 us at config 3) leave the critical
        path; the first GEMM waits for their event (in _split_weights)."""
        if not self.planes or self.device.type != "cuda" or getattr(self, "_capturing", False) or not self.WSPLIT_AHEAD:
            return
        side = self._ws.get("wsplit_stream")
        if side is None:
            side = self._ws["wsplit_stream"] = self._new_side_stream()
        # (the planes' and job buffers must exist before another stream writes them: sized on this stream by an earlier step)
        if "wjobs_train" not in self._ws:
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._wsplit_event = None
            # (its own abs-max vector: _forward zeroes the shared ones on the main stream meanwhile)
            amax = self._ws.get("w_amax_ahead")
            if amax is None:
                amax = self._ws["w_amax_ahead"] = torch.zeros(_lib.AMAX_SLOTS, dtype=torch.float32, device=self.device)
            amax.zero_()
            self._split_weights(True, amax)
            ev = torch.cuda.Event()
            ev.record(side)
        self._wsplit_event = ev

    def _layer_seed(self, layer):
        rank = 0 if self.shard is None else self.shard.rank
        chunk = getattr(self, "_chunk", 0)              # pipelined multi-GPU step: a mask per chunk
        # (while a step is being captured into a hipGraph the step term comes from the device-resident step state:
        # mi_step_advance sets seed_term = step * 1000003 and the kernels add it — the same masks as the eager step)
        step_term = 0 if getattr(self, "_capturing", False) else (self.step + 1) * 1000003
        return (self.seed * 0x9E3779B97F4A7C15 + step_term + layer * 7919 + rank * 104729 +
                chunk * 15485863) & (2 ** 64 - 1)

    @property
    def timers(self):
        return self.k.timers

    @timers.setter
    def timers(self, v):
        self.k.timers = v

    # ------------------------------------------------------------------ forward
    def _forward(self, ids, x_num, train, src=None, pieces=None):
        """ids [B,F] int32; returns the logit components and caches activations for the backward.
        src = (table, lin_w, field_off, ids) overrides where rows are read from (sharded path: the
        rows received from their owners, addressed by slot).
        pieces (sharded path): [(b0, b1, ready)] — the embedding-side kernels run once per range of examples, each after
        ready() has ordered the stream behind that range's row exchange; the MLP runs once, on the whole batch."""
        k = self.k
        B = ids.shape[0]
        self._last_B = B
        c = {"B": B}
        # (rows received from their owners: src also carries the receive buffer's row / weight strides — E and 1 for plain
        # arrays, E + 4 for the packed exchange's records; the model's own rows: one [w | slots] record apart)
        table, lin_w, field_off, rid, tst, ls = src if src is not None else (self.table, self.lin_w, self.field_off, ids, self.ts, self.ls)
        # the wide part's view of the batch: all columns, or (wide_fields) the wide columns' ids and field offsets
        w_off, w_ids, Fw = field_off, rid, self.F
        if self.wide_idx is not None:
            w_ids = self._buf("wide_ids", (B, self.Fw), rid.dtype)
            torch.index_select(rid, 1, self.wide_idx, out=w_ids)             # (a column copy: torch as plumbing)
            # (rows received from their owners are addressed by slot: every field's offset is 0 there)
            w_off, Fw = (self.wide_off_t if src is None else field_off[:self.Fw]), self.Fw
            c["wide_ids"] = w_ids
        concat = sumv = fm = None
        ld = self.D
        gathered = self.gather_mlp
        no_concat = gathered or self.pl_numeric          # layer 1 reads the rows / the planes the gather wrote
        F = self.F
        if self.use_emb:
            concat = None if no_concat else self._buf("concat", (B, ld))
            sumv = self._buf("sumv", (B, self.E)) if self.use_mf else None
            fm = self._buf("fm", (B,)) if self.use_mf else None
        elif self.raw_numeric and self.use_dnn:
            concat = self._buf("concat", (B, ld))
        lin = self._buf("lin", (B,)) if self.use_linear else None
        f16 = self.gemm == "f16x2" and self.use_dnn
        if f16:
            self._amax.zero_()
        rows_amax = self._av("x0") if (f16 and no_concat) else None
        # planes path, layer 1: the gather itself writes the concat as planes (one exponent per example)
        pl_gather = self.planes and no_concat and self.pl_gather_ok
        tail = (x_num, self.n_numeric, self.D - F * self.E) if self.pl_numeric else (None, 0, 0)
        # The wide part's 4-byte weight gathers drag a whole sector each through the row-gather kernel
        # (3.9 vs 4.9 TB/s of row bytes).  On a single GPU they run as their own kernel on a side stream
        # under the matrix-bound layer-1 GEMM instead; the head joins the two streams.
        # (from a few thousand examples on: below that a fork / join costs more than the 4-byte gathers it hides)
        side_lin = src is None and self._wide_on_side_stream(B)
        c["lin_join"] = None
        if F == 0:
            # numeric columns only (deep_fm.py:57-70 allows it): the sums start from zero (memsets)
            for t in (sumv, fm, lin):
                if t is not None:
                    t.zero_()
        elif side_lin:
            if pl_gather:
                k.mi_embed_fm_planes_fwd(table, field_off, rid, B, F, self.E, sumv, fm, self._planes("x0p", B, ld), rows_amax, *tail, tst)
            elif concat is not None or sumv is not None or rows_amax is not None:
                k.mi_embed_fm_linear_fwd(table, None, field_off, rid, B, F, self.E, concat, ld, sumv, fm, None,
                                         rows_amax, 1, tst)
            side = self._side_stream()
            side.wait_stream(torch.cuda.current_stream())        # (w_ids; the wide part's own catch-up is on this stream already)
            with torch.cuda.stream(side):
                k.tagged("mi_embed_fm_linear_fwd", "/wide")(None, lin_w, w_off, w_ids, B, Fw, self.E, None, 0, None,
                                                            None, lin, None, ls, 0)
            c["lin_join"] = side
        elif pl_gather:
            x0p = self._planes("x0p", B, ld)
            for b0, b1, ready in (pieces or [(0, B, None)]):
                if ready is not None:
                    ready()
                sl = slice(b0, b1)
                # (rows b0.. of a k-block-major planes matrix: every block's rows are contiguous, 64 B each)
                xp_ = x0p if (b0 == 0 and b1 == B) else _lib.Planes(x0p.data + 64 * b0, x0p.row_exp + 4 * b0, x0p.blk_stride)
                k.mi_embed_fm_planes_fwd(table, field_off, rid[sl], b1 - b0, F, self.E, None if sumv is None else sumv[sl],
                                         None if fm is None else fm[sl], xp_, rows_amax,
                                         None if tail[0] is None else tail[0][sl], tail[1], tail[2], tst)
                if lin is not None:
                    k.tagged("mi_embed_fm_linear_fwd", "/wide")(None, lin_w, w_off, w_ids[sl], b1 - b0, Fw, self.E, None, 0, None, None,
                                                                lin[sl], None, ls, 0)
        elif concat is not None or sumv is not None or lin is not None or rows_amax is not None:
            emb_on = self.use_emb
            one_call = self.wide_idx is None                         # (a wide part on other columns: a call of its own)
            for b0, b1, ready in (pieces or [(0, B, None)]):
                if ready is not None:
                    ready()
                sl = slice(b0, b1)
                v = lambda t: None if t is None else t[sl]
                if emb_on or one_call:
                    k.mi_embed_fm_linear_fwd(table if emb_on else None, lin_w if (self.use_linear and one_call) else None, field_off,
                                             rid[sl], b1 - b0, F, self.E, v(concat) if emb_on else None, ld, v(sumv), v(fm),
                                             v(lin) if one_call else None, rows_amax, ls, tst if emb_on else 0)
                if not one_call and lin is not None:
                    if Fw:
                        k.mi_embed_fm_linear_fwd(None, lin_w, w_off, w_ids[sl], b1 - b0, Fw, self.E, None, 0, None, None, lin[sl], None, ls, 0)
                    else:
                        lin[sl].zero_()
        elif pieces:
            for _, _, ready in pieces:           # (nothing reads the rows here, but the exchanges must be joined)
                if ready is not None:
                    ready()
        if self.n_numeric:
            wn = self._seg(self.dense, self.lin_num_off, (self.n_numeric,)) if self.use_linear else None
            if self.raw_numeric:
                # canned input_layer: the values themselves (+ zero pad) follow the embedding columns (pl_numeric: the
                # gather has written them into the planes; what is left here is the wide part's x . w)
                if concat is not None or lin is not None:
                    k.mi_numeric_raw_fwd(x_num, wn, B, self.n_numeric, concat, ld, self.D_emb, ld - self.D_emb, lin)
            else:
                V = self._seg(self.dense, self.num_emb_off, (self.n_numeric, self.E))
                k.mi_numeric_embed_fwd(x_num, V, wn, B, self.n_numeric, self.E, concat, ld, F * self.E, sumv, fm, lin)
        acts = []
        dnn = None
        k.query("mi_set_gemm_mode", 0 if self.gemm == "fp32" else 1)
        if self.use_dnn:
            x, ldx = concat, ld
            if f16:
                if not no_concat:
                    k.mi_absmax(concat, B * ld, self._av("x0"))
                if not self.planes:
                    # one bound for every layer's weights: the abs-max over the whole MLP parameter block
                    k.mi_absmax(self.dense, self.dnn_end, self._av("w"))
            keep = 1.0 - self.dropout if (train and self.dropout > 0) else 1.0
            nh = len(self.layers) - 1
            if self.planes:
                if not pl_gather:           # the concat exists in fp32 (numeric columns, small E): split it
                    k.mi_split_rows(concat, ld, B, ld, 0, self._planes("x0p", B, ld), None)
                self._split_weights(train)
            xp = "x0p"
            # TRAIN on the planes path: the logits layer (N = 1) runs inside the fused logits + head launch (_head)
            c["tail_fused"] = bool(train and self._tail_fusable())
            c["top_fused"] = bool(c["tail_fused"] and self._top_fusable(B))
            for i, (_, _, fan, h) in enumerate(self.layers):
                last = i == nh
                y = self._buf("act%d" % i, (B, h))
                if last and c["tail_fused"]:
                    acts.append(y)
                    break
                if self.planes and not last:
                    if c["top_fused"] and i == nh - 1:
                        # the last hidden layer: inside the fused logits + head launch (_head), its output stays on the chip
                        c["top"] = (i, xp, keep, self._layer_seed(i))
                        self._acts_in_planes.discard(i)
                        acts.append(y)
                        x, ldx = y, h
                        continue
                    self._hidden_forward_planes(c, i, xp, y, keep, self._layer_seed(i), train)
                    xp = "x%dp" % (i + 1)
                elif i == 0 and gathered:
                    k.mi_dense_fwd_gathered(table, field_off, rid, F, self.E, self.kernel(0), self.bias(0), y, h,
                                            B, h, 0 if last else self.act, 1.0 if last else keep, self._layer_seed(0),
                                            self._ga("x0", "w", "x1"), tst)
                else:
                    k.mi_dense_fwd(x, ldx, self.kernel(i), self.bias(i), y, h, B, h, fan, 0 if last else self.act,
                                   1.0 if last else keep, self._layer_seed(i),
                                   None if self.planes else self._ga("x%d" % i, "w", "x%d" % (i + 1)))
                acts.append(y)
                x, ldx = y, h
            dnn = acts[-1].view(B)
            c["keep"] = keep
        c.update(concat=concat, sumv=sumv, fm=fm, lin=lin, dnn=dnn, acts=acts, x_num=x_num, ids=rid,
                 gathered=gathered, g_table=table, g_off=field_off, g_ts=tst)
        return c

    def _tail_fusable(self):
        """the logits layer + head of a train step as ONE launch (mi_logits_head_fused): planes path, a hidden layer of 64 /
        128 / 256 units below a one-unit logits layer"""
        if not self.TAIL_FUSED or not self.planes or len(self.layers) < 2 or not hasattr(self.k, "mi_logits_head_fused"):
            return False
        _, _, fan, h = self.layers[-1]
        return h == 1 and fan in (64, 128, 256)

    def _top_fusable(self, B):
        """the last hidden layer runs inside the fused logits + head launch too (mi_hidden_logits_head_fused): a 128-unit layer
        below the one-unit logits layer whose weight gradient reads planes, a batch that fills the chip, and nobody about to
        look at the layer's output (summaries_next: the Estimator says so before a step whose layer_summary it records)"""
        if not (self.TOP_FUSED and self._tail_fusable() and hasattr(self.k, "mi_hidden_logits_head_fused")):
            return False
        nh = len(self.layers) - 1
        if nh < 1 or B < self.TOP_FUSED_MIN_BATCH or getattr(self, "summaries_next", False):
            return False
        _, _, fan, h = self.layers[nh - 1]
        return h == 128 and fan % 16 == 0 and self._wgrad_planes_ok(B, nh - 1)

    def _hidden_forward_planes(self, c, i, xp, y, keep, seed, train):
        """hidden layer i's forward on the planes path (see _forward): the next layer's operand as planes straight from the
        epilogue when a workgroup owns whole rows (h <= 512), the fp32 copy where something reads it"""
        k = self.k
        B = c["B"]
        nh = len(self.layers) - 1
        _, _, fan, h = self.layers[i]
        need_p = i + 1 < nh
        yp = self._planes("x%dp" % (i + 1), B, h) if need_p else None
        direct = need_p and h <= 512
        # (no fp32 copy when its only reader, the next layer's weight gradient, takes the planes)
        planes_only = train and direct and self._wgrad_planes_ok(B, i + 1)
        (self._acts_in_planes.add if planes_only else self._acts_in_planes.discard)(i)
        # (training: the relu/dropout mask as one bit per output, for the data gradients — 1/32 of the bytes of
        # the stored activation they would otherwise read it from)
        mb = self._buf("mbits%d" % i, (B, (h + 31) // 32), torch.int32) if train else None
        k.mi_dense_fwd_planes(self._planes(xp, B, fan), self._pl["wt%d" % i].struct, self.bias(i),
                              None if planes_only else y, h,
                              yp if direct else None, B, h, fan, 1, keep, seed,
                              self._av("x%d" % (i + 1)), mb, 0 if mb is None else mb.shape[1])
        if need_p and not direct:
            k.mi_split_rows(y, h, B, h, 0, yp, None)

    def _head(self, c, labels, want_grad, global_batch=None):
        k = self.k
        B = c["B"]
        logits = self._buf("logits", (B,))
        loss = self._buf("loss", (1,)) if labels is not None else None
        dlogit = self._buf("dlogit", (B,)) if want_grad else None
        n = global_batch if global_batch is not None else B
        scale = np.float32(1.0 / n) if self.reduction == "mean" else np.float32(1.0)
        ws = self._bytes("head_ws", k.query("mi_head_workspace_bytes", B))
        if c.get("lin_join") is not None:
            torch.cuda.current_stream().wait_stream(c["lin_join"])
        lb = self.dense[self.lin_bias_off:] if self.use_linear else None
        # d loss / d linear bias = sum_b dlogit lands straight in the dense gradient buffer
        dsum = self.d_grad[self.lin_bias_off:] if (want_grad and self.use_linear) else None
        if c.get("top") is not None:
            nh = len(self.layers) - 1
            i, xp, keep, seed = c.pop("top")
            _, _, fan, h = self.layers[i]
            if want_grad and labels is not None:
                tws = self._bytes("top_ws", k.query("mi_hidden_logits_head_fused_workspace_bytes", B, h))
                k.mi_hidden_logits_head_fused(self._planes(xp, B, fan), self._pl["wt%d" % i].struct, self.bias(i), B, h, fan, 1, keep,
                                              seed, self.kernel(nh), self.bias(nh), c["lin"], lb, c["fm"], labels, float(scale),
                                              c["acts"][nh], logits, loss, dlogit, dsum, self.kernel(nh, self.d_grad),
                                              self.bias(nh, self.d_grad), self._planes("dy%dp" % (nh - 1), B, h),
                                              self._av("dy%d" % (nh - 1)), tws, tws.numel())
                c["dnn"] = c["acts"][nh].view(B)
                c["tail_done"] = True
                self._top_step = self.step
                return logits, loss, dlogit
            # (a forward made for training but no gradient asked for: the layer as its own launch after all)
            self._hidden_forward_planes(c, i, xp, c["acts"][i], keep, seed, True)
        if c.get("tail_fused"):
            nh = len(self.layers) - 1
            _, _, fan, _ = self.layers[nh]
            x = c["acts"][nh - 1]
            if want_grad and labels is not None:
                # forward AND backward of the logits layer with the head between them: dnn, logits, loss, dlogit, the bias
                # gradient(s), the layer's weight gradient and its data gradient as planes (+ fp32 where the layer below's
                # weight gradient still reads fp32) — one pass over the last hidden layer's output
                mb = self._ws.get("mbits%d" % (nh - 1))
                mb = mb[:B * ((fan + 31) // 32)].view(B, (fan + 31) // 32) if mb is not None else None
                dx = None if self._wgrad_planes_ok(B, nh - 1) else self._buf("dact%d" % nh, (B, fan))
                tws = self._bytes("tail_ws", k.query("mi_logits_head_fused_workspace_bytes", B, fan))
                k.mi_logits_head_fused(x, fan, self.kernel(nh), self.bias(nh), c["lin"], lb, c["fm"], labels, B, fan, float(scale),
                                       mb, 0 if mb is None else mb.shape[1], c["keep"], c["acts"][nh], logits, loss, dlogit, dsum,
                                       self.kernel(nh, self.d_grad), self.bias(nh, self.d_grad), self._planes("dy%dp" % (nh - 1), B, fan),
                                       dx, fan, self._av("dy%d" % (nh - 1)), tws, tws.numel())
                c["dnn"] = c["acts"][nh].view(B)
                c["tail_done"] = True
                return logits, loss, dlogit
            # (a forward made for training but no gradient asked for: the logits layer as its own launch after all)
            k.mi_dense_fwd(x, fan, self.kernel(nh), self.bias(nh), c["acts"][nh], 1, B, 1, fan, 0, 1.0, self._layer_seed(nh), None)
            c["dnn"] = c["acts"][nh].view(B)
        k.mi_sigmoid_ce_head(c["lin"], lb, c["fm"], c["dnn"], labels, B, float(scale), logits, loss, dlogit, dsum,
                             ws, ws.numel())
        return logits, loss, dlogit

    # ------------------------------------------------------------------ public steps
    def _prep(self, ids, labels, x_num):
        if ids.dtype != torch.int32 or not ids.is_contiguous() or ids.dim() != 2 or ids.shape[1] != self.F:
            raise ValueError("ids must be a contiguous int32 [B, %d] tensor" % self.F)
        if ids.device.type != self.device.type:
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
        return self.loss(ids, None, x_num)[1]

    def loss(self, ids, labels, x_num=None):
        """EVAL forward: (loss [1] or None, logits [B]) without touching any model variable
        (Adam rows that TF would have moved meanwhile are first brought up to date)."""
        self._prep(ids, labels, x_num)
        self.finalize_rows()
        if self.shard is not None:
            from . import parallel
            return parallel.sharded_eval_step(self, ids, labels, x_num)
        c = self._forward(ids, x_num, False)
        logits, loss, _ = self._head(c, labels, False)
        self._top_step = None            # (every layer's output of THIS forward is in memory: layer_summaries may show them all)
        return loss, logits

    def finalize_rows(self):
        """Bring every Adam row up to date (all-rows mi_sparse_catchup).  No-op when nothing is stale."""
        if not self.adam_rows or self._final_step == self.step:
            return
        self._catchup(None, None, self.R_local)
        self._final_step = self.step

    GAP_SORT_MIN = 16384      # entries from which sorting the touched rows by staleness pays for itself
    # Scheduling choices of the single-GPU step, as class attributes (no environment switches in the product; bench.py
    # --engine-opt NAME=0/1 flips one for an A/B run; every combination gives the same bits):
    WSPLIT_AHEAD = True       # weight planes of the step made on a side stream at its head (_split_weights_ahead)
    LIN_SIDE = True           # the wide part's catch-up on the wide part's stream, beside the row kernel (_catchup)
    BYGAP_AHEAD = True        # the next batch's staleness order made a step ahead (_by_gap_ahead)
    ROW_RECORDS = True        # a table row and its optimizer slots as one [w | slot0 | slot1] record (__init__)
    TAIL_FUSED = True         # logits layer + head + the layer's backward as one launch (_head: mi_logits_head_fused)
    WGRAD_BATCH = True        # the planes weight gradients of a backward pass as one batch after the data gradients (_backward_dense)
    TOP_FUSED = True          # ... and the last hidden layer with them, in its GEMM's epilogue (_head: mi_hidden_logits_head_fused)
    TOP_FUSED_MIN_BATCH = 4096
    GRAPH_SHAPES_MAX = 4      # captured steps kept at a time, one per batch shape (graph_train_step)
    TEST_SIDE_STREAMS = True  # side streams are tested to run beside the step's stream and each other (_new_side_stream)
    SIDE_PRIORITY = 0         # -1: side streams are created with high priority.  HIP serves every stream priority from its own pool
                              # of hardware queues, so a high-priority stream can never share a queue with the step's (normal-priority)
                              # stream; normal-priority streams share 4 queues by reference count and may (parallel._side_stream)

    def _new_side_stream(self, priority=None):
        """A stream whose kernels really run BESIDE the step's stream and beside the side streams handed out before it.
        torch.cuda.Stream() deals the streams of a pool of 32 per priority in turn and HIP spreads them over a few hardware
        queues (4 per priority by default); two streams on one queue run one after the other.  Which pool stream the engine
        gets depends on how many the process took before (the user's own streams; torch's ProcessGroupNCCL takes its
        stream from the same pool) — tools/stream_alias_probe.py, one GPU: with 3 streams taken before, the single-GPU step
        2.74 -> 2.92 ms; with 4, the row-sharded step 4.41 -> 8.26 ms.  So a candidate is TESTED (a long and a short spin
        kernel, three events: do they overlap?) against the current stream and the earlier side streams, and the next
        pool stream is tried if it does not.  The k-th side stream is found once per process, device and priority."""
        prio = self.SIDE_PRIORITY if priority is None else priority
        if not self.TEST_SIDE_STREAMS:                          # (bench.py --engine-opt TEST_SIDE_STREAMS=0: the next pool stream, as before)
            return torch.cuda.Stream(device=self.device, priority=prio)
        n = self.__dict__.setdefault("_n_side", {})
        k = n[prio] = n.get(prio, -1) + 1                       # (this engine's k-th side stream of that priority)
        return _tested_side_stream(self.device, prio, k)

    def _catchup(self, uniq, num_uniq, n_max, defer=False, by_gap=None):
        """defer: the rows are about to be applied in this same step by ONE mi_sparse_apply call, which
        then decays m and v itself from the old stamps — the catch-up moves w only (a third less HBM
        traffic).  Not with a separate linear optimizer (two apply calls would see each other's stamps).
        by_gap: the rows already in staleness order (made ahead by _presort)."""
        t_adam = self.opt.name == "Adam" and self.table is not None
        l_adam = (self.lin_opt or self.opt).name == "Adam" and self.lin_w is not None
        if not (t_adam or l_adam) or n_max == 0:
            return
        defer = bool(defer and uniq is not None and self.lin_opt is None)
        for sc in (self.sched, self.lin_sched):
            if sc is not None:
                sc.lr_t(self.step)  # make sure the table covers step
        if uniq is not None and n_max >= self.GAP_SORT_MIN:
            uniq = by_gap if by_gap is not None else self._rows_by_gap(uniq, num_uniq, n_max, self.step)
        flags = (1 if defer else 0) | (2 if self.catchup == "bounded" else 0)
        t_sched = self.sched if t_adam else None
        l_sched = (self.lin_sched if self.lin_opt is not None else (self.lin_sched or self.sched)) if l_adam else None
        side = None
        if t_sched is not None and l_sched is not None and t_sched is not l_sched:
            # two different Adams (tables vs wide part): one call each, with its own lr_t table and betas.  The table
            # call must not move the stamps the wide call still has to read: it runs second.
            assert not defer
            parts = [(None, l_sched, 4), (t_sched, None, 0)]          # (4 = MI_CATCHUP_KEEP_STAMPS)
        elif (defer and t_sched is not None and l_sched is not None and self.shard is None and self.device.type == "cuda"
              and self._wide_on_side_stream(n_max // max(self.F, 1)) and self.LIN_SIDE):
            # deferred: neither call writes a stamp or touches the other's state — the wide part's 16-byte records
            # (scattered, latency-bound: 0.08 ms) are replayed on the stream that will run the wide part's forward,
            # beside the row kernel; the head joins that stream before anything else reads them
            parts = [(None, l_sched, 0), (t_sched, None, 0)]
            side = self._side_stream()
            side.wait_stream(torch.cuda.current_stream())
        else:
            parts = [(t_sched, l_sched, 0)]
        for i, (ts, lsch, extra) in enumerate(parts):
            s = (ts or lsch).spec
            # (the wide part's own call is timed under its own key: bench.py prices the row kernel and the wide kernel apart)
            entry = self.k.tagged("mi_sparse_catchup", "/wide") if (ts is None and hasattr(self.k, "tagged")) else self.k.mi_sparse_catchup
            call = lambda entry=entry: entry(
                self.table if ts is not None else None, self.t_s0 if ts is not None else None,
                self.t_s1 if ts is not None else None, self.lin_w if lsch is not None else None,
                self.l_s0 if lsch is not None else None, self.l_s1 if lsch is not None else None,
                self.last_step, uniq, num_uniq, n_max, self.E, self.step, (ts or lsch).table, s.beta1, s.beta2,
                s.epsilon, flags | extra, self.ls, self.ts)
            if side is not None and i == 0:
                with torch.cuda.stream(side):
                    call()
            else:
                call()

    def _side_stream(self):
        side = self._ws.get("side_stream")
        if side is None:
            side = self._ws["side_stream"] = self._new_side_stream()
        return side

    def _wide_on_side_stream(self, B):
        """the wide part's forward of a B-example single-GPU batch runs as its own kernel on the side stream (_forward)"""
        return (self.device.type == "cuda" and self.use_emb and self.use_linear and self.use_dnn and self.n_numeric == 0
                and self.F > 0 and B >= 4096)

    def _sort_unique(self, keys, n, key_range, tag, ws_name="sort_ws", cap=None, slot=None):
        """mi_sort_unique_rows into persistent buffers named after `tag` (ws_name: a workspace of its own for a sort that
        runs on a side stream beside the main stream's).  cap: allocate for that many keys (a count that varies from step
        to step must not reallocate between the collectives of a multi-GPU step)."""
        i32 = torch.int32
        cap = max(n, cap or 0)
        sorted_entry = self._buf(tag + "_sorted", (cap,), i32)[:n]
        uniq = self._buf(tag + "_uniq", (cap,), i32)[:n]
        seg = self._buf(tag + "_seg", (cap + 1,), i32)[:n + 1]
        num_uniq = self._buf(tag + "_nu", (1,), i32)
        ws = self._bytes(ws_name, self.k.query("mi_sort_unique_workspace_bytes", cap))
        if slot is not None:          # (+ every entry's segment, written by the compaction: parallel._route)
            self.k.mi_sort_unique_rows_slots(keys, n, key_range, sorted_entry, uniq, seg, num_uniq, slot, ws, ws.numel())
        else:
            self.k.mi_sort_unique_rows(keys, n, key_range, sorted_entry, uniq, seg, num_uniq, ws, ws.numel())
        return sorted_entry, uniq, seg, num_uniq

    PRESORT_MIN = 16384       # entries from which sorting the next batch beside this step's catch-up pays

    def _sort_batch(self, ids, tag, side=False):
        """(1) of a train step: which rows does a batch touch — sort + unique (TF: unique / unsorted_segment_sum) into
        the persistent buffers named after `tag`.  Returns (sorted_entry, uniq, seg, num_uniq)."""
        k = self.k
        B = ids.shape[0]
        n = B * self.F
        if B % 4096 == 0 and self.F <= 64 and hasattr(k, "mi_sort_unique_fields"):
            # every field's ids sorted on their own: the radix passes cover a field's id range, not the whole table
            i32 = torch.int32
            sorted_entry, uniq = self._buf(tag + "_sorted", (n,), i32), self._buf(tag + "_uniq", (n,), i32)
            seg, num_uniq = self._buf(tag + "_seg", (n + 1,), i32), self._buf(tag + "_nu", (1,), i32)
            ws = self._bytes("sortf_ws", k.query("mi_sort_unique_fields_workspace_bytes", B, self.F))
            sort = k.tagged("mi_sort_unique_fields", "/next batch, side stream") if (side and hasattr(k, "tagged")) else k.mi_sort_unique_fields
            # (beside: on a side stream next to the catch-up the many-short-launches form, alone on the step's stream the fused one)
            sort(ids, self.field_off, B, self.F, self.max_vocab, sorted_entry, uniq, seg, num_uniq, ws, ws.numel(), 1 if side else 0)
            return sorted_entry, uniq, seg, num_uniq
        rows = self._buf("rows", (n,), torch.int32)
        k.mi_global_rows(ids, self.field_off, B, self.F, rows)
        return self._sort_unique(rows, n, self.R, tag)

    def _presort(self, next_ids, tag):
        """The sort of the NEXT batch, on a side stream, enqueued right before this step's catch-up: ~14 small launches
        that are bound by launch latency and leave most of the chip idle (0.17 ms at config 3 alone).  Beside the catch-up
        — whose resident workgroups fill half the wave slots: the sort ends about when the catch-up does, at +35 us for it.
        (Measured alternatives, kernel timelines of tools/step_timeline.py: beside the sparse apply the apply takes 713 us
        instead of 631 and the one-workgroup scan waits 350 us for a slot; beside the MLP's big GEMMs, whose grids are whole
        waves of workgroups over the 256 CUs, those lose 5-15 %; after the head, beside the small layer-2 / layer-3 kernels of the backward: +0.03 ms per step, A/B on one
        box.)  The sort is a pure function of next_ids.  The order the
        next catch-up will walk the rows in (by staleness) follows later in the step: _by_gap_ahead.  It reads the rows'
        stamps BEFORE this step's apply has written this batch's: a row of both batches is filed under the gap it had
        before (its real gap is 0) — the order only decides which rows share a wave, the catch-up kernels read every row's
        stamp themselves.  (Each stamp is read once: mi_catchup_rows_by_gap materialises its keys before it counts them.)
        The result is used by the next train_step if it is given that very tensor, unmodified; otherwise it is dropped."""
        side = self._ws.get("presort_stream")
        if side is None:
            side = self._ws["presort_stream"] = self._new_side_stream()      # (a high-priority stream: no effect, same-box A/B, profiles/r04_ab_layout_and_fold.md)
        main = torch.cuda.current_stream()
        side.wait_stream(main)                       # next_ids exists, this step's own sort has left the shared workspace
        with torch.cuda.stream(side):
            out = self._sort_batch(next_ids, tag, side=True)
        # (the tensor itself is kept: while it is alive its memory cannot come back as another batch's)
        self._presorted = {"ids": next_ids, "version": next_ids._version, "tag": tag, "sorted": out, "stream": side,
                           "by_gap": None, "by_gap_step": None}

    def _by_gap_ahead(self):
        """Second half of _presort, enqueued after the head: the staleness order of the next batch's rows, on the same side
        stream, behind this step's forward.  There — the logits layer, the head and the small layer-3 kernels, ~0.15 ms in
        which most of the chip idles — it costs nothing; right after the sort it ran beside the gather (HBM-bound: 176 ->
        231 us) and the layer-1 forward GEMM (310 -> 374 us: kernel timelines, tools/step_timeline.py)."""
        ps = self._presorted
        if ps is None or not self.adam_rows or not self.BYGAP_AHEAD:
            return
        n = ps["ids"].shape[0] * self.F
        if n < self.GAP_SORT_MIN:
            return
        side = ps["stream"]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ps["by_gap"] = self._rows_by_gap(ps["sorted"][1], ps["sorted"][3], n, self.step + 1, ps["tag"] + "_ahead", side=True)
        ps["by_gap_step"] = self.step + 1

    def _rows_by_gap(self, uniq, num_uniq, n_max, step_to, tag="", side=False):
        """rows of equal staleness into the same wave (the replay runs as long as a wave's stalest row)"""
        by_gap = self._buf("uniq_by_gap" + tag, (n_max,), torch.int32)
        ws = self._bytes("gap_ws" + tag, self.k.query("mi_sort_unique_workspace_bytes", n_max))
        f = self.k.tagged("mi_catchup_rows_by_gap", "/next batch, side stream") if (side and hasattr(self.k, "tagged")) else \
            self.k.mi_catchup_rows_by_gap
        f(uniq, num_uniq, self.last_step, n_max, step_to, self.ls, by_gap, ws, ws.numel())
        return by_gap

    def train_step(self, ids, labels, x_num=None, next_ids=None):
        """One optimizer.minimize(loss): returns (loss [1], logits [B]) device tensors, no host sync.
        next_ids (optional): the ids of the batch the NEXT call will be given — an input pipeline knows them (the
        reference prefetches its tf.data batches, ml_100k.py:42-61).  Their sort then runs beside this step's catch-up
        (see _presort) instead of at the head of the next step.  Results are those of the plain sequence, bit for bit.
        Three drivers share the kernels: this one dispatches — N > 1 GPUs: parallel.sharded_train_step; one GPU:
        _train_step_single; one GPU as a hipGraph replay: graph_train_step (which captures _train_step_single)."""
        self._prep(ids, labels, x_num)
        if self.shard is not None:
            from . import parallel
            return parallel.sharded_train_step(self, ids, labels, x_num, next_ids)
        return self._train_step_single(ids, labels, x_num, next_ids)

    def _take_presorted(self, ids):
        """The sort of `ids` if the previous step was told about this very tensor (and it is unmodified), else None."""
        ps, self._presorted = getattr(self, "_presorted", None), None
        if ps is None:
            return None
        torch.cuda.current_stream().wait_stream(ps["stream"])           # (also before the workspace is reused)
        if ps["ids"].data_ptr() == ids.data_ptr() and ps["ids"].shape == ids.shape and ps["version"] == ids._version:
            return ps
        return None

    def _announce(self, ids, next_ids, x_num, tag):
        """Start the next batch's sort on the side stream when it can pay (see _presort)."""
        B = ids.shape[0]
        if (next_ids is not None and self.device.type == "cuda" and B * self.F >= self.PRESORT_MIN and next_ids.shape == ids.shape
                and B % 4096 == 0 and self.F <= 64 and hasattr(self.k, "mi_sort_unique_fields")   # (its own workspace)
                and not getattr(self, "_capturing", False)):
            self._prep(next_ids, None, x_num)
            self._presort(next_ids, "own2" if tag == "own" else "own")

    def _train_step_single(self, ids, labels, x_num, next_ids):
        """The single-GPU step: sort/unique -> lazy Adam catch-up -> forward + head -> MLP backward -> dense apply +
        fused sparse apply."""
        B = ids.shape[0]
        n = B * self.F
        if n == 0:                       # numeric columns only: no sparse variable exists
            c = self._forward(ids, x_num, True)
            logits, loss, dlogit = self._head(c, labels, True)
            self._backward_dense(c, dlogit)
            self._apply(None, None, None, None, 0, None, None)
            return loss, logits
        self._split_weights_ahead()
        # (1) which rows does this batch touch: sort + unique (TF: unique/unsorted_segment_sum)
        ps = self._take_presorted(ids)
        tag = ps["tag"] if ps is not None else "own"
        sorted_entry, uniq, seg, num_uniq = ps["sorted"] if ps is not None else self._sort_batch(ids, tag)
        self._announce(ids, next_ids, x_num, tag)       # the next batch's sort, beside the catch-up (see _presort)
        # (2) TF Adam moved these rows on every step they sat out: replay that now
        if self.adam_rows and self.step > 0:
            by_gap = ps["by_gap"] if (ps is not None and ps.get("by_gap_step") == self.step) else None
            self._catchup(uniq, num_uniq, n, defer=True, by_gap=by_gap)
        # (3) forward + head
        c = self._forward(ids, x_num, True)
        logits, loss, dlogit = self._head(c, labels, True)
        self._by_gap_ahead()
        # (4) backward through the MLP (+ numeric embeddings)
        d_concat = self._backward_dense(c, dlogit)
        # (5) sparse apply on the unique rows; the per-entry row gradients (deep_fm.py:54,81-87,39
        #     backward) are rebuilt inside the kernel from d_concat / sumv / dlogit
        self._apply(uniq, seg, sorted_entry, num_uniq, n, None, None, fused=(d_concat, c["sumv"], dlogit))
        return loss, logits

    def _entry_grads(self, c, d_concat, dlogit, pos, d_rows=None, d_lin=None):
        """Per-entry row gradients written at slot pos[b,f] of d_rows / d_lin (caller's buffers in the
        sharded path: the send order of the whole step)."""
        B = c["B"]
        n = B * self.F
        if d_rows is None and self.use_emb:
            d_rows = self._buf("d_rows", (n, self.E))
        if d_lin is None and self.use_linear:
            d_lin = self._buf("d_lin", (n,))
        # no concat (gathered layer 1): the rows sit in slot order in the exchange's receive buffer
        rows = c["g_table"] if (c["concat"] is None and self.use_mf) else None
        self.k.mi_embed_fm_linear_bwd(d_concat, self.D, c["concat"], self.D, rows, c["sumv"],
                                      dlogit if self.use_mf else None, dlogit if self.use_linear else None, pos,
                                      B, self.F, self.E, d_rows, d_lin)
        return d_rows, d_lin

    def _wgrad_planes_ok(self, B, i):
        """layer i's weight gradient can read planes: a hidden layer, whole tiles (mi_dense_bwd_weight_planes)"""
        if not self.planes or i < 0 or i >= len(self.layers) - 1:
            return False
        _, _, fan, h = self.layers[i]
        return B % 32 == 0 and fan % 128 == 0 and h % 128 == 0 and hasattr(self.k, "mi_dense_bwd_weight_planes")

    def _backward_dense(self, c, dlogit, on_d_concat=None):
        """Fills self.d_grad (dense gradients) and returns d_concat [B, D] (or None).
        on_d_concat (the row-sharded step): called with d_concat as soon as the input layer's DATA gradient is enqueued —
        which then runs BEFORE that layer's weight gradient (both need only dY of layer 1): the requester-side segment sums
        and the gradient exchange start there and travel under the largest weight-gradient GEMM and the dense all-reduce."""
        k = self.k
        B = c["B"]
        d_concat = None
        if self.use_dnn:
            nh = len(self.layers) - 1
            keep = c["keep"]
            dy, lddy = dlogit, 1
            wsz = max(k.query("mi_dense_bwd_weight_planes_workspace_bytes" if self._wgrad_planes_ok(B, j) else
                              "mi_dense_bwd_weight_workspace_bytes", B, h, fan) for j, (_, _, fan, h) in enumerate(self.layers))
            ws = self._bytes("wgrad_ws", wsz)
            # Weight gradients that read planes are collected and run AFTER the last data gradient, as one batch
            # (mi_dense_bwd_weight_planes_batch: one launch for every layer's per-example factors, one GEMM per layer, one
            # launch for every layer's slab fold — at config 3 five launches instead of nine, ~40 us of small launches and
            # the gaps around them off the step).  A data gradient needs only the dY of the layer above; the factors need
            # the finished dY planes and their abs-max, which is why they could not move earlier one by one.
            batch = [] if (self.WGRAD_BATCH and hasattr(k, "mi_dense_bwd_weight_planes_batch")) else None
            for i in range(nh, -1, -1):
                _, _, fan, h = self.layers[i]
                if i == nh and c.get("tail_done"):        # (the fused logits + head launch made dW, db and the planes of dX)
                    dy, lddy = self._ws.get("dact%d" % nh), fan
                    dy = dy[:B * fan].view(B, fan) if dy is not None else None
                    continue
                x = c["acts"][i - 1] if i else c["concat"]
                ldx = self.layers[i - 1][3] if i else self.D
                # abs-max of dY: known for every layer but the last (dlogit feeds the N = 1 layer,
                # which takes the fp32 path anyway); each data gradient emits the next one
                dyn = "dy%d" % i if i < nh else None
                ga_w = self._ga("x%d" % i, dyn, None) if dyn else None
                ga_d = self._ga(dyn, "w", "dy%d" % (i - 1) if i else None) if (dyn or i) else None
                def weight_gradient(i=i, fan=fan, h=h, x=x, ldx=ldx, dy=dy, lddy=lddy, ga_w=ga_w):
                    if self._wgrad_planes_ok(B, i) and batch is not None:
                        batch.append((i, fan, h, ga_w))
                    elif self._wgrad_planes_ok(B, i):
                        k.mi_dense_bwd_weight_planes(self._pl["x%dp" % i].struct, self._pl["dy%dp" % i].struct,
                                                     self.kernel(i, self.d_grad), self.bias(i, self.d_grad), B, h, fan, ws,
                                                     ws.numel(), ga_w)
                    elif i == 0 and c["gathered"]:
                        k.mi_dense_bwd_weight_gathered(c["g_table"], c["g_off"], c["ids"], self.F, self.E, dy, lddy,
                                                       self.kernel(0, self.d_grad), self.bias(0, self.d_grad), B, h, ws,
                                                       ws.numel(), ga_w, c["g_ts"])
                    else:
                        if x is None:     # (pl_numeric, a batch the planes weight gradient does not take: the concat back in fp32)
                            x = self._buf("concat", (B, ldx))
                            k.mi_merge_rows(self._pl["x0p"].struct, B, ldx, x, ldx)
                        k.mi_dense_bwd_weight(x, ldx, dy, lddy, self.kernel(i, self.d_grad), self.bias(i, self.d_grad), B,
                                              h, fan, ws, ws.numel(), ga_w)
                data_first = i == 0 and on_d_concat is not None
                if not data_first:
                    weight_gradient()
                dx = self._buf("dact%d" % i, (B, fan))
                if self.planes and i < nh:
                    # dY of this layer is at hand as planes ("dy<i>p": written by the layer above); the mask is
                    # the stored activation's high plane; the result becomes the next dY (planes + fp32)
                    need_p = i > 0
                    dxp = self._planes("dy%dp" % (i - 1), B, fan) if need_p else None
                    direct = need_p and fan <= 512
                    # the mask of the layer below's output: its bits (written by the planes forward), else its planes
                    mb = self._ws.get("mbits%d" % (i - 1)) if i > 0 else None
                    mb = mb[:B * ((fan + 31) // 32)].view(B, (fan + 31) // 32) if mb is not None else None
                    xa = self._pl["x%dp" % i].struct if (i > 0 and mb is None and ("x%dp" % i) in self._pl and i < nh) else None
                    if i > 0 and xa is None and mb is None:      # (the last hidden layer's output has no planes: make them)
                        xa = self._planes("x%dp" % i, B, fan)
                        k.mi_split_rows(x, ldx, B, fan, 0, xa, None)
                    # (the fp32 copy only where something reads it: d_concat, or a weight gradient on fp32 operands)
                    need_f = i == 0 or not (direct and self._wgrad_planes_ok(B, i - 1))
                    k.mi_dense_bwd_data_planes(self._planes("dy%dp" % i, B, h), self._pl["w%d" % i].struct, xa,
                                               dx if need_f else None, fan,
                                               dxp if direct else None, B, h, fan, keep if i else 1.0,
                                               self._av("dy%d" % (i - 1)) if i else None,
                                               mb, 0 if mb is None else mb.shape[1])
                    if need_p and not direct:
                        k.mi_split_rows(dx, fan, B, fan, 0, dxp, None)
                elif self.planes and i == nh and i > 0 and h == 1 and fan % 16 == 0:
                    # the logits layer's matrix-vector data gradient, written straight as the planes the layer below
                    # reads (and in fp32 only if that layer's weight gradient still runs on fp32 operands)
                    mb = self._ws.get("mbits%d" % (i - 1))
                    mb = mb[:B * ((fan + 31) // 32)].view(B, (fan + 31) // 32) if mb is not None else None
                    k.mi_dense_bwd_data_vec_planes(dy, lddy, self.kernel(i), x, ldx, keep,
                                                   None if self._wgrad_planes_ok(B, i - 1) else dx, fan,
                                                   self._planes("dy%dp" % (i - 1), B, fan), B, fan, self._av("dy%d" % (i - 1)),
                                                   mb, 0 if mb is None else mb.shape[1])
                else:
                    k.mi_dense_bwd_data(dy, lddy, self.kernel(i), x if i else None, ldx, dx, fan, B, h, fan,
                                        keep if i else 1.0, self.act, None if self.planes else ga_d)
                    if self.planes and i == nh and i > 0:
                        # the logits layer's matrix-vector data gradient writes fp32: split it for the layer below
                        k.mi_split_rows(dx, fan, B, fan, 0, self._planes("dy%dp" % (i - 1), B, fan), self._av("dy%d" % (i - 1)))
                if data_first:
                    on_d_concat(dx)
                    weight_gradient()
                dy, lddy = dx, fan
            d_concat = dy
            for lo in range(0, len(batch or ()), _lib.MAX_WEIGHT_JOBS):
                part = batch[lo:lo + _lib.MAX_WEIGHT_JOBS]
                arr = (_lib.WgradJob * len(part))()
                for q, (i, fan, h, ga_w) in enumerate(part):
                    arr[q].X, arr[q].dY = self._pl["x%dp" % i].struct, self._pl["dy%dp" % i].struct
                    arr[q].dW, arr[q].db = ptr(self.kernel(i, self.d_grad)), ptr(self.bias(i, self.d_grad))
                    arr[q].N, arr[q].K, arr[q].amax = h, fan, ga_w
                bws = self._bytes("wgrad_batch_ws", k.query("mi_dense_bwd_weight_planes_batch_workspace_bytes", arr, len(part), B))
                k.mi_dense_bwd_weight_planes_batch(arr, len(part), B, bws, bws.numel())
        if self.n_numeric and self.raw_numeric:
            if self.use_linear:       # a raw numeric column owns no deep variable: only its linear_model weight
                ws = self._bytes("num_ws", k.query("mi_numeric_raw_bwd_workspace_bytes", B, self.n_numeric))
                k.mi_numeric_raw_bwd(c["x_num"], dlogit, B, self.n_numeric, self.d_grad[self.lin_num_off:], ws, ws.numel())
        elif self.n_numeric:
            ws = self._bytes("num_ws", k.query("mi_numeric_embed_bwd_workspace_bytes", B, self.n_numeric, self.E))
            k.mi_numeric_embed_bwd(c["x_num"], d_concat, self.D, c["concat"], self.D, self.F * self.E, c["sumv"],
                                   dlogit if self.use_mf else None, dlogit if self.use_linear else None, B,
                                   self.n_numeric, self.E, self.d_grad[self.num_emb_off:],
                                   self.d_grad[self.lin_num_off:] if self.use_linear else None, ws, ws.numel())
        return d_concat

    def _apply(self, uniq, seg, sorted_entry, num_uniq, n_max, d_rows, d_lin, fused=None, d_stride=0):
        """apply_gradients: dense Apply*, sparse apply on the unique rows, step += 1.
        fused = (d_concat, sumv, dlogit): single-GPU form, entry gradients rebuilt in the kernel.
        d_stride: d_rows / d_lin are views of ONE record buffer, d_stride floats per entry (the packed exchange); 0: two arrays."""
        k = self.k
        step = self.step + 1
        lr_t = self.sched.lr_t(step) if self.sched else 0.0
        hp = self.opt.hparams(lr_t)
        lin_lr_t = self.lin_sched.lr_t(step) if self.lin_sched else 0.0
        if self._frozen is not None:
            self.d_grad.index_fill_(0, self._frozen, 0.0)
        if self.lin_opt is None:
            if self.P:
                k.mi_dense_apply(self.dense, self.d_s0, self.d_s1, self.d_grad, self.P, hp)
            sparse_hp = [(True, True, hp)]
        else:
            lhp = self.lin_opt.hparams(lin_lr_t)
            if self.wide_off:
                k.mi_dense_apply(self.dense, self.d_s0, self.d_s1, self.d_grad, self.wide_off, hp)
            o = self.wide_off
            sl = lambda t: t[o:] if t is not None else None
            k.mi_dense_apply(self.dense[o:], sl(self.dl_s0), sl(self.dl_s1), self.d_grad[o:], self.P - o, lhp)
            sparse_hp = [(True, False, hp), (False, True, lhp)]
        if n_max > 0:
            for do_table, do_lin, h in sparse_hp:
                tb = self.table if (do_table and self.use_emb) else None
                lw = self.lin_w if (do_lin and self.use_linear) else None
                if tb is None and lw is None:
                    continue
                slots = (tb, self.t_s0 if tb is not None else None, self.t_s1 if tb is not None else None,
                         lw, self.l_s0 if lw is not None else None, self.l_s1 if lw is not None else None)
                if fused is not None:
                    d_concat, sumv, dlogit = fused
                    k.mi_sparse_apply_fused(*slots, self.last_step, uniq, seg, sorted_entry, num_uniq, n_max,
                                            d_concat if tb is not None else None, self.D,
                                            sumv if (tb is not None and self.use_mf) else None,
                                            dlogit if (tb is not None and self.use_mf) else None,
                                            dlogit if lw is not None else None, self.F, self.E, step, h, self.ls, self.ts)
                else:
                    k.mi_sparse_apply(*slots, self.last_step, uniq, seg, sorted_entry, num_uniq, n_max,
                                      d_rows if tb is not None else None, d_lin if lw is not None else None, self.E,
                                      step, h, self.ls, self.ts, d_stride)
        self.step = step

    # ------------------------------------------------------------------ replayable step (hipGraph)
    def graph_ok(self):
        """a captured step exists for this model: single GPU, at most ONE Adam schedule (the device-resident step state
        carries one lr_t)"""
        return (self.shard is None and self.device.type == "cuda" and
                not (self.sched is not None and self.lin_sched is not None and self.sched is not self.lin_sched))

    def graph_train_step(self, ids, labels, x_num=None):
        """train_step as ONE hipGraph launch (single GPU; numeric columns are a third input copy): the launch-bound small-batch
        configurations (trainers.deep_fm's defaults: B = 32, ~50 launches) pay one graph launch instead of ~50
        kernel launches.  The first call runs eagerly (it sizes the workspaces), the second captures, later
        calls copy the batch into the captured buffers and replay.  Per-step scalars — global step, Adam's lr_t,
        the dropout seeds — live in a device-resident step state that the captured mi_step_advance node
        advances (include/mi355x_rec.h), so a replay is bit-identical to the eager step it replaces."""
        if self.shard is not None or self.device.type != "cuda":
            raise NotImplementedError("graph_train_step: single-GPU models")
        if not self.graph_ok():
            raise NotImplementedError("graph_train_step: the device-resident step state carries ONE lr_t (two different Adams)")
        self._prep(ids, labels, x_num)
        # one captured graph per batch shape (an epoch's short last batch no longer evicts the full batch's graph: two
        # captures per epoch before), a handful at most
        graphs = self.__dict__.setdefault("_graphs", {})
        shape = tuple(ids.shape)
        g = self._graph = graphs.get(shape)
        if g is not None and g["gen"] != self._graph_gen():
            # A captured graph holds the raw addresses of the workspaces, the planes and the lr_t table.  Something
            # since the capture made one of them move (loss() / predict on a larger batch, a train_step of another
            # shape, layer_summaries' workspaces, a restored checkpoint far into a run): the old storage may already
            # belong to another tensor, so every graph is dropped and the step captured again (buffers only grow: the
            # new capture needs no sizing step).
            graphs.clear()
            g = self._graph = None
            self._warm_shapes().add(shape)
        if g is None:
            warm = self._warm_shapes()
            if shape not in warm:
                warm.add(shape)
                return self.train_step(ids, labels, x_num)              # sizes every workspace
            if len(graphs) >= self.GRAPH_SHAPES_MAX:
                graphs.pop(next(iter(graphs)))
            self._top_step = None
            g = self._graph = graphs[shape] = self._capture(ids, labels, x_num)
            g["top"] = getattr(self, "_top_step", None) == self.step - 1     # (the captured step keeps the last hidden layer on the chip)
            return g["loss"], g["logits"]
        if self._gsched() is not None and self.step + 2 >= len(self._gsched().host):
            graphs.clear()                                              # the lr_t table has to grow: capture again
            self._graph = None
            return self.graph_train_step(ids, labels, x_num)
        if g["dev_step"] != self.step:                                  # eager steps ran in between: resync
            self._write_step_state(g["state"])
            g["dev_step"] = self.step
        g["ids"].copy_(ids)
        g["y"].copy_(labels)
        if x_num is not None:
            g["x"].copy_(x_num)
        g["graph"].replay()
        self.step += 1
        g["dev_step"] = self.step
        # (layer_summaries: did the step that just ran leave the last hidden layer's output in memory?  A replay runs what
        # was captured, whatever summaries_next says now)
        self._top_step = self.step - 1 if g.get("top") else None
        self._acts_in_planes = set(g.get("acts_in_planes", self._acts_in_planes))
        return g["loss"], g["logits"]

    def _warm_shapes(self):
        """batch shapes whose sizing step (an eager step before the capture) has run"""
        if not isinstance(self.__dict__.get("_graph_warm"), set):
            self._graph_warm = set()
        return self._graph_warm

    def drop_graphs(self):
        """forget every captured step (the next graph_train_step of a shape captures again, without a sizing step)"""
        self._graph = None
        self.__dict__.setdefault("_graphs", {}).clear()

    def _gsched(self):
        """the one Adam schedule a captured step reads lr_t from"""
        return self.sched if self.sched is not None else self.lin_sched

    def _graph_gen(self):
        return (self._alloc_gen, self._gsched().gen if self._gsched() is not None else 0)

    def _write_step_state(self, state):
        blob = np.zeros(1, np.dtype([("step", np.int32), ("lr_t", np.float32), ("seed_term", np.uint64)]))
        blob["step"] = self.step
        state.copy_(torch.from_numpy(blob.view(np.uint8)))

    def _capture(self, ids, labels, x_num=None):
        if self._gsched() is not None:                                  # the lr_t table must not move under the graph
            self._gsched().lr_t(self.step + (1 << 20))
        ps, self._presorted = getattr(self, "_presorted", None), None
        if ps is not None:                                              # (nothing of another stream inside the capture)
            torch.cuda.current_stream().wait_stream(ps["stream"])
        state = torch.zeros(16, dtype=torch.uint8, device=self.device)
        self._write_step_state(state)
        g_ids, g_y = ids.clone(), labels.clone()
        g_x = None if x_num is None else x_num.clone()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        self.k.query("mi_set_step_state", state.data_ptr())
        self._capturing = True
        try:
            with torch.cuda.graph(graph):
                self.k.mi_step_advance(state, self._gsched().table if self._gsched() is not None else None)
                loss, logits = self.train_step(g_ids, g_y, g_x)
        finally:
            self._capturing = False
            self.k.query("mi_set_step_state", None)
        graph.replay()                                                  # capture records, this executes the step
        return {"graph": graph, "state": state, "ids": g_ids, "y": g_y, "x": g_x, "loss": loss, "logits": logits,
                "shape": tuple(ids.shape), "dev_step": self.step, "gen": self._graph_gen(),
                "acts_in_planes": tuple(self._acts_in_planes)}

    def layer_summaries(self):
        """What the reference's layer_summary calls record (model_utils.py:4-6 at deep_fm.py:43,89,105,
        110,115): fraction of zeros (+ min / max / mean) of the linear, mf and dnn logits, of every hidden
        layer's output and of the summed logits — computed on the activations the last step left in the
        workspace (mi_layer_stats), one small host copy.  Keys follow the reference's scopes."""
        k = self.k
        B = getattr(self, "_last_B", 0)
        if not B:
            return {}
        named = []
        if self.use_linear and "lin" in self._ws:
            named.append(("linear/logits", self._ws["lin"][:B]))
        if self.use_mf and "fm" in self._ws:
            named.append(("mf/logits", self._ws["fm"][:B]))
        if self.use_dnn:
            for i, (_, _, _, h) in enumerate(self.layers):
                a = self._ws.get("act%d" % i)
                if i == len(self.layers) - 2 and getattr(self, "_top_step", None) == self.step - 1:
                    continue      # (the last step kept this layer's output on the chip: _top_fusable — summaries_next was not set)
                if a is not None and i in self._acts_in_planes:      # the last step wrote planes only: merge them
                    k.mi_merge_rows(self._pl["x%dp" % (i + 1)].struct, B, h, a, h)
                if a is not None:
                    named.append(("dnn/hiddenlayer_%d" % i if i < len(self.layers) - 1 else "dnn/logits", a[:B * h]))
        if "logits" in self._ws:
            named.append(("logits", self._ws["logits"][:B]))
        out = self._buf("layer_stats", (len(named), 4))
        hist = getattr(k, "mi_layer_histogram", None) is not None and self.device.type == "cuda"
        if hist:
            from .metrics import histogram_limits, histogram_proto
            lim = self._ws.get("hist_limits")
            if lim is None:
                lim = self._ws["hist_limits"] = torch.from_numpy(histogram_limits()).to(self.device)
            counts = torch.zeros(len(named), lim.numel() + 1, dtype=torch.int64, device=self.device)
            sums = torch.zeros(len(named), 2, dtype=torch.float64, device=self.device)
        for j, (_, x) in enumerate(named):
            ws = self._bytes("layer_stats_ws", k.query("mi_layer_stats_workspace_bytes", x.numel()))
            k.mi_layer_stats(x, x.numel(), out[j], ws, ws.numel())
            if hist:
                k.mi_layer_histogram(x, x.numel(), lim, lim.numel(), counts[j], sums[j])
        vals = out.cpu().tolist()
        res = {n: dict(zip(("fraction_of_zero_values", "min", "max", "mean"), v)) for (n, _), v in zip(named, vals)}
        if hist:        # tf.summary.histogram("activation", value): the second half of layer_summary
            cn, sm, ln = counts.cpu().numpy(), sums.cpu().numpy(), lim.cpu().numpy()
            for j, (n, _) in enumerate(named):
                res[n]["activation"] = histogram_proto(ln, cn[j], sm[j], res[n]["min"], res[n]["max"])
        return res

    # ------------------------------------------------------------------ checkpoint
    _STATE_KEYS = ("dense", "d_s0", "d_s1", "table", "lin_w", "t_s0", "t_s1", "l_s0", "l_s1", "last_step",
                   "dl_s0", "dl_s1")
    STATE_FORMAT = 3       # 3: carries the layout table below (rounds 1-2 wrote bare tensors; the flat buffer's layout changed between them)

    def _layout(self):
        """What a checkpoint must agree on before its flat buffers may be copied in: the model's shape and where each
        dense variable sits in `dense` (name -> [offset, rows, cols]).  The reference's TF checkpoints are name- and
        shape-checked (conf_utils.py:6-10); a flat buffer is only as safe as this table."""
        seg = {}
        for i, (k_off, b_off, fan, h) in enumerate(self.layers):
            seg["kernel_%d" % i] = [k_off, fan, h]
            seg["bias_%d" % i] = [b_off, h, 1]
        if self.num_emb_off is not None:
            seg["numeric_embeddings"] = [self.num_emb_off, self.n_numeric, self.E]
        seg["linear_bias"] = [self.lin_bias_off, 1, 1]
        if self.lin_num_off is not None:
            seg["linear_numeric_weights"] = [self.lin_num_off, self.n_numeric, 1]
        world, rank = (1, 0) if self.shard is None else (self.shard.world, self.shard.rank)
        return {"vocab_sizes": list(self.vocab_sizes), "embedding_size": self.E, "hidden_units": list(self.hidden),
                "n_numeric": self.n_numeric, "numeric": self.numeric, "P": self.P, "R": self.R, "world": world, "rank": rank,
                "use": [self.use_linear, self.use_mf, self.use_dnn], "optimizer": self.opt.name,
                "linear_optimizer": None if self.lin_opt is None else self.lin_opt.name,
                "lin_record_stride": self.ls, "segments": seg, "field_dims": self.field_dims, "wide_fields": self.wide_fields,
                "deep_numeric": self.deep_numeric, "wide_numeric": self.wide_numeric}

    def state_dict(self):
        """Everything needed to resume (reference: Estimator checkpoints, conf_utils.py:6-10)."""
        self.finalize_rows()
        sd = {"step": self.step, "format": self.STATE_FORMAT, "layout": self._layout()}
        for key in self._STATE_KEYS:
            v = getattr(self, key, None)
            if v is not None:
                sd[key] = v.detach().to("cpu", copy=True).contiguous()    # (views of one record array: independent copies)
        if self.wide_fields is not None and "lin_w" in sd:
            # a column outside linear_feature_columns owns no linear weight in the reference model; one apply kernel serves
            # every row of a batch and does write those slots (never read): a checkpoint carries them at their initial values
            a, b = (self.lin_opt or self.opt).slot_init
            for f, on in enumerate(self.wide_fields):
                if not on:
                    lo, hi = self._field_rows(f)
                    for key, fill in (("lin_w", 0.0), ("l_s0", a), ("l_s1", b)):
                        if key in sd and fill is not None:
                            sd[key][lo:hi] = fill
        return sd

    def load_state_dict(self, sd):
        """Verifies format, layout and every tensor's shape BEFORE anything is copied: a checkpoint of another model,
        of another hidden / embedding size with the same parameter count, or of an older flat-buffer layout is an
        error, never a silent permutation of the variables."""
        if sd.get("format") != self.STATE_FORMAT:
            raise ValueError("checkpoint format %r, this build reads %d: the flat dense buffer's layout is only defined "
                             "by the layout table newer checkpoints carry (re-export the variables by name: "
                             "mi355x_rec/tf_names.py)" % (sd.get("format"), self.STATE_FORMAT))
        mine, theirs = self._layout(), sd["layout"]
        canon = lambda v: json.loads(json.dumps(v))                   # (tuples / lists alike)
        bad = sorted(k for k in set(mine) | set(theirs) if canon(mine.get(k)) != canon(theirs.get(k)))
        if bad:
            raise ValueError("checkpoint does not fit this model: %s" % "; ".join(
                "%s = %r in the checkpoint, %r here" % (k, theirs.get(k), mine.get(k)) for k in bad))
        have = {k for k in self._STATE_KEYS if getattr(self, k, None) is not None}
        given = {k for k in sd if k not in ("step", "format", "layout")}
        if have != given:
            raise ValueError("checkpoint tensors %s, the model has %s" % (sorted(given), sorted(have)))
        for key in given:
            dst, v = getattr(self, key), sd[key]
            if tuple(v.shape) != tuple(dst.shape) or v.dtype != dst.dtype:
                raise ValueError("checkpoint tensor %r is %s %s, the model's is %s %s" %
                                 (key, tuple(v.shape), v.dtype, tuple(dst.shape), dst.dtype))
        for key in given:
            getattr(self, key).copy_(sd[key].to(self.device))
        self.step = int(sd["step"])
        self._final_step = self.step
        self._presorted = None           # (a sort of a batch announced before the restore: dropped)
        self.drop_graphs()               # (captured steps are re-captured against the restored state)
