"""The reference graph restated on PyTorch-CPU, multi-threaded.  Oracle: tests + bench.py's cpu_baseline only.

Same arithmetic as ``oracle/deepfm.py`` (which follows ``trainers/deep_fm.py:36-125`` line by line) for the
DeepFM configuration bench.py measures — categorical columns only, Adam — but on torch CPU tensors, so
that every elementwise pass and reduction runs on all host threads the way TensorFlow's Eigen thread pool
would run them, at the FULL vocabulary.  Deliberately un-fused, as the TF graph
executes: one table per field, a materialised [B, d, E] tensor, separate sum->square / square->sum
reductions, dense -> relu layers, and TF-form Adam whose sparse update sweeps EVERY row of every table
(SURVEY Appendix A.6): m *= b1; m[idx] += (1-b1) g; v *= b2; v[idx] += (1-b2) g^2; w -= lr_t m / (sqrt(v) + eps).
``tests/test_oracle.py`` checks it against the numpy oracle.  PARITY UNPINNED (see oracle/__init__.py)."""
import math

import torch


class State:
    def __init__(self, vocab_sizes, E, hidden, seed=0, lin_scale=1e-3, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        g = torch.Generator().manual_seed(seed)
        self.E, self.hidden = E, list(hidden)
        # (N(0, 1/sqrt(E)) clamped at 2 sigma, drawn for 65,536 rows and tiled: torch's CPU generator is single-
        # threaded, 1.7 G fresh draws take two minutes, and the values do not change what a step costs)
        def tiled(v, draw):
            base = draw(min(v, 1 << 16))
            return base.repeat((v + len(base) - 1) // len(base), *([1] * (base.dim() - 1)))[:v].contiguous()
        sd = 1 / math.sqrt(E)
        self.emb = [tiled(v, lambda n: torch.empty(n, E).normal_(0.0, sd, generator=g).clamp_(-2 * sd, 2 * sd)) for v in vocab_sizes]
        self.lin_w = [tiled(v, lambda n: torch.randn(n, generator=g) * lin_scale) for v in vocab_sizes]
        self.lin_bias = torch.zeros(1)
        self.mlp = []
        fan = len(vocab_sizes) * E
        for h in self.hidden + [1]:
            lim = math.sqrt(6.0 / (fan + h))
            self.mlp.append((torch.empty(fan, h).uniform_(-lim, lim, generator=g), torch.zeros(h)))
            fan = h
        z = torch.zeros_like
        self.slots = {"emb": [(z(a), z(a)) for a in self.emb], "lin": [(z(a), z(a)) for a in self.lin_w],
                      "bias": (z(self.lin_bias), z(self.lin_bias)), "mlp": [((z(k), z(k)), (z(b), z(b))) for k, b in self.mlp]}
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.b1p, self.b2p = torch.tensor(beta1, dtype=torch.float32), torch.tensor(beta2, dtype=torch.float32)

    def load_numpy(self, p):
        """take the variables of an oracle.deepfm.Params (tests)"""
        t = lambda a: torch.from_numpy(a.copy())
        self.emb = [t(a) for a in p.emb]
        self.lin_w = [t(a) for a in p.lin_w]
        self.lin_bias = t(p.lin_bias)
        self.mlp = [(t(k), t(b)) for k, b in p.mlp]
        z = torch.zeros_like
        self.slots = {"emb": [(z(a), z(a)) for a in self.emb], "lin": [(z(a), z(a)) for a in self.lin_w],
                      "bias": (z(self.lin_bias), z(self.lin_bias)), "mlp": [((z(k), z(k)), (z(b), z(b))) for k, b in self.mlp]}


def _dense_adam(st, w, m, v, g, lr_t):
    m.add_((g - m) * (1 - st.b1))                                   # fused ApplyAdam
    v.add_((g * g - v) * (1 - st.b2))
    w.sub_((m * lr_t) / (v.sqrt() + st.eps))


def _sparse_adam(st, w, m, v, idx, g, lr_t):
    """_apply_sparse_shared: duplicates summed first, then EVERY row decays and moves"""
    uniq, inv = torch.unique(idx, return_inverse=True)
    gs = torch.zeros((len(uniq),) + g.shape[1:], dtype=g.dtype).index_add_(0, inv, g)
    m.mul_(st.b1)
    m.index_add_(0, uniq, gs * (1 - st.b1))
    v.mul_(st.b2)
    v.index_add_(0, uniq, (gs * gs) * (1 - st.b2))
    w.sub_((lr_t * m) / (v.sqrt() + st.eps))


def train_step(st, ids, labels):
    """one optimizer.minimize(loss) of the DeepFM graph; ids [B, F] int64, labels [B] float32.  Returns (loss, logits)."""
    B, F = ids.shape
    E = st.E
    # forward (deep_fm.py:36-115)
    lin = torch.zeros(B)
    for f in range(F):
        lin = lin + st.lin_w[f][ids[:, f]]
    lin = lin + st.lin_bias
    parts = [st.emb[f][ids[:, f]] for f in range(F)]
    concat = torch.cat(parts, 1)
    mat = concat.view(B, F, E)
    s = mat.sum(1)
    fm = 0.5 * ((s * s) - (mat * mat).sum(1)).sum(1)
    net, acts = concat, []
    for k, b in st.mlp[:-1]:
        net = torch.relu(net @ k + b)
        acts.append(net)
    k, b = st.mlp[-1]
    dnn = (net @ k + b)[:, 0]
    logits = lin + fm + dnn
    # head (deep_fm.py:118-125): mean sigmoid cross-entropy
    per = torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))
    loss = per.mean()
    dl = (torch.sigmoid(logits) - labels) / B
    # backward
    grads = []
    d_net = dl[:, None] * st.mlp[-1][0][:, 0][None, :]
    grads.append((acts[-1].t() @ dl[:, None], dl.sum(0, keepdim=True)))
    for i in range(len(st.mlp) - 2, -1, -1):
        d_pre = d_net * (acts[i] > 0)
        inp = acts[i - 1] if i else concat
        grads.append((inp.t() @ d_pre, d_pre.sum(0)))
        d_net = d_pre @ st.mlp[i][0].t()
    grads.reverse()
    g_v = d_net.view(B, F, E) + dl[:, None, None] * (s[:, None, :] - mat)
    # apply (dense: ApplyAdam; sparse: whole-table sweep)
    lr_t = st.lr * torch.sqrt(1 - st.b2p) / (1 - st.b1p)
    for (kk, bb), (gk, gb), ((mk, vk), (mb, vb)) in zip(st.mlp, grads, st.slots["mlp"]):
        _dense_adam(st, kk, mk, vk, gk, lr_t)
        _dense_adam(st, bb, mb, vb, gb.reshape(bb.shape), lr_t)
    _dense_adam(st, st.lin_bias, *st.slots["bias"], dl.sum(0, keepdim=True), lr_t)
    for f in range(F):
        _sparse_adam(st, st.emb[f], *st.slots["emb"][f], ids[:, f], g_v[:, f, :], lr_t)
        _sparse_adam(st, st.lin_w[f], *st.slots["lin"][f], ids[:, f], dl, lr_t)
    st.b1p = st.b1p * st.b1
    st.b2p = st.b2p * st.b2
    return loss, logits
