"""numpy stand-ins for the device entry points of libmi355x_rec.so — TEST INFRASTRUCTURE ONLY.
(The wide part's strided state views arrive as strided numpy views: lin_stride needs no handling here.)

The CPU (gloo, world_size 2) tests hand an instance to ``DeepFM(_kernels=...)`` so that the REAL
host orchestration (engine.py) and the REAL multi-rank exchange plumbing (parallel.py) run without a
GPU.  Every method restates the contract documented in include/mi355x_rec.h on CPU torch tensors
(in place, through ``.numpy()`` views), using the oracle's update rules for the optimizers.  Nothing
under recommender-tensorflow_amd/ imports this file.
"""
import numpy as np
import torch

from oracle import optimizers as OO
from tests.util import dropout_mask

_NAMES = {0: "Adam", 1: "Adagrad", 2: "Ftrl", 3: "RMSProp", 4: "SGD"}


def _np(t):
    return None if t is None else t.numpy()


def _hyper(hp):
    return OO.Hyper(_NAMES[hp.kind], lr=np.float32(hp.lr), beta1=np.float32(hp.beta1), beta2=np.float32(hp.beta2),
                    epsilon=np.float32(hp.epsilon), decay=np.float32(hp.decay), momentum=np.float32(hp.momentum),
                    lr_power=-0.5, l1=np.float32(hp.l1), l2=np.float32(hp.l2))


class NumpyKernels:
    timers = None

    def query(self, name, *args):
        return 256

    # ---- ids / routing -----------------------------------------------------------------
    def mi_global_rows(self, ids, field_off, B, F, rows):
        _np(rows)[:] = (_np(ids).astype(np.int64) + _np(field_off)[None, :]).reshape(-1)

    def mi_shard_keys(self, rows, n, world, entries_per_chunk, rows_per_rank, self_rank, keys):
        r = _np(rows)[:n].astype(np.int64)
        chunk = (np.arange(n) // entries_per_chunk) if entries_per_chunk > 0 else 0
        o = r % world
        if self_rank >= 0:
            o = np.where(o == self_rank, world - 1, np.where(o > self_rank, o - 1, o))
        _np(keys)[:n] = (chunk * world + o) * rows_per_rank + r // world

    def mi_route_requests(self, uniq, num_uniq, n_max, rows_per_rank, n_groups, send_rows, counts):
        U = int(_np(num_uniq)[0])
        key = _np(uniq)[:U].astype(np.int64)
        _np(send_rows)[:U] = key % rows_per_rank
        _np(counts)[:n_groups] = np.bincount(key // rows_per_rank, minlength=n_groups)

    def mi_segment_slots(self, seg, sorted_entry, num_uniq, n, slot):
        U = int(_np(num_uniq)[0])
        sg, se = _np(seg), _np(sorted_entry)
        for u in range(U):
            _np(slot)[se[sg[u]:sg[u + 1]]] = u

    def mi_entry_grads_segsum(self, rows, seg, sorted_entry, u_begin, u_count, d_concat, ldd, sumv, dlf, dll, b0, F, E,
                              out_rows, out_lin, out_row0=0, rows_stride=0, out_stride=0):
        sg, se = _np(seg), _np(sorted_entry)
        for u in range(u_begin, u_begin + u_count):
            g = np.zeros(E, np.float32)
            gl = np.float32(0)
            for e in se[sg[u]:sg[u + 1]]:
                b, f = int(e) // F - b0, int(e) % F
                if out_rows is not None:
                    v = np.zeros(E, np.float32)
                    if d_concat is not None:
                        v = v + _np(d_concat)[b, f * E:(f + 1) * E]
                    if dlf is not None:
                        v = v + _np(dlf)[b] * (_np(sumv)[b] - _np(rows)[u])
                    g = g + v
                if out_lin is not None:
                    gl = gl + _np(dll)[b]
            if out_rows is not None:
                _np(out_rows)[u - out_row0] = g
            if out_lin is not None:
                _np(out_lin)[u - out_row0] = gl

    def mi_axpy(self, y, x, n, alpha):
        _np(y)[:n] += np.float32(alpha) * _np(x)[:n]

    def mi_gather_u32(self, src, idx, n, out):
        _np(out)[:n] = _np(src)[_np(idx)[:n]]

    def mi_sort_unique_rows(self, rows, n, total, sorted_entry, uniq, seg, num_uniq, ws, wsb):
        r = _np(rows)[:n]
        order = np.argsort(r, kind="stable").astype(np.int32)
        sr = r[order]
        starts = np.flatnonzero(np.r_[True, sr[1:] != sr[:-1]]).astype(np.int32)
        U = len(starts)
        _np(sorted_entry)[:n] = order
        if uniq is None:
            return
        _np(uniq)[:U] = sr[starts]
        _np(seg)[:U] = starts
        _np(seg)[U] = n
        _np(num_uniq)[0] = U

    def mi_sort_unique_rows_slots(self, rows, n, total, sorted_entry, uniq, seg, num_uniq, slot, ws, wsb):
        self.mi_sort_unique_rows(rows, n, total, sorted_entry, uniq, seg, num_uniq, ws, wsb)
        self.mi_segment_slots(seg, sorted_entry, num_uniq, n, slot)

    # ---- embedding side ------------------------------------------------------------------
    def mi_embed_fm_linear_fwd(self, table, lin_w, field_off, ids, B, F, E, concat, ld, sumv, fm, lin, amax=None, ls=1, ts=0):
        rows = _np(ids).astype(np.int64) + _np(field_off)[None, :]
        if table is not None:
            v = _np(table)[rows]                               # [B,F,E]
            if concat is not None:
                _np(concat)[:, :F * E] = v.reshape(B, F * E)
            s = v.sum(1)
            if sumv is not None:
                _np(sumv)[:] = s
            if fm is not None:
                _np(fm)[:] = np.float32(0.5) * (s * s - (v * v).sum(1)).sum(1)
        if lin is not None:
            _np(lin)[:] = _np(lin_w)[rows].sum(1)

    def mi_gather_rows(self, table, lin_w, rows, n, E, out_rows, out_lin, ls=1, ts=0, out_stride=0):
        r = _np(rows)[:n]
        if table is not None:
            _np(out_rows)[:n] = _np(table)[r]
        if lin_w is not None and out_lin is not None:
            _np(out_lin)[:n] = _np(lin_w)[r]

    def mi_numeric_embed_fwd(self, x, V, w_num, B, nd, E, concat, ld, col0, sumv, fm, lin):
        xv, Vv = _np(x), _np(V)
        r = xv[:, :, None] * Vv[None, :, :]
        _np(concat)[:, col0:col0 + nd * E] = r.reshape(B, nd * E)
        if sumv is not None:
            sc = _np(sumv).copy()
            st = sc + r.sum(1)
            _np(sumv)[:] = st
            if fm is not None:
                _np(fm)[:] += np.float32(0.5) * ((st * st - sc * sc) - (r * r).sum(1)).sum(1)
        if lin is not None and w_num is not None:
            _np(lin)[:] += xv @ _np(w_num)

    def mi_numeric_raw_fwd(self, x, w_num, B, nd, concat, ld, col0, ncols, lin):
        xv = _np(x)
        if concat is not None:
            cc = _np(concat)
            cc[:, col0:col0 + ncols] = 0
            cc[:, col0:col0 + nd] = xv
        if lin is not None and w_num is not None:
            acc = _np(lin).copy()
            for j in range(nd):
                acc = acc + xv[:, j] * _np(w_num)[j]
            _np(lin)[:] = acc

    def mi_numeric_raw_bwd(self, x, dll, B, nd, dw, ws, wsb):
        _np(dw)[:nd] = (_np(dll)[:, None] * _np(x)).sum(0)

    def mi_embed_fm_linear_bwd(self, d_concat, lddc, concat, ldc, rows, sumv, dlf, dll, pos, B, F, E, d_rows, d_lin):
        p = np.arange(B * F) if pos is None else _np(pos).reshape(-1)[:B * F].astype(np.int64)
        if d_rows is not None:
            g = np.zeros((B, F, E), np.float32)
            if d_concat is not None:
                g += _np(d_concat)[:, :F * E].reshape(B, F, E)
            if dlf is not None:
                v = _np(rows)[p].reshape(B, F, E) if rows is not None else _np(concat)[:, :F * E].reshape(B, F, E)
                g += _np(dlf)[:, None, None] * (_np(sumv)[:, None, :] - v)
            _np(d_rows)[p] = g.reshape(B * F, E)
        if d_lin is not None:
            _np(d_lin)[p] = np.repeat(_np(dll), F)

    def mi_numeric_embed_bwd(self, x, d_concat, lddc, concat, ldc, col0, sumv, dlf, dll, B, nd, E, dV, dw, ws, wsb):
        g = np.zeros((B, nd, E), np.float32)
        if d_concat is not None:
            g += _np(d_concat)[:, col0:col0 + nd * E].reshape(B, nd, E)
        if dlf is not None:
            v = _np(concat)[:, col0:col0 + nd * E].reshape(B, nd, E)
            g += _np(dlf)[:, None, None] * (_np(sumv)[:, None, :] - v)
        _np(dV)[:nd * E] = np.einsum("bj,bje->je", _np(x), g).reshape(-1)
        if dw is not None:
            _np(dw)[:nd] = (_np(dll)[:, None] * _np(x)).sum(0) if dll is not None else 0

    # ---- MLP ----------------------------------------------------------------------------------
    def mi_absmax(self, x, n, out):
        pass          # abs-max vectors only steer the HIP kernels' fp16 scales

    def mi_dense_fwd(self, X, ldx, W, bias, Y, ldy, M, N, K, relu, keep, seed, amax=None):
        y = _np(X)[:, :K] @ _np(W) + _np(bias)
        y = {0: lambda v: v, 1: lambda v: np.maximum(v, 0), 2: lambda v: 1 / (1 + np.exp(-v)), 3: np.tanh}[int(relu)](y).astype(np.float32)
        if keep < 1.0:
            y = (y / np.float32(keep)) * dropout_mask(seed, M, N, keep)
        _np(Y)[:, :N] = y

    def mi_dense_bwd_data(self, dY, lddy, W, Xact, ldxa, dX, lddx, M, N, K, keep, act=1, amax=None):
        dy = _np(dY).reshape(M, -1)[:, :N]
        g = dy @ _np(W).T
        if Xact is not None:
            xa = _np(Xact)[:, :K]
            if act == 1:
                g = (g * (xa > 0)) / np.float32(keep)
            else:
                y = xa * np.float32(keep)
                d = {0: np.ones_like(y), 2: y * (1 - y), 3: 1 - y * y}[int(act)]
                g = np.where((keep < 1) & (xa == 0), 0, (g / np.float32(keep)) * d).astype(np.float32)
        _np(dX)[:, :K] = g

    def mi_dense_bwd_weight(self, X, ldx, dY, lddy, dW, db, M, N, K, ws, wsb, amax=None):
        dy = _np(dY).reshape(M, -1)[:, :N]
        _np(dW)[:] = _np(X)[:, :K].T @ dy
        if db is not None:
            _np(db)[:N] = dy.sum(0)

    def mi_sigmoid_ce_head(self, lin, lin_bias, fm, dnn, labels, B, scale, logits, loss, dlogit, dsum, ws, wsb):
        x = np.zeros(B, np.float32)
        if lin is not None:
            x = x + (_np(lin) + _np(lin_bias)[0])
        if fm is not None:
            x = x + _np(fm)
        if dnn is not None:
            x = x + _np(dnn)
        _np(logits)[:] = x
        if labels is not None:
            y = _np(labels).astype(np.float32)
            per = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
            if loss is not None:
                _np(loss)[0] = (per * np.float32(scale)).sum(dtype=np.float32)
            if dlogit is not None:
                e = np.exp(-np.abs(x))
                sig = np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(np.float32)
                d = (sig - y) * np.float32(scale)
                _np(dlogit)[:] = d
                if dsum is not None:
                    _np(dsum)[0] = d.sum(dtype=np.float32)

    # ---- optimizers ---------------------------------------------------------------------------
    def mi_dense_apply(self, param, s0, s1, grad, n, hp):
        h = _hyper(hp)
        z = np.zeros(n, np.float32)
        OO.dense_apply(h, _np(param)[:n], _np(s0)[:n] if s0 is not None else z, _np(s1)[:n] if s1 is not None else z,
                       _np(grad)[:n], np.float32(hp.lr_t))

    def mi_catchup_gap_keys(self, uniq, num_uniq, last_step, n_max, step_to, keys, ls=1):
        U = int(_np(num_uniq)[0])
        k = np.full(n_max, 63, np.int32)
        ls = _np(last_step)[_np(uniq)[:U]]
        k[:U] = np.where((ls > 0) & (ls < step_to), np.minimum(step_to - ls, 62), 0)
        _np(keys)[:n_max] = k

    def mi_catchup_rows_by_gap(self, uniq, num_uniq, last_step, n_max, step_to, ls, rows_out, ws, ws_bytes):
        keys = torch.empty(n_max, dtype=torch.int32)
        self.mi_catchup_gap_keys(uniq, num_uniq, last_step, n_max, step_to, keys, ls)
        _np(rows_out)[:n_max] = _np(uniq)[:n_max][np.argsort(_np(keys), kind="stable")]

    def mi_sparse_catchup(self, table, tm, tv, lin_w, lm, lv, last_step, uniq, num_uniq, n_max, E, step_to, lr_table,
                          b1, b2, eps, flags=0, ls=1, ts=0):
        defer = bool(flags & 1) and uniq is not None          # (flag 2, the bounded-error replay: the exact sweep stands in)
        rows = np.arange(n_max) if uniq is None else _np(uniq)[:int(_np(num_uniq)[0])]
        ls = _np(last_step)
        lr = _np(lr_table)
        b1, b2, eps = np.float32(b1), np.float32(b2), np.float32(eps)
        for r in rows:
            if ls[r] >= step_to:
                continue
            if ls[r] > 0:
                for w, m, v in ((table, tm, tv), (lin_w, lm, lv)):
                    if w is None:
                        continue
                    W, M, V = _np(w), _np(m), _np(v)
                    mr, vr = M[r].copy(), V[r].copy()
                    for s in range(ls[r] + 1, step_to + 1):
                        mr = mr * b1
                        vr = vr * b2
                        W[r] = W[r] - (lr[s] * mr) / (np.sqrt(vr) + eps)
                    if not defer:
                        M[r], V[r] = mr, vr
            if not defer and not (flags & 4):
                ls[r] = step_to

    def mi_sparse_apply(self, table, t0, t1, lin_w, l0, l1, last_step, uniq, seg, sorted_entry, num_uniq, n_max,
                        d_rows, d_lin, E, step, hp, ls=1, ts=0, grad_stride=0):
        h = _hyper(hp)
        U = int(_np(num_uniq)[0])
        rows = _np(uniq)[:U].astype(np.int64)
        sg, se = _np(seg), _np(sorted_entry)
        from mi355x_rec.engine import OptimizerSpec  # noqa: F401
        for w, a, b, g, width in ((table, t0, t1, d_rows, E), (lin_w, l0, l1, d_lin, 1)):
            if w is None:
                continue
            W = _np(w).reshape(-1, width)
            A = _np(a).reshape(-1, width) if a is not None else np.zeros_like(W)
            Bm = _np(b).reshape(-1, width) if b is not None else np.zeros_like(W)
            G = _np(g).reshape(-1, width)
            gsum = np.zeros((U, width), np.float32)
            for u in range(U):
                for kk in range(sg[u], sg[u + 1]):
                    gsum[u] += G[se[kk]]
            wv, av, bv = W[rows], A[rows], Bm[rows]
            if h.name == "Adam":
                b1, b2, eps = np.float32(h.beta1), np.float32(h.beta2), np.float32(h.epsilon)
                if last_step is not None:                       # deferred decay of the steps the row sat out
                    ls = _np(last_step)[rows]
                    missed = np.where(ls > 0, np.maximum(0, step - 1 - ls), 0)
                    for j in range(int(missed.max()) if len(missed) else 0):
                        on = (missed > j)[:, None]
                        av = np.where(on, av * b1, av)
                        bv = np.where(on, bv * b2, bv)
                av = av * b1 + gsum * (np.float32(1) - b1)
                bv = bv * b2 + (gsum * gsum) * (np.float32(1) - b2)
                wv = wv - (np.float32(hp.lr_t) * av) / (np.sqrt(bv) + eps)
            else:
                OO.dense_apply(h, wv, av, bv, gsum, None)
            W[rows] = wv
            if a is not None:
                A[rows] = av
            if b is not None:
                Bm[rows] = bv
        if last_step is not None:
            _np(last_step)[rows] = step

    # ---- gathered / fused single-GPU forms --------------------------------------------------------
    def _concat(self, table, field_off, ids, F, E):
        rows = _np(ids).astype(np.int64) + _np(field_off)[None, :]
        return _np(table)[rows].reshape(len(rows), F * E)

    def mi_dense_fwd_gathered(self, table, field_off, ids, F, E, W, bias, Y, ldy, M, N, relu, keep, seed, amax=None, ts=0):
        X = torch.from_numpy(self._concat(table, field_off, ids, F, E))
        self.mi_dense_fwd(X, F * E, W, bias, Y, ldy, M, N, F * E, relu, keep, seed)

    def mi_dense_bwd_weight_gathered(self, table, field_off, ids, F, E, dY, lddy, dW, db, M, N, ws, wsb, amax=None, ts=0):
        X = torch.from_numpy(self._concat(table, field_off, ids, F, E))
        self.mi_dense_bwd_weight(X, F * E, dY, lddy, dW, db, M, N, F * E, ws, wsb)

    def mi_sparse_apply_fused(self, table, t0, t1, lin_w, l0, l1, last_step, uniq, seg, sorted_entry, num_uniq,
                              n_max, d_concat, ldd, sumv, dlf, dll, F, E, step, hp, ls=1, ts=0):
        n = int(_np(seg)[int(_np(num_uniq)[0])])
        e = np.arange(n)
        b, f = e // F, e % F
        d_rows = d_lin = None
        if table is not None:
            g = np.zeros((n, E), np.float32)
            if d_concat is not None:
                g += _np(d_concat)[:, :F * E].reshape(-1, F, E)[b, f]
            if dlf is not None:
                # w = the row as it is before this update
                U = int(_np(num_uniq)[0])
                row_of = np.empty(n, np.int64)
                sg, se = _np(seg), _np(sorted_entry)
                for u in range(U):
                    row_of[se[sg[u]:sg[u + 1]]] = _np(uniq)[u]
                g += _np(dlf)[b][:, None] * (_np(sumv)[b] - _np(table)[row_of])
            d_rows = torch.from_numpy(g)
        if lin_w is not None:
            d_lin = torch.from_numpy(_np(dll)[b].astype(np.float32))
        self.mi_sparse_apply(table, t0, t1, lin_w, l0, l1, last_step, uniq, seg, sorted_entry, num_uniq, n_max,
                             d_rows, d_lin, E, step, hp)

    # ---- eval counters --------------------------------------------------------------------------
    def mi_binary_predictions(self, logits, labels, B, logistic, probabilities, class_ids, unreduced_loss):
        from oracle import deepfm as O
        x = _np(logits)[:B]
        sig = O.predictions(x)["logistic"]
        if logistic is not None:
            _np(logistic).reshape(-1)[:B] = sig
        if probabilities is not None:
            _np(probabilities).reshape(-1, 2)[:B] = np.stack([1 - sig, sig], 1)
        if class_ids is not None:
            _np(class_ids).reshape(-1)[:B] = sig > 0.5
        if unreduced_loss is not None:
            _np(unreduced_loss).reshape(-1)[:B] = O.head(x, _np(labels)[:B])[2]

    def mi_layer_stats(self, x, n, out4, ws, wsb):
        v = _np(x).reshape(-1)[:n]
        _np(out4)[:4] = [np.mean(v == 0), v.min(), v.max(), v.mean()]

    def mi_eval_accumulate(self, logits, labels, B, hist, counts, sums):
        from oracle.metrics import auc_thresholds
        x = _np(logits)[:B].astype(np.float32)
        y = _np(labels)[:B].astype(np.int64)
        e = np.exp(-np.abs(x))
        p = np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(np.float32)
        kk = (auc_thresholds()[None, :] < p[:, None]).sum(1)
        np.add.at(_np(hist), y * 201 + kk, 1)
        cls = (p > 0.5).astype(np.int64)
        c = _np(counts)
        c[0] += B; c[1] += y.sum(); c[2] += cls.sum(); c[3] += (cls == y).sum()
        c[4] += (cls & y).sum(); c[5] += (cls & (1 - y)).sum(); c[6] += ((1 - cls) & y).sum()
        xd = x.astype(np.float64)
        s = _np(sums)
        s[0] += (np.maximum(xd, 0) - xd * y + np.log1p(np.exp(-np.abs(xd)))).sum()
        s[1] += p.astype(np.float64).sum(); s[2] += y.sum()
