"""Multi-GPU step: data-parallel MLP + row-sharded embedding tables, one process per GPU.

What the reference does for >1 worker is TensorFlow's asynchronous parameter-server replication
behind ``tf.estimator.train_and_evaluate`` (trainers/deep_fm.py:178, distributed.md:58-82): every
variable read / sparse update crosses worker<->ps over gRPC.  Here the same data flow is made
synchronous and mapped onto xGMI:

  examples      split across ranks (global batch = world x local batch; the mean loss of the
                contrib head is over the GLOBAL batch, so an N-rank step equals a 1-rank step on
                the concatenated batch)
  table rows    row r lives on rank r % world (interleaved: Zipf heads spread evenly) at local
                index r // world, together with its optimizer slots and Adam step stamp
  forward       DISTINCT row requests -> owners (all_to_all) ; owners gather rows ; rows -> requesters
                (all_to_all): a row many entries of a batch ask for (skewed ids) crosses the link once
  backward      requesters sum the entry gradients of each distinct request, then gradient rows -> owners
                (all_to_all) = the sparse "reduce-scatter"; owners sum over requesters and apply the
                optimizer locally.  Volumes are U_local x E, not B F x E.
  dense grads   one flat buffer, one all_reduce(SUM); every rank applies the same update
  pipelining    a train step splits its local batch into chunks: while chunk c runs forward and
                backward, the rows of chunk c+1 and the row gradients of chunk c-1 are on the
                links (asynchronous all_to_all on RCCL's stream).  Routing, the id exchange, the
                owners' unique/catch-up bookkeeping and the sparse apply are done once per step, so
                the result is the unchunked step's up to fp32 summation order of the dense gradient

Collectives go through torch.distributed: backend "nccl" is RCCL on ROCm and takes device tensors
directly (point-to-point xGMI links: the all_to_all uses all 7 at once).  With the "gloo" backend
(CPU tests of this plumbing, or 2 ranks sharing one GPU in the -m gpu test) tensors are staged
through host memory.
"""
import torch
import torch.distributed as dist


class RowShard:
    def __init__(self, rank, world, group=None, chunks=None):
        """chunks: pipeline depth of a train step (None: chosen from the batch and world size, see _n_chunks)."""
        if not (0 <= rank < world):
            raise ValueError("rank %d not in [0, %d)" % (rank, world))
        self.rank, self.world, self.group = int(rank), int(world), group
        self.chunks = chunks
        self.comm = None

    def local_rows(self, R):
        return (R - self.rank + self.world - 1) // self.world


class Comm:
    """Thin wrapper over torch.distributed for the three collectives the step needs."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.direct = dist.get_backend(group) == "nccl"   # RCCL: device tensors go straight in

    def exchange_counts(self, counts_dev, C):
        """counts_dev [C * world] int32 (device): distinct requests of chunk c for rank j at c * world + j.
        One small all_to_all on the device buffers, then ONE device->host copy of both tables (the split
        sizes of all_to_all_single must be host integers): (send_counts[c][j], recv_counts[c][j])."""
        W = self.world
        send = counts_dev.view(C, W).t().contiguous().view(-1).to(torch.int64)       # [dest rank][chunk]
        if not self.direct:
            send = send.cpu()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        both = torch.stack([send, recv]).cpu().tolist()                                # the step's one host sync
        sc = [[int(both[0][j * C + c]) for j in range(W)] for c in range(C)]
        rc = [[int(both[1][j * C + c]) for j in range(W)] for c in range(C)]
        return sc, rc

    def all_to_all(self, out, inp, out_counts, in_counts, async_op=False):
        """Rows (dim 0) of `inp`, split by in_counts, go to the ranks; `out` receives out_counts rows.
        async_op (RCCL only): returns a handle whose wait() orders the current stream after the
        exchange; the buffers must stay untouched until then."""
        if self.direct:
            return dist.all_to_all_single(out, inp, list(out_counts), list(in_counts), group=self.group,
                                          async_op=async_op) if async_op else \
                dist.all_to_all_single(out, inp, list(out_counts), list(in_counts), group=self.group)
        else:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.detach().cpu().contiguous(), list(out_counts), list(in_counts),
                                   group=self.group)
            out.copy_(o)

    def all_reduce(self, t):
        if self.direct:
            dist.all_reduce(t, group=self.group)
        else:
            c = t.detach().cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)

    def broadcast(self, t, src=0):
        if self.direct:
            dist.broadcast(t, src, group=self.group)
        else:
            c = t.detach().cpu()
            dist.broadcast(c, src, group=self.group)
            t.copy_(c)


def _comm(m):
    if m.shard.comm is None:
        m.shard.comm = Comm(m.shard.group)
        if m.shard.comm.world != m.shard.world or m.shard.comm.rank != m.shard.rank:
            raise ValueError("RowShard(rank=%d, world=%d) does not match the process group (%d of %d)" %
                             (m.shard.rank, m.shard.world, m.shard.comm.rank, m.shard.comm.world))
    return m.shard.comm


def broadcast_dense(m, src=0):
    """Make the replicated dense variables (and their slots) identical on every rank."""
    c = _comm(m)
    for t in (m.dense, m.d_s0, m.d_s1, getattr(m, "dl_s0", None), getattr(m, "dl_s1", None)):
        if t is not None:
            c.broadcast(t, src)


def _wait(handles):
    for h in handles:
        if h is not None:
            h.wait()


def _n_chunks(m, B, train):
    if not train:
        return 1
    c = m.shard.chunks
    if c is None:
        # A chunk is the M of every MLP GEMM: below 32768 examples their grids no longer cover the 256 CUs (128-row
        # tiles; measured with one rank, bench.py --force-shard, B = 65536: 5.2 / 5.9 / 7.0 ms per step with 1 / 2 / 4
        # chunks — GEMMs 1.5 / 1.8 / 2.6 ms).  What a chunk buys is exchange time hidden under compute, (1 - 1/C) of it;
        # xGMI is point to point, so a rank's 7/8 x 436 MB per direction share 7 links at 8 ranks but 218 MB share ONE
        # link at 2: small worlds are exchange-bound and take the smaller chunks.  (Estimated, not measured: no
        # multi-GPU box in this pool.)
        size = 16384 if m.shard.world <= 4 else 32768
        c = min(4, B // size) if B >= 2 * size else (2 if 2048 <= B < 16384 else 1)
    c = max(1, min(int(c), B))
    while B % c:
        c -= 1
    return c


def _route(m, ids, C):
    """Plan the exchange for this batch, C chunks of B/C examples.  The entries are sorted by request key
    (chunk, owner, owner-local row); the DISTINCT keys are the requests that travel, in send order.  Returns
    (slot [B*F]: distinct request of every entry = slot of its row in the receive buffer, send_rows [U]: the
    owner-local rows to ask for, the sort's (sorted_entry, seg_start) over the entries, send_counts[c][rank],
    recv_counts[c][rank])."""
    k, sh = m.k, m.shard
    comm = _comm(m)
    i32 = torch.int32
    B = ids.shape[0]
    n = B * m.F
    Rl = (m.R + sh.world - 1) // sh.world                     # rows per rank (upper bound): the key's row range
    rows = m._buf("rows", (n,), i32)
    k.mi_global_rows(ids, m.field_off, B, m.F, rows)
    key = m._buf("route_key", (n,), i32)
    k.mi_shard_keys(rows, n, sh.world, (n // C) if C > 1 else 0, Rl, key)
    sorted_entry, uniq, seg, num_uniq = m._sort_unique(key, n, C * sh.world * Rl, "route")
    send_rows = m._buf("send_rows", (n,), i32)
    counts = m._buf("route_counts", (C * sh.world,), i32)
    k.mi_route_requests(uniq, num_uniq, n, Rl, C * sh.world, send_rows, counts)
    slot = m._buf("route_slot", (n,), i32)
    k.mi_segment_slots(seg, sorted_entry, num_uniq, n, slot)
    send_counts, recv_counts = comm.exchange_counts(counts, C)
    return slot, send_rows, sorted_entry, seg, send_counts, recv_counts


def _zero_off(m):
    z = m._ws.get("zero_off")
    if z is None:
        z = m._ws["zero_off"] = torch.zeros(m.F, dtype=torch.int64, device=m.device)
    return z


def _sharded_step(m, ids, labels, x_num, train):
    """Forward (+ backward and apply when train) of one local batch on N ranks; see the module
    docstring.  Returns (this rank's share of the loss — already divided by the global batch —,
    local logits)."""
    k = m.k
    comm = _comm(m)
    i32 = torch.int32
    B = ids.shape[0]
    F, E = m.F, m.E
    n = B * F
    C = _n_chunks(m, B, train)
    Bc = B // C
    if train and hasattr(m, "_split_weights_ahead"):
        m._split_weights_ahead()            # the MLP's weight planes, on a side stream beside the routing
    slot, send_rows, sorted_entry, seg, send_counts, recv_counts = _route(m, ids, C)
    nsc = [sum(sc) for sc in send_counts]               # distinct requests of chunk c (all owners)
    uoff = [0]
    for v in nsc:
        uoff.append(uoff[-1] + v)
    U = uoff[-1]
    nrc = [sum(rc) for rc in recv_counts]
    roff = [0]
    for v in nrc:
        roff.append(roff[-1] + v)
    nr = roff[-1]
    m.last_exchange = {"entries": n, "requests_sent": U, "requests_received": nr}    # (tests / bench: the dedup's effect)

    # requests to their owners (small), then the owners' bookkeeping for the WHOLE step: which rows are
    # touched, and TF Adam's catch-up on them before any of them is read
    recv_ids = m._buf("recv_ids", (max(nr, 1),), i32)[:nr]
    for c in range(C):
        comm.all_to_all(recv_ids[roff[c]:roff[c + 1]], send_rows[uoff[c]:uoff[c + 1]], recv_counts[c], send_counts[c])
    book = None
    if train and nr > 0:
        book = m._sort_unique(recv_ids, nr, m.R_local, "own")     # (sorted_entry, uniq, seg, num_uniq)
        if m.adam_rows and m.step > 0:
            m._catchup(book[1], book[3], nr, defer=True)

    own_rows = m._buf("own_rows", (max(nr, 1), E))[:nr] if m.use_emb else None
    own_lin = m._buf("own_lin", (max(nr, 1),))[:nr] if m.use_linear else None
    got_rows = m._buf("got_rows", (max(U, 1), E)) if m.use_emb else None
    got_lin = m._buf("got_lin", (max(U, 1),)) if m.use_linear else None

    def serve(c):
        """owners gather chunk c's rows and send them back; returns the exchange handles"""
        lo, hi = roff[c], roff[c + 1]
        hs = []
        if m.use_emb:
            k.mi_gather_rows(m.table, m.lin_w if m.use_linear else None, recv_ids[lo:hi], hi - lo, E, own_rows[lo:hi],
                             own_lin[lo:hi] if m.use_linear else None, m.ls)
            hs.append(comm.all_to_all(got_rows[uoff[c]:uoff[c + 1]], own_rows[lo:hi], send_counts[c], recv_counts[c], True))
        elif m.use_linear:
            k.mi_gather_rows(None, m.lin_w, recv_ids[lo:hi], hi - lo, E, None, own_lin[lo:hi], m.ls)
        if m.use_linear:
            hs.append(comm.all_to_all(got_lin[uoff[c]:uoff[c + 1]], own_lin[lo:hi], send_counts[c], recv_counts[c], True))
        return hs

    d_rows = m._buf("d_rows", (max(U, 1), E)) if (train and m.use_emb) else None      # one row per distinct request, send order
    d_lin = m._buf("d_lin", (max(U, 1),)) if (train and m.use_linear) else None
    r_rows = m._buf("recv_d_rows", (max(nr, 1), E))[:nr] if (train and m.use_emb) else None
    r_lin = m._buf("recv_d_lin", (max(nr, 1),))[:nr] if (train and m.use_linear) else None
    logits_all = m._buf("logits_all", (B,)) if C > 1 else None
    loss_all = m._buf("loss_all", (1,)) if C > 1 else None
    acc = m._buf("d_grad_acc", (m.P,)) if (train and C > 1) else None
    slot2 = slot.view(B, F)
    zero_off = _zero_off(m)

    rows_h = serve(0)
    grad_h = []
    loss = logits = None
    for c in range(C):
        nxt = serve(c + 1) if c + 1 < C else []          # on the links while chunk c computes
        _wait(rows_h)
        rows_h = nxt
        sl = slice(c * Bc, (c + 1) * Bc)
        m._chunk = c
        cc = m._forward(ids[sl], None if x_num is None else x_num[sl], train, (got_rows, got_lin, zero_off, slot2[sl]))
        logits, loss, dlogit = m._head(cc, None if labels is None else labels[sl], train, global_batch=B * m.shard.world)
        if C > 1:
            logits_all[sl].copy_(logits)
            if loss is not None:
                if c == 0:
                    loss_all.copy_(loss)
                else:
                    k.mi_axpy(loss_all, loss, 1, 1.0)
        if not train:
            continue
        d_concat = m._backward_dense(cc, dlogit)
        if C > 1:
            if c == 0:
                acc.copy_(m.d_grad)
            else:
                k.mi_axpy(acc, m.d_grad, m.P, 1.0)
        # the chunk's entry gradients summed per distinct request, written at the request's send slot
        ulo, uhi = uoff[c], uoff[c + 1]
        if uhi > ulo:
            k.mi_entry_grads_segsum(got_rows if m.use_mf else None, seg, sorted_entry, ulo, uhi - ulo,
                                    d_concat if m.use_emb else None, m.D, cc["sumv"] if m.use_mf else None,
                                    dlogit if m.use_mf else None, dlogit if m.use_linear else None, c * Bc, F, E,
                                    d_rows, d_lin)
        lo, hi = roff[c], roff[c + 1]
        if m.use_emb:
            grad_h.append(comm.all_to_all(r_rows[lo:hi], d_rows[ulo:uhi], recv_counts[c], send_counts[c], True))
        if m.use_linear:
            grad_h.append(comm.all_to_all(r_lin[lo:hi], d_lin[ulo:uhi], recv_counts[c], send_counts[c], True))
    m._chunk = 0
    if C > 1:
        logits, loss = logits_all, (loss_all if loss is not None else None)
    if not train:
        return loss, logits
    if C > 1:
        m.d_grad.copy_(acc)
    comm.all_reduce(m.d_grad)                                   # dense gradients: SUM over ranks
    _wait(grad_h)
    if book is not None:
        bs_entry, buniq, bseg, bnum = book
        m._apply(buniq, bseg, bs_entry, bnum, nr, r_rows, r_lin)
    else:
        m._apply(None, None, None, None, 0, None, None)
    return loss, logits


def sharded_eval_step(m, ids, labels, x_num):
    return _sharded_step(m, ids, labels, x_num, False)


def sharded_train_step(m, ids, labels, x_num):
    """N-rank synchronous step.  Every rank must call it with the same local batch size."""
    return _sharded_step(m, ids, labels, x_num, True)
