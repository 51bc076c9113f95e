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
  forward       ids -> owners (all_to_all) ; owners gather rows ; rows -> requesters (all_to_all)
  backward      per-entry row gradients -> owners (all_to_all) = the sparse "reduce-scatter";
                owners sum duplicates and apply the optimizer locally
  dense grads   one flat buffer, one all_reduce(SUM); every rank applies the same update

Collectives go through torch.distributed: backend "nccl" is RCCL on ROCm and takes device tensors
directly (point-to-point xGMI links: the all_to_all uses all 7 at once).  With the "gloo" backend
(CPU tests of this plumbing, or 2 ranks sharing one GPU in the -m gpu test) tensors are staged
through host memory.
"""
import torch
import torch.distributed as dist


class RowShard:
    def __init__(self, rank, world, group=None):
        if not (0 <= rank < world):
            raise ValueError("rank %d not in [0, %d)" % (rank, world))
        self.rank, self.world, self.group = int(rank), int(world), group
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

    def exchange_counts(self, send_counts, device):
        """counts[j] entries go to rank j -> returns how many arrive from each rank (host ints)."""
        t = torch.tensor(send_counts, dtype=torch.int64, device=device if self.direct else "cpu")
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t, group=self.group)
        return [int(v) for v in out.tolist()]

    def all_to_all(self, out, inp, out_counts, in_counts):
        """Rows (dim 0) of `inp`, split by in_counts, go to the ranks; `out` receives out_counts rows."""
        if self.direct:
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


def _route(m, ids):
    """Plan the exchange for this batch.  Returns (pos, send_ids, send_counts, recv_counts, nr):
    entry (b,f) travels in slot pos[b*F+f] of the send buffer, which is ordered by owner rank."""
    k, sh = m.k, m.shard
    comm = _comm(m)
    i32 = torch.int32
    B = ids.shape[0]
    n = B * m.F
    rows = m._buf("rows", (n,), i32)
    k.mi_global_rows(ids, m.field_off, B, m.F, rows)
    owner = m._buf("route_owner", (n,), i32)
    local = m._buf("route_local", (n,), i32)
    k.mi_shard_route(rows, n, sh.world, owner, local)
    order, present, seg, npresent = m._sort_unique(owner, n, sh.world, "route")   # stable partition by owner
    send_ids = m._buf("send_ids", (n,), i32)
    k.mi_gather_u32(local, order, n, send_ids)
    pos = m._buf("route_pos", (n,), i32)
    k.mi_invert_perm(order, n, pos)
    # split sizes must be host integers: one small device->host copy per step
    npres = int(npresent.item())
    pres = present[:npres].tolist()
    segs = seg[:npres + 1].tolist()
    send_counts = [0] * sh.world
    for j, r in enumerate(pres):
        send_counts[r] = segs[j + 1] - segs[j]
    recv_counts = comm.exchange_counts(send_counts, m.device)
    return pos, send_ids, send_counts, recv_counts, sum(recv_counts)


def _fetch_rows(m, ids, send_ids, pos, send_counts, recv_counts, nr, train):
    """Owners serve the requested rows; returns (src for _forward, owner-side bookkeeping)."""
    k = m.k
    comm = _comm(m)
    i32, f32 = torch.int32, torch.float32
    B = ids.shape[0]
    n = B * m.F
    recv_ids = m._buf("recv_ids", (max(nr, 1),), i32)[:nr]
    comm.all_to_all(recv_ids, send_ids, recv_counts, send_counts)
    book = None
    if train and nr > 0:
        book = m._sort_unique(recv_ids, nr, m.R_local, "own")     # (sorted_entry, uniq, seg, num_uniq)
        if m.adam_rows and m.step > 0:
            m._catchup(book[1], book[3], nr)
    got_rows = got_lin = None
    if m.use_emb:
        own_rows = m._buf("own_rows", (max(nr, 1), m.E))[:nr]
        own_lin = m._buf("own_lin", (max(nr, 1),))[:nr] if m.use_linear else None
        k.mi_gather_rows(m.table, m.lin_w if m.use_linear else None, recv_ids, nr, m.E, own_rows, own_lin)
        got_rows = m._buf("got_rows", (n, m.E))
        comm.all_to_all(got_rows, own_rows, send_counts, recv_counts)
    elif m.use_linear:
        own_lin = m._buf("own_lin", (max(nr, 1),))[:nr]
        k.mi_gather_u32(m.lin_w, recv_ids, nr, own_lin)
    if m.use_linear:
        got_lin = m._buf("got_lin", (n,))
        comm.all_to_all(got_lin, own_lin, send_counts, recv_counts)
    zero_off = m._ws.get("zero_off")
    if zero_off is None:
        zero_off = m._ws["zero_off"] = torch.zeros(m.F, dtype=torch.int64, device=m.device)
    return (got_rows, got_lin, zero_off, pos.view(B, m.F)), book, recv_ids


def sharded_eval_step(m, ids, labels, x_num):
    pos, send_ids, send_counts, recv_counts, nr = _route(m, ids)
    src, _, _ = _fetch_rows(m, ids, send_ids, pos, send_counts, recv_counts, nr, False)
    c = m._forward(ids, x_num, False, src)
    logits, loss, _ = m._head(c, labels, False, global_batch=ids.shape[0] * m.shard.world)
    return loss, logits


def sharded_train_step(m, ids, labels, x_num):
    """N-rank synchronous step.  Every rank must call it with the same local batch size.  Returns
    (this rank's share of the loss — already divided by the global batch —, local logits)."""
    comm = _comm(m)
    B = ids.shape[0]
    pos, send_ids, send_counts, recv_counts, nr = _route(m, ids)
    src, book, _ = _fetch_rows(m, ids, send_ids, pos, send_counts, recv_counts, nr, True)
    c = m._forward(ids, x_num, True, src)
    logits, loss, dlogit = m._head(c, labels, True, global_batch=B * m.shard.world)
    d_concat = m._backward_dense(c, dlogit)
    comm.all_reduce(m.d_grad)                                   # dense gradients: SUM over ranks
    d_rows, d_lin = m._entry_grads(c, d_concat, dlogit, pos)    # written in send (owner) order
    r_rows = r_lin = None
    if m.use_emb:
        r_rows = m._buf("recv_d_rows", (max(nr, 1), m.E))[:nr]
        comm.all_to_all(r_rows, d_rows, recv_counts, send_counts)
    if m.use_linear:
        r_lin = m._buf("recv_d_lin", (max(nr, 1),))[:nr]
        comm.all_to_all(r_lin, d_lin, recv_counts, send_counts)
    if book is not None:
        sorted_entry, uniq, seg, num_uniq = book
        m._apply(uniq, seg, sorted_entry, num_uniq, nr, r_rows, r_lin)
    else:
        m._apply(None, None, None, None, 0, None, None)
    return loss, logits
