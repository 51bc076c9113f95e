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


def rccl_options(timeout=None):
    """Keyword arguments for dist.init_process_group / dist.new_group with the "nccl" (RCCL) backend: RCCL's stream from torch's
    HIGH-priority pool.  HIP serves each stream priority from its own pool of hardware queues, and every stream of the engine
    is a normal-priority one tested to run beside the step's stream (engine._new_side_stream) — the collectives' stream, which
    torch takes from its pool wherever the pool's cursor happens to stand, then cannot land on a hardware queue the step
    uses (two streams on one queue run one after the other: tools/stream_alias_probe.py, profiles/r05_stream_aliasing.md)."""
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        if timeout is not None:
            opts._timeout = timeout                     # (the group's timeout keyword overrides it anyway; equal values: no warning)
        return {"pg_options": opts}
    except Exception:                                   # (a build without the NCCL backend: gloo runs need no options)
        return {}


class RowShard:
    def __init__(self, rank, world, group=None, chunks=None, chunk_compute=None, route_ahead=None, packed=False, sim_links=None):
        """chunks: pipeline depth of a train step (None: chosen from the batch and world size, see _n_chunks).
        route_ahead: True (default) — an announced next batch is routed during this step, on a side stream and a SECOND
        RCCL communicator (see Comm / _route_ahead); False — the whole step runs on ONE communicator: no second
        communicator is created, every COLLECTIVE of a step is issued in program order on the step's stream (the
        conservative form for a first run on new hardware: two communicators progressing concurrently on two streams
        can deadlock if a device-synchronising call lands between their kernels in different orders on different
        ranks; all buffers the step needs are sized before its first collective either way, see _sharded_step).  An
        announced next batch is still prepared ahead (round 5): its request sort — local work, no collective — on a
        side stream; its count exchange and id exchange on THIS communicator and the step's stream, in program order
        (after this step's row exchanges / before its dense all-reduce); the owners' sort behind that on the side
        stream (_ahead_in_order).  The next step then starts with the catch-up, and no host wait drains the queue.
        chunk_compute: True (the default) — every chunk runs its own forward / backward: the exchanges of one chunk travel
        under the whole compute of its neighbours, the MLP's GEMMs shrink to a chunk's examples; False — only the exchanges
        and the embedding-side kernels are chunked, the MLP runs once on the whole batch (the row exchange can then hide under
        the owners' gathers only, the gradient exchange under the weight gradients).  Rounds 3-5 defaulted to False from 8
        ranks on, by arithmetic ("7 links per rank: the exchanges are short against the step").  The rehearsal with modelled
        link time says otherwise (tools/sim_ranks.py, profiles/r05_sim_ranks.md: real kernels of one rank's share of an 8-rank
        job, 45 / 60 GB/s per link): 2 chunks with their own MLP pass 4.63 / 4.34 ms per step against 5.38 / 4.79 ms with one
        pass — half of the ~2.2 ms of link time hidden instead of a fifth.
        sim_links: see below (tools/sim_ranks.py only)."""
        if not (0 <= rank < world):
            raise ValueError("rank %d not in [0, %d)" % (rank, world))
        self.rank, self.world, self.group = int(rank), int(world), group
        self.chunks = chunks
        self.chunk_compute = True if chunk_compute is None else bool(chunk_compute)
        self.route_ahead = True if route_ahead is None else bool(route_ahead)
        # packed: rows and wide weights (and their gradients) travel as one record of E + 4 floats per request — one
        # collective per chunk and direction instead of two (_sharded_step).  OFF by default: measured with one rank
        # (bench.py --force-shard, profiles/r04_sharded_one_rank.md) the 272-byte records cost every kernel that walks them
        # a third cache line per row — gather_rows +0.04, the planes gather +0.05, the segment sum +0.05, the sparse apply
        # +0.13 ms per step — more than two small RCCL launches per chunk can give back.
        self.packed = bool(packed)
        # sim_links (tools/sim_ranks.py ONLY, a one-rank group): {"world": N, "gbs": aggregate GB/s per direction, "latency_us": per
        # collective} — every exchange of the step is followed, on a stream of its own (as RCCL's is), by a spin kernel as long as
        # the N-rank exchange of the same requests would keep the links busy: (N - 1) / N of the bytes at `gbs` + the latency.  The
        # step's dependency structure and its real kernels against MODELLED link time: a rehearsal of the overlap, not a
        # measurement of xGMI.
        self.sim_links = dict(sim_links) if sim_links else None
        self.comm = None

    def local_rows(self, R):
        return (R - self.rank + self.world - 1) // self.world


class Comm:
    """Thin wrapper over torch.distributed for the three collectives the step needs."""

    def __init__(self, group=None, second=True):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.direct = dist.get_backend(group) == "nccl"   # RCCL: device tensors go straight in
        # A second communicator for the count exchange of a batch routed AHEAD (_route_ahead): RCCL runs a communicator's
        # collectives in issue order on one stream, and that exchange waits for the next batch's sorts — on this step's
        # communicator it would hold up this step's row exchanges behind them.  (Created by every rank, here, in the
        # same order.)
        # (RowShard(route_ahead=False): no second communicator exists at all)
        ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
        self.ahead_group = dist.new_group(ranks=ranks, **(rccl_options() if self.direct else {})) if second else None
        self._pinned = {}                # (C, parity) -> pinned host buffer of the count tables, allocated once
        self._pin_turn = 0
        if self.direct and torch.cuda.is_available():
            # RCCL creates a communicator — and the point-to-point channels an all_to_all uses — at the first collective
            # that needs them, with allocations and device-wide synchronisations of its own.  Do that HERE, for both
            # communicators, one after the other and in the same order on every rank, with the device idle: the first
            # train step then issues collectives on communicators that already exist (the second one's first use would
            # otherwise fall on a side stream in the middle of the first step's exchanges).
            dev = torch.device("cuda", torch.cuda.current_device())
            for g in ([self.group] if self.ahead_group is None else [self.group, self.ahead_group]):
                one = torch.ones(self.world, dtype=torch.int64, device=dev)
                got = torch.empty_like(one)
                dist.all_reduce(one, group=g)
                dist.all_to_all_single(got, one, group=g)
                torch.cuda.synchronize(dev)
                if int(got.sum().item()) != self.world * self.world:
                    raise RuntimeError("rank %d: the warm-up all_to_all returned %s" % (self.rank, got.tolist()))
        # send order of the owners (mi_shard_keys, self_rank): the other ranks in rank order, this rank LAST
        self.pos_of_rank = [j if j < self.rank else (self.world - 1 if j == self.rank else j - 1) for j in range(self.world)]
        self._pos_dev = {}

    def exchange_counts(self, counts_dev, C):
        """counts_dev [C * world] int32 (device): distinct requests of chunk c for the owner at send position p
        (pos_of_rank) at c * world + p.  One small all_to_all on the device buffers, then ONE device->host copy of both
        tables (the split sizes of all_to_all_single must be host integers): (send_counts[c][j], recv_counts[c][j]),
        j = rank."""
        return self.finish_counts(self.start_counts(counts_dev, C))

    def start_counts(self, counts_dev, C, ahead=False):
        """The count exchange WITHOUT the host wait: the all_to_all and an asynchronous copy of both tables into pinned
        host memory are enqueued on the current stream; finish_counts() waits for the copy's event — immediately
        (exchange_counts), or a step later, when the routing of the next batch was started ahead."""
        W = self.world
        pos = self._pos_dev.get(counts_dev.device)
        if pos is None:
            pos = self._pos_dev[counts_dev.device] = torch.tensor(self.pos_of_rank, dtype=torch.int64, device=counts_dev.device)
        send = counts_dev.view(C, W).index_select(1, pos).t().contiguous().view(-1).to(torch.int64)   # [dest rank][chunk]
        if not self.direct:
            send = send.cpu()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.ahead_group if (ahead and self.ahead_group is not None) else self.group)
        both = torch.stack([send, recv])
        if both.device.type != "cuda":
            return {"host": both, "event": None, "C": C}
        # pinned host memory is allocated ONCE per table shape (two buffers in turn: a plan made ahead is still pending
        # while this step's is read) — a hipHostMalloc between two ranks' collectives is a device-wide synchronisation
        key = (tuple(both.shape), self._pin_turn & 1)
        self._pin_turn += 1
        host = self._pinned.get(key)
        if host is None:
            host = self._pinned[key] = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
        host.copy_(both, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return {"host": host, "event": ev, "C": C, "keep": both}

    def finish_counts(self, pending):
        if pending["event"] is not None:
            pending["event"].synchronize()                                            # the step's one host sync
        both, C, W = pending["host"].tolist(), pending["C"], self.world
        sc = [[int(both[0][j * C + c]) for j in range(W)] for c in range(C)]
        rc = [[int(both[1][j * C + c]) for j in range(W)] for c in range(C)]
        return sc, rc

    def all_to_all(self, out, inp, out_counts, in_counts, async_op=False, ahead=False):
        """Rows (dim 0) of `inp`, split by in_counts, go to the ranks; `out` receives out_counts rows.
        async_op (RCCL only): returns a handle whose wait() orders the current stream after the
        exchange; the buffers must stay untouched until then.  The step's exchanges carry nothing from a rank to
        itself (split size 0 at its own index: those rows never leave the device, see _sharded_step); with one rank
        there is nothing to exchange at all."""
        if self.world == 1:
            return None
        group = self.ahead_group if (ahead and self.ahead_group is not None) else self.group      # (ahead: the next batch's id exchange, see _own_ahead)
        if self.direct:
            return dist.all_to_all_single(out, inp, list(out_counts), list(in_counts), group=group,
                                          async_op=async_op) if async_op else \
                dist.all_to_all_single(out, inp, list(out_counts), list(in_counts), group=group)
        else:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.detach().cpu().contiguous(), list(out_counts), list(in_counts), group=group)
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
        m.shard.comm = Comm(m.shard.group, second=m.shard.route_ahead)
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


class _SimHandle:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def _sim_exchange(m, nbytes, collectives=1, ahead=False):
    """RowShard.sim_links: a spin kernel on the simulated communicator's stream, behind everything the current stream has
    enqueued (RCCL's stream waits for the caller's at the call) and behind the simulated exchanges before it (one
    communicator: its collectives run in issue order); returns a handle like an asynchronous collective's."""
    sim = m.shard.sim_links
    if sim is None or m.device.type != "cuda":
        return None
    key = "sim_link_stream2" if (ahead and m.shard.route_ahead) else "sim_link_stream"      # (a second communicator: its own stream)
    ls = m._ws.get(key)
    if ls is None:
        ls = m._ws[key] = m._new_side_stream(priority=int(sim.get("priority", -1)))
    if "sim_cycles_per_us" not in m._ws:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); torch.cuda._sleep(20_000_000); b.record(); torch.cuda.synchronize()
        m._ws["sim_cycles_per_us"] = 20_000_000 / (a.elapsed_time(b) * 1e3)
    N = int(sim["world"])
    us = collectives * float(sim.get("latency_us", 40.0)) + nbytes * (N - 1) / N / (float(sim["gbs"]) * 1e3)
    m._ws["sim_link_us"] = m._ws.get("sim_link_us", 0.0) + us
    ls.wait_stream(torch.cuda.current_stream())
    trace = m._ws.get("sim_trace")                              # (tools/sim_ranks.py --trace: events around every modelled exchange)
    with torch.cuda.stream(ls):
        if trace is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        torch.cuda._sleep(int(us * m._ws["sim_cycles_per_us"]))
        ev = torch.cuda.Event(enable_timing=trace is not None)
        ev.record()
        if trace is not None:
            trace.append((e0, ev, us, nbytes))
    return _SimHandle(ev)


def _n_chunks(m, B, train):
    if not train:
        return 1
    c = m.shard.chunks
    if c is None:
        # A chunk is the M of every MLP GEMM: below 32768 examples their grids no longer cover the 256 CUs (128-row tiles;
        # one rank, B = 65536: GEMMs 1.5 / 1.8 / 2.6 ms with 1 / 2 / 4 chunks).  What a chunk buys is exchange time hidden
        # under compute.  Rounds 3-5 took 16,384-example chunks up to 4 ranks ("small worlds are exchange-bound") by
        # arithmetic; with modelled link time (tools/sim_ranks.py, profiles/r05_sim_ranks.md) 2 chunks of 32,768 beat 4 of
        # 16,384 at 2, 4 AND 8 ranks (45 GB/s per link: 12.3 / 7.0 / 4.6 against 12.7 / 7.4 / 4.9 ms): what the smaller
        # chunks hide more, their GEMMs lose.
        size = 32768
        c = min(4, B // size) if B >= 2 * size else (2 if 2048 <= B < 16384 else 1)
    c = max(1, min(int(c), B))
    while B % c:
        c -= 1
    return c


def _route(m, ids, C, tag="", ahead=False, exchange=True):
    """Plan the exchange for this batch, C chunks of B/C examples.  The entries are sorted by request key
    (chunk, owner, owner-local row) — owners in send order: the other ranks by rank, this rank last; the DISTINCT keys
    are the requests, in send order (a chunk's requests to the rank itself end its run and stay on the device).  Returns
    (slot [B*F]: distinct request of every entry = slot of its row in the receive buffer, send_rows [U]: the
    owner-local rows to ask for, the sort's (sorted_entry, seg_start) over the entries, and the pending count
    exchange: Comm.finish_counts gives send_counts[c][rank], recv_counts[c][rank]).
    tag / ahead: a second set of buffers and a sort workspace of its own, for the routing of the NEXT batch started on a
    side stream while this step still uses the first set (_route_ahead).  exchange=False: the local half only — the
    plan carries the request counts ("counts") and no collective has been issued (_take_route starts the count exchange)."""
    k, sh = m.k, m.shard
    comm = _comm(m)
    i32 = torch.int32
    B = ids.shape[0]
    n = B * m.F
    Rl = (m.R + sh.world - 1) // sh.world                     # rows per rank (upper bound): the key's row range
    rows = m._buf("rows" + tag, (n,), i32)
    k.mi_global_rows(ids, m.field_off, B, m.F, rows)
    key = m._buf("route_key" + tag, (n,), i32)
    k.mi_shard_keys(rows, n, sh.world, (n // C) if C > 1 else 0, Rl, sh.rank, key)
    # (slot: the distinct request every entry belongs to = where its row lands in the receive buffer; the sort's
    # compaction writes it as it goes)
    slot = m._buf("route_slot" + tag, (n,), i32)
    sorted_entry, uniq, seg, num_uniq = m._sort_unique(key, n, C * sh.world * Rl, "route" + tag,
                                                       ws_name="sort_ws_ahead" if ahead else "sort_ws", slot=slot)
    send_rows = m._buf("send_rows" + tag, (n,), i32)
    counts = m._buf("route_counts" + tag, (C * sh.world,), i32)
    k.mi_route_requests(uniq, num_uniq, n, Rl, C * sh.world, send_rows, counts)
    plan = {"slot": slot, "send_rows": send_rows, "sorted_entry": sorted_entry, "seg": seg, "C": C, "tag": tag}
    if exchange:
        plan["pending"] = comm.start_counts(counts, C, ahead=ahead)
        if m.shard.sim_links and comm.world == 1:
            _wait([_sim_exchange(m, 0, ahead=ahead)])
    else:
        plan["counts"] = counts
    return plan


def _finish_plan(m, plan):
    """The plan's split sizes on the host (Comm.finish_counts: the step's one host wait) and what follows from them:
    where every chunk's requests sit in the buffers of both sides.
    A rank's requests to ITSELF never leave the device.  They end every chunk's run on both sides (mi_shard_keys: self
    last; the owners' buffers: sources in rank order, self last), the exchanges get split size 0 at the rank's own index
    and ship the runs' heads; the tails are served in place: the owner's gather writes them straight into the receive
    buffer, the requester's segment sum straight into the buffer its own apply reads.  (Through RCCL the self piece was
    a device copy at ~1.1 TB/s: 0.74 ms of the one-rank step, 1/8 of it at 8 ranks.)"""
    if "uoff" in plan:
        return plan
    me, C = m.shard.rank, plan["C"]
    if "pending" not in plan:            # (the local half was made ahead on one communicator: the count exchange starts here, in program order)
        plan["pending"] = _comm(m).start_counts(plan.pop("counts"), C)
        if m.shard.sim_links and _comm(m).world == 1:
            _wait([_sim_exchange(m, 0)])
    send_counts, recv_counts = _comm(m).finish_counts(plan.pop("pending"))
    uoff, roff = [0], [0]
    for c in range(C):
        uoff.append(uoff[-1] + sum(send_counts[c]))       # distinct requests of chunk c (all owners)
        roff.append(roff[-1] + sum(recv_counts[c]))       # requests chunk c brings this owner
    n_self = [sc[me] for sc in send_counts]
    for c in range(C):
        if recv_counts[c][me] != n_self[c]:
            raise RuntimeError("count exchange: rank %d sends itself %d requests but receives %d" % (me, n_self[c], recv_counts[c][me]))
    plan.update(uoff=uoff, roff=roff, n_self=n_self,
                sc0=[[0 if j == me else v for j, v in enumerate(sc)] for sc in send_counts],
                rc0=[[0 if j == me else v for j, v in enumerate(rc)] for rc in recv_counts],
                # chunk c: requests [uoff[c], umid[c]) travel, [umid[c], uoff[c+1]) are the rank's own;
                # owner side: [roff[c], rmid[c]) arrived, [rmid[c], roff[c+1]) are its own
                umid=[uoff[c + 1] - n_self[c] for c in range(C)], rmid=[roff[c + 1] - n_self[c] for c in range(C)])
    return plan


def _owners_side(m, plan, train, ahead=False, sort_stream=None):
    """Requests to their owners (small), then — for a train step — which rows this owner's step touches (sort + unique
    of everything it was asked for: the sparse apply's segments and the catch-up's row list).  ahead: on the second
    communicator (if there is one), into the plan's own buffer set (see _own_ahead).  sort_stream: the id exchange on the
    current stream, the sort — local work — behind it on that stream (_ahead_in_order)."""
    if "recv_ids" in plan:
        return plan
    comm = _comm(m)
    tag, C = plan["tag"], plan["C"]
    uoff, roff, umid, rmid = plan["uoff"], plan["roff"], plan["umid"], plan["rmid"]
    nr = roff[-1]
    recv_ids = m._buf("recv_ids" + tag, (max(2 * plan["slot"].numel(), nr, 1),), torch.int32)[:nr]      # (capacity: see _sharded_step)
    send_rows = plan["send_rows"]
    for c in range(C):
        comm.all_to_all(recv_ids[roff[c]:rmid[c]], send_rows[uoff[c]:umid[c]], plan["rc0"][c], plan["sc0"][c], ahead=ahead)
        if plan["n_self"][c]:
            recv_ids[rmid[c]:roff[c + 1]].copy_(send_rows[umid[c]:uoff[c + 1]])
    if m.shard.sim_links and comm.world == 1:                   # (tools/sim_ranks.py: the id exchanges, blocking collectives)
        _wait([_sim_exchange(m, nr * 4, C, ahead=ahead)])
    plan["recv_ids"] = recv_ids
    plan["book"] = None
    if train and nr > 0:        # (sorted_entry, uniq, seg, num_uniq)
        sort = lambda: m._sort_unique(recv_ids, nr, m.R_local, "own" + tag, ws_name="sort_ws_ahead" if ahead else "sort_ws",
                                      cap=2 * plan["slot"].numel())
        if sort_stream is None:
            plan["book"] = sort()
        else:
            sort_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sort_stream):
                plan["book"] = sort()
    return plan


def _route_ahead(m, next_ids, C, after=None):
    """The routing of the NEXT batch — a pure function of its ids: a radix sort's worth of small kernels, the count
    exchange (on a communicator of its own) and the copy of the split sizes to the host — started on a side stream at
    the HEAD of this step, as soon as this step's own plan is on the host.  The next sharded_train_step picks it up if it
    is given that very tensor, unmodified: its split sizes have then been on the host for a whole step — the host, which
    enqueues a step in less time than the GPU takes to run it, never waits for them (started at the END of the step
    the wait only moved: measured, no gain), and no routing sort stands at the head of the step.
    Every rank must announce (or not announce) its next batch alike: the count exchange is a collective.
    after: an event of the step's stream recorded at the head of the step — the side stream waits for THAT, while the host
    enqueues these ~15 launches after the step's own catch-up (the host is less than a step ahead of the GPU here: what it
    enqueues first starts first; kernel trace of round 5: the step's stream sat idle for 0.13 ms behind them)."""
    side = m._ws.get("route_stream")
    if side is None:
        # A NORMAL-priority stream that has been TESTED to run beside the step's stream and the engine's other side streams
        # (engine._new_side_stream).  History: a plain normal-priority pool stream had landed on the step's hardware queue
        # with one communicator (rocprofv3 kernel trace of round 5: the owners' sort serial between the weight gradients and
        # the apply, +0.2 ms per step), so the one-communicator mode used a HIGH-priority stream (its own pool of hardware
        # queues) — which, with 4 pool streams taken earlier in the process, doubled the step (tools/stream_alias_probe.py:
        # 4.4 -> 8.2 ms; profiles/r05_stream_aliasing.md).  The tested stream is flat over 0-6 streams taken before.
        prio = 0
        side = m._ws["route_stream"] = (m._new_side_stream(priority=prio) if hasattr(m, "_new_side_stream") else
                                        torch.cuda.Stream(device=m.device, priority=prio)) if m.device.type == "cuda" else None
    tag = "_b" if getattr(m, "_route_tag", "") == "" else ""      # the buffer set this step's plan does NOT live in
    second = bool(m.shard.route_ahead)     # a second communicator exists: the count exchange goes ahead too
    if side is None:
        plan = _route(m, next_ids, C, tag, ahead=True, exchange=second)
    else:
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            plan = _route(m, next_ids, C, tag, ahead=True, exchange=second)
    m._routed = {"ids": next_ids, "version": next_ids._version, "plan": plan, "C": C, "tag": tag, "stream": side, "second": second}


def _own_ahead(m):
    """Second half of the routing ahead, at the END of a step's enqueue: the host takes the next batch's split sizes (the
    event it would otherwise wait for at the head of the next step — the GPU is at the same point of its queue either
    way) and enqueues, on the side stream and the second communicator, the next batch's id exchange and the owners' sort
    of the requests they receive.  Both depend on ids only; the next step then starts with the catch-up."""
    r = getattr(m, "_routed", None)
    if r is None or not r.get("second", True):       # (one communicator: the collectives of the next batch wait for its own step)
        return
    plan = _finish_plan(m, r["plan"])
    if r["stream"] is None:
        _owners_side(m, plan, True, ahead=True)
    else:
        # beside this step's sparse apply (HBM-bound, 0.6 ms), not beside the weight-gradient GEMMs the host's enqueue
        # would otherwise put it next to: their grids are whole waves of workgroups over the 256 CUs, and a sort kernel
        # on a few CUs costs them a third wave (kernel trace: wgrad_pl_k 298 -> 341 us, the layer-1 data gradient 335 -> 354)
        if r.get("gate") is not None:
            r["stream"].wait_event(r["gate"])
        with torch.cuda.stream(r["stream"]):
            _owners_side(m, plan, True, ahead=True)


def _ahead_in_order(m, stage):
    """RowShard(route_ahead=False) with an announced next batch: what the second communicator's mode does ahead, on ONE
    communicator — every collective on the step's stream, at the same place of every rank's program:
      stage "counts" (right after this step's row exchanges are enqueued): the step's stream waits for the side stream's
        request sort (started at the head of the step: long done) and the next batch's count exchange is issued;
      stage "owners" (right before the dense all-reduce): the host takes the split sizes — the GPU passed the count exchange
        long ago and still has the backward queued behind it: no drained queue, which is what a host wait at the HEAD of a
        step costs — issues the next batch's id exchange, and the owners' sort of what arrived (local work) goes to the
        side stream, beside the weight gradients, the all-reduce and the sparse apply."""
    r = getattr(m, "_routed", None)
    if r is None or r.get("second", True):
        return
    plan = r["plan"]
    if stage == "counts":
        if "counts" in plan:
            if r["stream"] is not None:
                torch.cuda.current_stream().wait_stream(r["stream"])
            plan["pending"] = _comm(m).start_counts(plan.pop("counts"), plan["C"])
            if m.shard.sim_links and _comm(m).world == 1:
                _wait([_sim_exchange(m, 0)])
    elif "pending" in plan or "uoff" in plan:
        _owners_side(m, _finish_plan(m, plan), True, ahead=True, sort_stream=r["stream"])


def _take_route(m, ids, C):
    """The plan _route_ahead made for exactly this tensor, or a fresh one."""
    r, m._routed = getattr(m, "_routed", None), None
    if r is not None:
        if r["stream"] is not None:
            torch.cuda.current_stream().wait_stream(r["stream"])
        if C > 0 and r["ids"].data_ptr() == ids.data_ptr() and r["ids"].shape == ids.shape and r["version"] == ids._version and r["C"] == C:
            m._route_tag = r["tag"]
            m.route_ahead_hits = getattr(m, "route_ahead_hits", 0) + 1
            return r["plan"]
        if "pending" in r["plan"]:
            _finish_plan(m, r["plan"])                # (an announced batch that did not come: its exchange still completes)
    m._route_tag = ""
    return _route(m, ids, max(C, 1))


def _zero_off(m):
    z = m._ws.get("zero_off")
    if z is None:
        z = m._ws["zero_off"] = torch.zeros(m.F, dtype=torch.int64, device=m.device)
    return z


def _sharded_step(m, ids, labels, x_num, train, next_ids=None):
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
    # (an evaluation between two train steps plans into the first buffer set: a plan made ahead is dropped first)
    plan = _finish_plan(m, _take_route(m, ids, C if train else -1))
    announce = train and next_ids is not None and next_ids.shape == ids.shape and not getattr(m, "_capturing", False)
    head_ev = None
    if announce and m.device.type == "cuda":
        head_ev = torch.cuda.Event()
        head_ev.record()                    # (the previous step's work on this stream is done: the next batch's routing may start)
    slot, sorted_entry, seg = plan["slot"], plan["sorted_entry"], plan["seg"]
    uoff, roff, umid, rmid, sc0, rc0 = plan["uoff"], plan["roff"], plan["umid"], plan["rmid"], plan["sc0"], plan["rc0"]
    U, nr = uoff[-1], roff[-1]
    m.last_exchange = {"entries": n, "requests_sent": U, "requests_received": nr, "requests_to_self": sum(plan["n_self"])}

    # requests to their owners, the owners' bookkeeping for the WHOLE step (done ahead if the batch was announced), and
    # TF Adam's catch-up on the rows this step touches, before any of them is read
    _owners_side(m, plan, train)
    recv_ids, book = plan["recv_ids"], plan["book"]
    if book is not None and m.adam_rows and m.step > 0:
        # (the staleness order is made here, on the step's stream, and the wide records are replayed by the same call: made a
        # step ahead / on a side stream beside the row kernel, as in the single-GPU step, each costs the one-rank step 0.05 ms —
        # same-box A/B, tools/shard_opt_ab.sh: whatever runs beside the catch-up stretches it by as much)
        m._catchup(book[1], book[3], nr, defer=True)
    if announce:
        _route_ahead(m, next_ids, C, after=head_ev)        # (RowShard(route_ahead=False): its local work only)

    # Buffers whose size follows the batch's content (U distinct requests <= n entries; nr requests received: ~n for ids
    # spread over the ranks) are allocated at their CAPACITY the first time — n rows, and 2 n on the owner's side — so that
    # no later step allocates between two collectives (an allocation that misses torch's cache is a hipMalloc: a
    # device-wide synchronisation between kernels other ranks are waiting on).  Only a batch more than twice as
    # concentrated on this owner as a uniform one grows them again.
    cap_u, cap_r = max(n, 1), max(2 * n, nr, 1)
    # PACKED exchange (a model with both an embedding and a wide part): a request's row and its wide weight travel as ONE
    # record of E + 4 floats [row | weight | pad x 3] — and their gradients likewise — so a chunk costs one collective
    # per direction instead of two (xGMI collectives are latency-bound at these sizes: 4 -> 2 launches per chunk).  The
    # kernels take the record stride (include/mi355x_rec.h: out_stride / table_stride / rows_stride / grad_stride); the
    # *_rows / *_lin names below are then strided views of the record buffers.
    packed = bool(m.use_emb and m.use_linear and m.shard.packed)
    sim = m.shard.sim_links if (comm.world == 1 and m.device.type == "cuda") else None       # (tools/sim_ranks.py)
    EP = E + 4
    xs = EP if packed else 0                                     # the exchange buffers' record stride as the entries take it

    def rec(name, cap, cnt, need_rows, need_lin):
        """(record buffer or None, rows view, weight view) of `cnt` requests"""
        if packed:
            r = m._buf(name + "_rec", (cap, EP))[:cnt]
            return r, r[:, :E], r[:, E]
        return (None, m._buf(name + "_rows", (cap, E))[:cnt] if need_rows else None,
                m._buf(name + "_lin", (cap,))[:cnt] if need_lin else None)
    own_rec, own_rows, own_lin = rec("own", cap_r, nr, m.use_emb, m.use_linear)
    got_rec, got_rows, got_lin = rec("got", cap_u, max(U, 1), m.use_emb, m.use_linear)

    def gather(ids_, n_, rows_out, lin_out):
        if n_ > 0:
            k.mi_gather_rows(m.table if m.use_emb else None, m.lin_w if m.use_linear else None, ids_, n_, E,
                             rows_out if m.use_emb else None, lin_out if m.use_linear else None, m.ls, m.ts, xs)

    def serve(c):
        """owners gather chunk c's rows and send them back (their own requests: straight into the receive buffer);
        returns the exchange handles"""
        lo, mid, hi = roff[c], rmid[c], roff[c + 1]
        ulo, um, uhi = uoff[c], umid[c], uoff[c + 1]
        hs = []
        gather(recv_ids[lo:mid], mid - lo, own_rows[lo:mid] if m.use_emb else None, own_lin[lo:mid] if m.use_linear else None)
        if packed:
            hs.append(comm.all_to_all(got_rec[ulo:um], own_rec[lo:mid], sc0[c], rc0[c], True))
        else:
            if m.use_emb:
                hs.append(comm.all_to_all(got_rows[ulo:um], own_rows[lo:mid], sc0[c], rc0[c], True))
            if m.use_linear:
                hs.append(comm.all_to_all(got_lin[ulo:um], own_lin[lo:mid], sc0[c], rc0[c], True))
        gather(recv_ids[mid:hi], hi - mid, got_rows[um:uhi] if m.use_emb else None, got_lin[um:uhi] if m.use_linear else None)
        if sim is not None:
            hs.append(_sim_exchange(m, (hi - lo) * 4 * ((E if m.use_emb else 0) + (1 if m.use_linear else 0)),
                                    1 if (packed or not (m.use_emb and m.use_linear)) else 2))
        return hs

    # gradients: one row (record) per distinct request, in send order; the owner's side receives them in request order
    d_rec, d_rows, d_lin = rec("d", cap_u, max(U, 1), m.use_emb, m.use_linear) if train else (None, None, None)
    r_rec, r_rows, r_lin = rec("recv_d", cap_r, nr, m.use_emb, m.use_linear) if train else (None, None, None)
    logits_all = m._buf("logits_all", (B,)) if C > 1 else None
    loss_all = m._buf("loss_all", (1,)) if C > 1 else None
    acc = m._buf("d_grad_acc", (m.P,)) if (train and C > 1) else None
    slot2 = slot.view(B, F)
    zero_off = _zero_off(m)

    grad_h = []

    def send_gradients(c, d_concat, cc, dlogit, b0):
        """chunk c's entry gradients summed per distinct request, written at the request's send slot — those of the
        rank's own rows straight into the buffer its apply reads — and the exchange started.  (d_concat, sumv, dlogit
        belong to examples b0.. of the local batch.)"""
        ulo, um, uhi = uoff[c], umid[c], uoff[c + 1]
        lo, mid, hi = roff[c], rmid[c], roff[c + 1]
        for u0, cnt, o_rows, o_lin, row0 in ((ulo, um - ulo, d_rows, d_lin, 0),
                                            (um, uhi - um, r_rows[mid:hi] if m.use_emb else None,
                                             r_lin[mid:hi] if m.use_linear else None, um)):
            if cnt > 0:
                k.mi_entry_grads_segsum(got_rows if m.use_mf else None, seg, sorted_entry, u0, cnt,
                                        d_concat if m.use_emb else None, m.D, cc["sumv"] if m.use_mf else None,
                                        dlogit if m.use_mf else None, dlogit if m.use_linear else None, b0, F, E,
                                        o_rows, o_lin, row0, xs, xs)
        if packed:
            grad_h.append(comm.all_to_all(r_rec[lo:mid], d_rec[ulo:um], rc0[c], sc0[c], True))
        else:
            if m.use_emb:
                grad_h.append(comm.all_to_all(r_rows[lo:mid], d_rows[ulo:um], rc0[c], sc0[c], True))
            if m.use_linear:
                grad_h.append(comm.all_to_all(r_lin[lo:mid], d_lin[ulo:um], rc0[c], sc0[c], True))
        if sim is not None:
            grad_h.append(_sim_exchange(m, (uhi - ulo) * 4 * ((E if m.use_emb else 0) + (1 if m.use_linear else 0)),
                                        1 if (packed or not (m.use_emb and m.use_linear)) else 2))

    def backward(cc, dlogit, chunks, b0):
        """MLP backward of the examples b0..; the gradients of `chunks` leave as soon as the input layer's data gradient
        is enqueued — before its weight gradient, the step's largest GEMM, and under it (without an MLP there is only
        the FM term's / the wide part's dlogit)"""
        sent = []

        def go(d_concat):
            for c in chunks:
                send_gradients(c, d_concat, cc, dlogit, b0)
            sent.append(1)
        m._backward_dense(cc, dlogit, on_d_concat=go if m.use_dnn else None)
        if not sent:
            go(None)

    loss = logits = None
    if C > 1 and not m.shard.chunk_compute:
        # chunked exchanges, ONE forward / backward: every chunk's rows are served at once (gather, exchange, gather, ...),
        # the embedding-side kernels of chunk c wait for its exchange only, the MLP sees the whole batch
        handles = [serve(c) for c in range(C)]
        if train:
            _ahead_in_order(m, "counts")
        pieces = [(c * Bc, (c + 1) * Bc, (lambda c=c: _wait(handles[c]))) for c in range(C)]
        m._chunk = 0
        cc = m._forward(ids, x_num, train, (got_rows, got_lin, zero_off, slot2, xs or E, xs or 1), pieces=pieces)
        logits, loss, dlogit = m._head(cc, labels, train, global_batch=B * m.shard.world)
        if train:
            backward(cc, dlogit, range(C), 0)
        C_acc = 1
    else:
        C_acc = C
        rows_h = serve(0)
        if train:
            _ahead_in_order(m, "counts")
        for c in range(C):
            nxt = serve(c + 1) if c + 1 < C else []          # on the links while chunk c computes
            _wait(rows_h)
            rows_h = nxt
            sl = slice(c * Bc, (c + 1) * Bc)
            m._chunk = c
            cc = m._forward(ids[sl], None if x_num is None else x_num[sl], train, (got_rows, got_lin, zero_off, slot2[sl], xs or E, xs or 1))
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
            backward(cc, dlogit, [c], c * Bc)
            if C > 1:
                if c == 0:
                    acc.copy_(m.d_grad)
                else:
                    k.mi_axpy(acc, m.d_grad, m.P, 1.0)
    m._chunk = 0
    if C_acc > 1:
        logits, loss = logits_all, (loss_all if loss is not None else None)
    if not train:
        return loss, logits
    if C_acc > 1:
        m.d_grad.copy_(acc)
    _ahead_in_order(m, "owners")
    comm.all_reduce(m.d_grad)                                   # dense gradients: SUM over ranks
    if sim is not None:                                         # (a ring all-reduce moves 2 (N - 1) / N of the buffer per rank)
        _wait([_sim_exchange(m, 2 * m.d_grad.numel() * 4)])
    _wait(grad_h)
    r = getattr(m, "_routed", None)
    if r is not None and r["stream"] is not None:
        r["gate"] = torch.cuda.Event()
        r["gate"].record()                                      # (the next batch's owner-side work starts here: _own_ahead)
    if book is not None:
        bs_entry, buniq, bseg, bnum = book
        m._apply(buniq, bseg, bs_entry, bnum, nr, r_rows, r_lin, d_stride=xs)
    else:
        m._apply(None, None, None, None, 0, None, None)
    _own_ahead(m)
    return loss, logits


def sharded_eval_step(m, ids, labels, x_num):
    return _sharded_step(m, ids, labels, x_num, False)


def sharded_train_step(m, ids, labels, x_num, next_ids=None):
    """N-rank synchronous step.  Every rank must call it with the same local batch size — and, if it announces its
    next batch (next_ids: see _route_ahead), every rank must."""
    return _sharded_step(m, ids, labels, x_num, True, next_ids)
