#!/usr/bin/env python3
"""A REHEARSAL of the N-rank row-sharded step on ONE GPU: real kernels, real dependency structure, MODELLED link time.

No multi-GPU box exists in this pool, so parallel.py's chunking (`_n_chunks`, `chunk_compute`) and DESIGN section 4's 8-GPU
estimate were arithmetic.  This tool runs config 3's step as ONE rank of an N-rank job would see it:

  * the rank's share of the tables — V / N rows per field (row r lives on rank r % N) — with its own batch of 65,536
    examples: the OWNER side (requests received, distinct rows to catch up and apply, staleness of those rows) then has the
    N-rank job's load exactly (ids uniform: N ranks x 65,536 examples over V rows = 65,536 examples over V / N); the
    REQUESTER side sees V / N ids per field instead of V: ~20 % fewer distinct requests than in the real job at N = 8;
  * a one-rank RCCL group (bench.py --force-shard's path: routing, owners' sort, gather_rows, slot gathers, segment sums,
    un-fused apply — everything but the wire), ONE communicator, the next batch announced;
  * every exchange followed by a spin kernel on the "communicator's" stream (parallel._sim_exchange) for
    latency + (N - 1) / N x bytes / link rate: row and gradient all-to-alls, id exchange, count exchange, dense all-reduce.

What it is good for: how much of the link time the schedule hides, which chunking wins at which link rate, a step-time
estimate with stated assumptions.  What it is not: a measurement of xGMI or RCCL (no contention for HBM or CUs by the
collectives' own kernels, no stragglers, no skew between ranks).  Nothing here feeds bench.py's `value`."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F, V, E, HIDDEN, B = 26, 1_000_000, 64, [512, 256, 128], 65536


def batches(n, vocab, gen, dev):
    out = []
    for _ in range(n):
        ids = torch.randint(0, vocab, (B, F), generator=gen, device=dev, dtype=torch.int32)
        y = (torch.rand(B, generator=gen, device=dev) < 0.25).to(torch.uint8)
        out.append((ids, y))
    return out


def timed(m, pool, cur, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ids, y = pool[cur[0] % len(pool)]
        cur[0] += 1
        m.train_step(ids, y, next_ids=pool[cur[0] % len(pool)][0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, nargs="+", default=[8])
    ap.add_argument("--link-gbs", type=float, nargs="+", default=[30.0, 45.0, 60.0],
                    help="what an all-to-all gets out of ONE xGMI link, GB/s per direction (peak 76.8; a rank of N uses its N - 1 links at once)")
    ap.add_argument("--latency-us", type=float, default=40.0, help="per collective")
    ap.add_argument("--chunks", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--chunk-compute", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-single", action="store_true", help="skip the single-GPU reference step")
    ap.add_argument("--link-priority", type=int, default=-1, help="priority of the simulated communicator's stream (bench.py / trainers._cli create the process group with is_high_priority_stream: RCCL's stream then lives in the high-priority pool of hardware queues, where the engine keeps nothing)")
    ap.add_argument("--route-ahead", type=int, choices=[0, 1], default=0, help="1: the next batch's count / id exchanges on a second communicator (its own modelled stream)")
    ap.add_argument("--trace", action="store_true", help="one step per setting: when each modelled exchange started and how long it took")
    a = ap.parse_args()
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    from mi355x_rec.parallel import RowShard
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    mk = lambda vocab, shard: DeepFM([vocab] * F, embedding_size=E, hidden_units=HIDDEN, dropout=0.1,
                                     optimizer=OptimizerSpec("Adam", 0.001), device=dev, seed=1, shard=shard)
    t1 = None
    if not a.no_single:
        m = mk(V, None)
        m.init_variables(gen, lin_scale=1e-3)
        pool = batches(48, V, gen, dev)
        cur = [0]
        timed(m, pool, cur, 75)                                  # state preparation: >= 99 % of the rows updated once
        t1 = timed(m, pool, cur, a.steps)
        print("single GPU, V = %d: %.3f ms / step" % (V, t1), flush=True)
        del m, pool
        torch.cuda.empty_cache()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from mi355x_rec.parallel import rccl_options
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29700 + os.getpid() % 200), rank=0, world_size=1, device_id=dev, **rccl_options())
    print("| ranks | GB/s per link | chunks | MLP per chunk | ms / step | modelled link time, ms | exposed, ms | x one GPU |")
    print("|---:|---:|---:|---|---:|---:|---:|---:|", flush=True)
    for N in a.world:
        vocab = V // N
        sh = RowShard(0, 1, route_ahead=bool(a.route_ahead), sim_links={"world": N, "gbs": a.link_gbs[0] * (N - 1), "latency_us": a.latency_us, "priority": a.link_priority})
        m = mk(vocab, sh)
        m.init_variables(gen, lin_scale=1e-3)
        pool = batches(32, vocab, gen, dev)
        cur = [0]
        sh.chunks, sh.chunk_compute = 1, False
        timed(m, pool, cur, 16)                                  # state preparation (a row is touched with p = 1 - exp(-N B / V) per step)
        for C in a.chunks:
            for cc in a.chunk_compute:
                sh.chunks, sh.chunk_compute = C, bool(cc)
                sh.sim_links = None                              # the same step with free links: its own overhead
                timed(m, pool, cur, a.warmup)
                t0 = timed(m, pool, cur, a.steps)
                for gbs in a.link_gbs:
                    sh.sim_links = {"world": N, "gbs": gbs * (N - 1), "latency_us": a.latency_us, "priority": a.link_priority}
                    timed(m, pool, cur, a.warmup)
                    m._ws["sim_link_us"] = 0.0
                    t = timed(m, pool, cur, a.steps)
                    link = m._ws.get("sim_link_us", 0.0) / a.steps * 1e-3
                    print("| %d | %.0f | %d | %s | %.3f | %.3f | %.3f | %s |" % (
                        N, gbs, C, "yes" if cc else "no", t, link, t - t0, ("%.2f" % (N * t1 / t)) if t1 else "-"), flush=True)
                    if a.trace:
                        torch.cuda.synchronize()
                        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        m._ws["sim_trace"] = []
                        s0.record()
                        timed(m, pool, cur, 1)
                        s1.record(); torch.cuda.synchronize()
                        tr, m._ws["sim_trace"] = m._ws["sim_trace"], None
                        print("    one step: %.3f ms; exchanges (start ms, took ms, modelled ms, MB): %s" % (
                            s0.elapsed_time(s1), "  ".join("%.2f/%.2f/%.2f/%.0f" % (s0.elapsed_time(e0), e0.elapsed_time(e1), us * 1e-3, nb / 1e6)
                                                           for e0, e1, us, nb in tr)), flush=True)
                print("| %d | free | %d | %s | %.3f | 0 | 0 | %s |" % (N, C, "yes" if cc else "no", t0, ("%.2f" % (N * t1 / t0)) if t1 else "-"), flush=True)
        del m, pool
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
