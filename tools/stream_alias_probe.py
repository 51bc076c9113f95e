#!/usr/bin/env python3
"""Does the step's time depend on how many torch streams the PROCESS created before the engine made its own?

torch.cuda.Stream() hands out the streams of a pool of 32 per priority in turn; HIP spreads streams over a few hardware queues
(GPU_MAX_HW_QUEUES, default 4) and the null stream — the step's stream — has one of them.  Two streams on one hardware queue
run one after the other.  A user's script (or torch's ProcessGroupNCCL, which takes its stream from the same pool) that has
taken K streams before the engine takes its side streams shifts every one of them by K places.
usage: python tools/stream_alias_probe.py [--shard] K [K ...]      (one process per K: the pool's cursor cannot be reset)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))


def one(k, shard):
    import torch
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    F, V, E, B = 26, 1_000_000, 64, 65536
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hold = [torch.cuda.Stream(device=dev) for _ in range(k)]
    for s in hold:                                     # (used once: HIP creates the stream)
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
    sh = None
    if shard:
        import torch.distributed as dist
        from mi355x_rec.parallel import RowShard
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from mi355x_rec.parallel import rccl_options
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29800 + os.getpid() % 100), rank=0, world_size=1, device_id=dev,
                                **({} if os.environ.get("PROBE_RCCL_NORMAL") == "1" else rccl_options()))
        sh = RowShard(0, 1, route_ahead=False)

    gen = torch.Generator(device=dev); gen.manual_seed(1)
    m = DeepFM([V] * F, embedding_size=E, hidden_units=[512, 256, 128], dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001),
               device=dev, seed=1, shard=sh)
    m.init_variables(gen, lin_scale=1e-3)
    pool = [(torch.randint(0, V, (B, F), generator=gen, device=dev, dtype=torch.int32),
             (torch.rand(B, generator=gen, device=dev) < 0.25).to(torch.uint8)) for _ in range(40)]
    cur = 0

    def run(n):
        nonlocal cur
        for _ in range(n):
            ids, y = pool[cur % 40]
            cur += 1
            m.train_step(ids, y, next_ids=pool[cur % 40][0])
    run(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(20)
    torch.cuda.synchronize()
    print("%d streams taken before the engine's%s: %.3f ms / step" % (k, " (row-sharded, one rank)" if shard else "", (time.perf_counter() - t0) / 20 * 1e3), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--one":
        one(int(args[1]), args[2] == "1")
    else:
        shard = "--shard" in args
        for k in [a for a in args if a != "--shard"]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", k, "1" if shard else "0"], check=False)
