#!/usr/bin/env python3
"""bench.py — examples/sec of one DeepFM train step (BASELINE.json config 3) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path over one synthetic batch that is already resident in
HBM: unique-row bookkeeping, lazy Adam catch-up, embedding gather + FM + wide linear, the
[512,256,128] MLP (fp32 accumulate / results; the matrix cores are fed fp16 high+low parts of the
operands with one power-of-two exponent per row, three MFMA products per fp32 product), sigmoid-CE
head, full backward, dense + sparse TF-form Adam.
`value` is the STEADY-STATE rate (timed after enough steps that >= 99 % of the rows have optimizer state
to catch up on); `cold_start` is the same K steps from a fresh model; `gemm_fp32` and
`lazy_adam_off_semantics` price the matrix-pipe path and the exact-Adam catch-up.
Workload (config.workload): trainers.deep_fm with --embedding-size 64 --hidden-units 512 256 128
--batch-size 65536 (dropout 0.1 = the CLI default), 26 categorical fields x 1,000,000 ids each
(Criteo-shaped), uniform ids, labels Bernoulli(0.25).  N > 1: one process per GPU, 65536 examples
per GPU (weak scaling; --scaling strong divides 65536 over the GPUs instead), embedding rows
sharded row % N with all-to-all over RCCL.
On one GPU the ids of step t + 1 are announced to step t (train_step(next_ids=...): an input pipeline holds them; their
sort then runs beside step t's catch-up; --no-presort for the plain sequence, same bits).  --force-shard [--chunks C]
runs the multi-GPU step with a one-rank RCCL group: what that path costs by itself, links aside.

Prints ONE JSON line on rank 0 with `roofline` (embedding gather kernel, HBM bound: algorithmic bytes read
and written per launch / launch time, timed live with HIP events on the launch stream) and `cpu_baseline`
(oracle/cpu_torch.py: the reference's TF graph restated on multi-threaded PyTorch-CPU at the full vocabulary,
timed on this host's cores for a few steps).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_MEASURED_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA (no sparsity)
SPLIT_PRODUCTS = 3            # f16x2 split: MFMA products issued per fp32 product

F, V, E, HIDDEN, B_FULL = 26, 1_000_000, 64, [512, 256, 128], 65536
DROPOUT = 0.1
POOL = 256                    # distinct synthetic batches at most
SEED = 20240521


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-dist", action="store_true", help="skip the short run on the other id distribution")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-MFMA and LazyAdam-semantics legs")
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform", help="id distribution")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (the measured path); gloo only rehearses the N>1 flow on one GPU")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 65536 examples per GPU (default); strong: 65536 examples over all GPUs")
    ap.add_argument("--force-shard", action="store_true",
                    help="N = 1 only: run the row-sharded multi-GPU step with a one-rank RCCL group (every all_to_all is a "
                         "copy to self) — what the sharded step's own machinery costs next to the single-GPU step")
    ap.add_argument("--chunks", type=int, default=None, help="pipeline depth of the row-sharded step (default: parallel.py's)")
    ap.add_argument("--no-presort", action="store_true",
                    help="do not announce the next batch's ids to train_step (its sort then runs at the head of the next step "
                         "instead of on a side stream beside this step's catch-up)")
    ap.add_argument("--gemm", choices=["f16x2", "bf16x3", "fp32"], default="f16x2", help="matrix-pipe path of the MLP GEMMs")
    return ap.parse_args()


def make_batches(n, gen, device, zipf, B):
    out = []
    for _ in range(n):
        if zipf:
            # Zipf s=1.05 truncated to V by inverse CDF on a uniform draw
            u = torch.rand(B, F, device=device, generator=gen, dtype=torch.float64)
            s = 1.05
            hmax = (V ** (1 - s) - 1) / (1 - s)
            ids = (((u * hmax) * (1 - s) + 1) ** (1 / (1 - s)) - 1).clamp_(0, V - 1).to(torch.int32)
        else:
            ids = torch.randint(0, V, (B, F), device=device, dtype=torch.int32, generator=gen)
        y = (torch.rand(B, device=device, generator=gen) < 0.25).to(torch.uint8)
        out.append((ids.contiguous(), y))
    return out


def kernel_ms(timers):
    """name -> (mean ms per launch, launches) from the HIP-event pairs the engine recorded."""
    out = {}
    for name, evs in timers.items():
        t = [s.elapsed_time(e) for s, e in evs]
        out[name] = (float(np.mean(t)), len(t), float(np.sum(t)))
    return out


def mlp_roofline(gemm, flops, gemm_ms):
    """All dense fwd/bwd GEMM launches of a step against the matrix-pipe peak of the path taken.
    `achieved` counts ALGORITHMIC fp32 flops (2·M·N·K per GEMM); the split paths issue several
    16-bit MFMA products per fp32 product, so their peak is the f16/bf16 dense peak divided by that."""
    ach = flops / (gemm_ms * 1e-3) / 1e12
    if gemm == "fp32":
        peak, kern, prod = MFMA_F32_PEAK_TFLOPS, "gemm_f32_k (v_mfma_f32_32x32x2_f32)", 1
    elif gemm == "bf16x3":
        peak, kern, prod = MFMA_F16_PEAK_TFLOPS / 6, "gemm_split_k<bf16x3> (v_mfma_f32_32x32x16_bf16, 6 products)", 6
    else:
        peak, kern, prod = MFMA_F16_PEAK_TFLOPS / SPLIT_PRODUCTS, ("gemm_pl_k (forward, data gradient: pre-split fp16 high/low planes, per-row exponents) + "
                                                                   "gemm_split_k<f16x2> (weight gradient); v_mfma_f32_32x32x16_f16, 3 products"), SPLIT_PRODUCTS
    return {"kernel": kern + "; all dense fwd/bwd launches incl. the N=1 logits layer (matrix-vector kernels)", "bound": "mfma",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s (fp32-equivalent)", "frac": ach / peak,
            "mfma_products_per_fp32_product": prod, "executed_mfma_tflops": ach * prod,
            "fp32_input_mfma_peak": MFMA_F32_PEAK_TFLOPS, "flops_per_step": flops, "gemm_ms_per_step": gemm_ms}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    job a share of the node, e.g. 16 of 256 CPUs: more threads than that only fight over the quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline():
    """The reference graph restated on PyTorch-CPU (oracle/cpu_torch.py: one table per field, materialised
    [B,d,E], un-fused layers, TF Adam's whole-table sweep) on this host, ALL threads, FULL vocabulary
    (26 x 1M rows: 20 GB of variables + slots), same B / F / E / hidden as the GPU run."""
    from oracle import cpu_torch as T
    cores = host_threads()
    torch.set_num_threads(cores)
    B = B_FULL
    t_init = time.perf_counter()
    st = T.State([V] * F, E, HIDDEN, seed=SEED)
    g = torch.Generator().manual_seed(SEED)
    draw = lambda: (torch.randint(0, V, (B, F), generator=g), (torch.rand(B, generator=g) < 0.25).float())
    log("cpu baseline: %d threads, variables built in %.1f s" % (cores, time.perf_counter() - t_init))
    tw = time.perf_counter()
    T.train_step(st, *draw())                         # warm-up (page faults, thread pools)
    log("cpu baseline: warm-up step %.1f s" % (time.perf_counter() - tw))
    steps, t0 = 0, time.perf_counter()
    while steps < 8 and (steps < 1 or time.perf_counter() - t0 < 15.0):
        ids, y = draw()
        T.train_step(st, ids, y)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "examples/sec", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "torch_threads": torch.get_num_threads(), "ms_per_step": dt / steps * 1e3,
            "sample": "CPU restatement of the reference graph (TensorFlow 1.12 unavailable): oracle/cpu_torch.py on PyTorch-CPU fp32, "
                      "B=%d F=%d E=%d hidden=%s, FULL vocabulary %d ids/field, TF Adam's whole-table sweep, fresh uniform batch "
                      "per step, %d steps after 1 warm-up (batch generation inside the timed loop: <1 %%), no dropout" %
                      (B, F, E, HIDDEN, V, steps)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (args.gpus, args.gpus))
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world == 1 and args.force_shard:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 300), rank=0,
                                world_size=1, **({"device_id": device} if args.backend == "nccl" else {}))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    from mi355x_rec.engine import DeepFM, OptimizerSpec
    B = B_FULL if args.scaling == "weak" else B_FULL // world      # examples per GPU and step
    shard = None
    if world > 1 or args.force_shard:
        from mi355x_rec.parallel import RowShard
        shard = RowShard(rank, world, chunks=args.chunks)
    m = DeepFM([V] * F, embedding_size=E, hidden_units=HIDDEN, dropout=DROPOUT,
               optimizer=OptimizerSpec("Adam", 0.001), device=device, seed=SEED, shard=shard, gemm=args.gemm)
    gen = torch.Generator(device=device)
    gen.manual_seed(SEED + rank)
    m.init_variables(gen, lin_scale=1e-3)
    if shard is not None:
        from mi355x_rec.parallel import broadcast_dense
        broadcast_dense(m)                       # replicated MLP must start identical on every rank
    # A fresh batch every step, as in training on a real dataset: which rows sit out how many steps
    # (the gap TF Adam's dense-equivalent sparse update is replayed over) then follows the id
    # distribution — uniform ids: geometric with mean 1M/65536 = 15 steps — instead of being pinned
    # to the period of a small rotating pool.  All batches are generated before any timed region
    # (256 at most: 1.7 GB of ids; a longer run cycles through them).
    # STEADY STATE: a step costs most once every row it touches has optimizer state to catch up on, so the
    # headline is timed only after a state-preparation phase of plain train steps that has given >= 99 % of
    # the rows of a field their first update (ln(100) / -ln(1 - B_global/V) = 68 steps at B = 65536).  The
    # same K steps timed from the freshly initialised model are reported as `cold_start` (round 1's headline).
    B_glob = B * world
    prep_total = int(math.ceil(math.log(100.0) / -math.log(1.0 - min(B_glob / V, 0.999))))
    n_cold = args.warmup + args.steps
    n_prep = max(0, prep_total - n_cold)
    batches = make_batches(min(2 * n_cold + n_prep, POOL), gen, device, args.dist == "zipf", B)
    cursor = [0]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The ids of step t + 1 are known while step t runs (all batches exist before the timed region; a real input pipeline
    # prefetches: tf.data in ml_100k.py:42-61).  They are announced to train_step, which sorts them on a side stream
    # beside step t's VALU-bound catch-up (~14 small launch-bound kernels that leave most of the chip idle) instead of at
    # the head of step t + 1.  All of a step's work still happens inside the timed region — K timed steps run K sorts —
    # and the results are bitwise those of the plain sequence (tests/test_hip_model.py).  Single GPU only.
    presort = [world == 1 and shard is None and not args.no_presort]

    def run(nsteps):
        out = None
        for _ in range(nsteps):
            ids, y = batches[cursor[0] % len(batches)]
            cursor[0] += 1
            out = m.train_step(ids, y, next_ids=batches[cursor[0] % len(batches)][0] if presort[0] else None)
        return out

    def timed(nsteps):
        """barrier + synchronize, nsteps steps, barrier + synchronize; max over ranks"""
        sync()
        t0 = time.perf_counter()
        out = run(nsteps)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, out

    log("model and %d batches ready" % len(batches))
    run(args.warmup)
    cold_dt, _ = timed(args.steps)
    log("cold start: %.3f ms/step" % (cold_dt / args.steps * 1e3))
    cold = {"value": world * B * args.steps / cold_dt, "unit": "examples/sec", "ms_per_step": cold_dt / args.steps * 1e3,
            "note": "the same %d steps after %d warm-up steps from a freshly initialised model (few rows have optimizer "
                    "state to catch up on yet)" % (args.steps, args.warmup)}
    run(n_prep)
    run(args.warmup)
    # The timed region: only the roofline kernel (the gather) is bracketed by HIP events — an event pair costs ~10 us of
    # stream time, and bracketing all ~50 launches of a step (as round 1 and the first builds of round 2 did) made
    # `value` 5 % worse than the step really is (rocprofv3 kernel trace: 0.22 ms of 10-us bubbles per step).  The
    # per-entry breakdown (`kernel_ms_per_step`, `roofline_mlp`) comes from the same K steps run once more right
    # after, fully instrumented; that second pass is not part of `value`.
    roof_keys = {"mi_embed_fm_planes_fwd", "mi_embed_fm_linear_fwd"}
    m.timers, m.k.timer_only = {}, roof_keys
    dt, (loss, _) = timed(args.steps)
    roof_timers, m.timers, m.k.timer_only = m.timers, None, None
    final_loss = float(loss.item())
    log("steady state: %.3f ms/step" % (dt / args.steps * 1e3))
    steps_before = cursor[0] - args.steps
    m.timers = {}
    idt, _ = timed(args.steps)
    timers, m.timers = m.timers, None
    timers.update(roof_timers)                           # the roofline kernel's launches: those of the timed region
    log("instrumented pass: %.3f ms/step" % (idt / args.steps * 1e3))

    # second distribution of SURVEY 8d (Criteo-like skew), a short run after the headline one: same
    # protocol (barrier + synchronize on both sides, max over ranks); reported beside `value`
    other = None
    if not args.no_second_dist:
        o_zipf = args.dist != "zipf"
        batches = make_batches(35, gen, device, o_zipf, B)
        cursor[0] = 0
        ow, os_ = 5, 30
        run(ow)
        odt, _ = timed(os_)
        other = {"data": "zipf s=1.05 ids" if o_zipf else "uniform ids", "value": world * B * os_ / odt,
                 "unit": "examples/sec", "ms_per_step": odt / os_ * 1e3, "steps": os_, "warmup": ow}
        batches = make_batches(24, gen, device, args.dist == "zipf", B)
        cursor[0] = 0

    # what the matrix-pipe path and the exact-Adam semantics cost (single GPU, short legs after the headline)
    extras = {}
    if world == 1 and not args.no_extras and not args.force_shard:
        # the same step replayed as ONE hipGraph launch (engine.graph_train_step): what the ~50 launch gaps cost.
        # Not the headline: the per-kernel HIP events of `roofline` / `kernel_ms_per_step` cannot sit inside a replay.
        def run_graph(nsteps):
            out = None
            for _ in range(nsteps):
                ids, y = batches[cursor[0] % len(batches)]
                cursor[0] += 1
                out = m.graph_train_step(ids, y)
            return out
        run_graph(4)                                      # eager step (sizes), capture, two replays
        sync()
        t0 = time.perf_counter()
        run_graph(10)
        sync()
        gdt = time.perf_counter() - t0
        extras["hip_graph"] = {"value": B * 10 / gdt, "unit": "examples/sec", "ms_per_step": gdt / 10 * 1e3,
                               "note": "the whole train step as one hipGraph launch (bitwise the same step; global step, lr_t and "
                                       "dropout seeds in a device-resident step state).  At this batch size the eager step is not "
                                       "launch-bound and the replay is SLOWER than `value`; the capture pays at small batches "
                                       "(tools/small_step_bench.py: B = 32, 0.34 -> 0.146 ms)"}
        m._graph = None
        if args.gemm != "fp32":
            keep = (m.gemm, m.planes, m.gather_mlp)
            m.gemm, m.planes, m.gather_mlp = "fp32", False, True
            run(3)
            fdt, _ = timed(10)
            extras["gemm_fp32"] = {"value": B * 10 / fdt, "unit": "examples/sec", "ms_per_step": fdt / 10 * 1e3,
                                   "note": "same steps with every MLP GEMM on the fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact "
                                           "products): the un-emulated number beside the headline's fp16 high/low operand split"}
            m.gemm, m.planes, m.gather_mlp = keep
        # LazyAdam semantics: rows that sat out are NOT replayed (no catch-up, no deferred slot decay).  NOT the
        # reference's tf.train.AdamOptimizer — here only to put a price on exactness.  Last leg: it
        # leaves the model's stamps stale.
        m.adam_rows, stamps, m.last_step = False, m.last_step, None
        run(3)
        ldt, _ = timed(10)
        extras["lazy_adam_off_semantics"] = {"value": B * 10 / ldt, "unit": "examples/sec", "ms_per_step": ldt / 10 * 1e3,
                                             "note": "LazyAdam semantics (touched rows only; NOT the reference's dense-equivalent "
                                                     "AdamOptimizer): the step without the catch-up replay"}
        m.adam_rows, m.last_step = True, stamps

    if rank == 0:
        km = kernel_ms(timers)
        ms_step = dt / args.steps * 1e3
        planes_gather = "mi_embed_fm_planes_fwd" in km
        g_ms = km["mi_embed_fm_planes_fwd" if planes_gather else "mi_embed_fm_linear_fwd"][0]
        # Algorithmic bytes per launch (per example): the kernel READS F rows of 4E bytes + F ids and
        # WRITES the input_layer concat (as fp16 high/low planes: 4 bytes per element, the bytes of the fp32
        # concat SURVEY counts) + sumv + fm + the example's exponent.  `achieved` counts both directions — the
        # kernel is HBM bound on their sum; `row_read_GBs` is the read side alone (what round 1's read-only
        # kernel reported: that form never wrote the concat, each of the layer-1 GEMMs re-gathered the rows).
        g_key = "mi_embed_fm_planes_fwd" if planes_gather else "mi_embed_fm_linear_fwd"
        Bl = B * args.steps / km[g_key][1]               # examples per launch (N > 1: one launch per chunk of the local batch)
        row_bytes = Bl * F * 4 * E
        wide_split = "mi_embed_fm_linear_fwd/wide" in km  # the wide part's 4-byte gathers run as their own kernel
        if planes_gather:
            gather_bytes = Bl * (F * (4 * E + 4) + F * 4 * E + 4 * E + 8)
        else:
            gather_bytes = row_bytes
        total_bytes = gather_bytes if planes_gather else Bl * (F * (4 * E + (4 if wide_split else 8)) + 4 * E + (4 if wide_split else 8))
        achieved = gather_bytes / (g_ms * 1e-3) / 1e9
        gemm_ms = sum(v[2] for k, v in km.items() if k in ("mi_dense_fwd", "mi_dense_fwd_gathered", "mi_dense_bwd_data", "mi_dense_bwd_weight",
                                                        "mi_dense_bwd_weight_gathered", "mi_dense_fwd_planes",
                                                        "mi_dense_bwd_data_planes", "mi_dense_bwd_weight_planes")) / args.steps
        dims = [F * E] + HIDDEN + [1]
        flops = 3 * 2 * B * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f)["embed_fm_planes_fwd_k" if planes_gather else "embed_fm_linear_fwd_k"]["bytes_per_launch"] if world == 1 else None
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "examples/sec DeepFM batch=65536 (full train step)",
            "value": world * B * args.steps / dt,
            "unit": "examples/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (%s ids, a fresh batch every step from a pool of up to %d, random-init weights)" % (args.dist, POOL),
            "config": {"workload": "config 3: trainers.deep_fm --embedding-size 64 --hidden-units 512 256 128 "
                                   "--batch-size 65536 --dropout 0.1, 26 fields x 1M ids (Criteo-shaped), Adam(1e-3)",
                       "per_gpu_batch": B, "global_batch": world * B, "fields": F, "vocab_per_field": V,
                       "embedding_size": E, "hidden_units": HIDDEN,
                       "gemm": {"f16x2": "fp32 GEMMs via scaled fp16 high+low operand split, fp32 accumulate",
                                "bf16x3": "fp32 GEMMs via 3-way bf16 operand split, fp32 accumulate",
                                "fp32": "fp32-input MFMA"}[args.gemm],
                       "parallelism": "dp%d + row-sharded embeddings (all-to-all)" % world if (world > 1 or args.force_shard) else "single GPU",
                       "input_pipeline": ("next batch's ids announced one step ahead (train_step(next_ids=...)): their sort runs on a "
                                          "side stream beside this step's catch-up" if presort[0] else "ids handed over step by step")},
            "roofline": {"kernel": ("embed_fm_planes_fwd_k: embedding gather + FM second order, writes the input_layer concat as fp16 "
                                    "high/low planes with one exponent per example (the operand of the layer-1 GEMMs); the wide part's "
                                    "4-byte gathers run as linear_only_fwd_k on a side stream" if planes_gather else
                                    "embed_fm_linear_fwd_k (embedding gather + FM second order; read-only form)"), "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of_measured_copy_peak": achieved / HBM_MEASURED_GBS,
                         "algorithmic_bytes_per_launch": int(gather_bytes), "examples_per_launch": int(Bl), "avg_launch_ms": g_ms,
                         "row_read_GBs": row_bytes / (g_ms * 1e-3) / 1e9, "row_read_frac": row_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_note": "HBM bytes/launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE), profiles/"},
            "roofline_mlp": mlp_roofline(args.gemm, flops, gemm_ms),
            "kernel_ms_per_step": {k: v[2] / args.steps for k, v in sorted(km.items())},
            "kernel_ms_note": "HIP-event pairs around every launch of a second pass over %d steps right after the timed "
                              "region (%.3f ms/step with the events' ~10-us bubbles); inside the timed region only the roofline "
                              "kernel is bracketed" % (args.steps, idt / args.steps * 1e3),
            "final_loss": final_loss,
            "state": {"steps_before_timed_region": steps_before, "state_prep_steps": n_prep,
                      "rows_with_optimizer_state": 1.0 - (1.0 - min(B_glob / V, 0.999)) ** steps_before,
                      "note": "value = steady state (>= 99 % of a field's rows have Adam state and a stamp to catch up from)"},
            "cold_start": cold,
        }
        out.update(extras)
        if other is not None:
            out["other_distribution"] = other
        if world == 1 and not args.no_cpu_baseline:
            log("extras done; cpu baseline")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
