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

`config.catchup` names the lazy-Adam replay the headline runs ("bounded": every variable within 3 ulp + 2e-6 of the movement
the replay covers, tests/test_hip_kernels.py::test_bounded_catchup_stays_within_its_bound_of_the_sweep; --catchup exact = TF's bits); the
other mode is timed as the extra `catchup_exact` / `catchup_bounded`.  `configs` carries the other BASELINE.json
workloads as short legs on the same GPU (config 2: us per step eager and as one hipGraph launch; config 4 at the CLI
defaults and at config 3's sizes; one rank's share of config 5), `roofline_sparse_apply` and `roofline_catchup` the two
other bandwidth / issue-bound kernels of the step.

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
SUSTAINED_F16_EXECUTED_TFLOPS = 1050.0   # executed f16 MFMA TFLOP/s this chip sustains on live (random) operands: power-limited, profiles/r03_gemm_power_limit.md

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
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-MFMA, other-catch-up-mode and LazyAdam-semantics legs (and the configs legs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs on the other BASELINE.json configs (2, 4, 5-share)")
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
    ap.add_argument("--chunk-compute", type=int, choices=[0, 1], default=None,
                    help="row-sharded step: 1 = every chunk runs its own forward / backward, 0 = only the exchanges and the embedding-side "
                         "kernels are chunked (default: parallel.RowShard's — 1 since the rehearsal of profiles/r05_sim_ranks.md)")
    ap.add_argument("--no-presort", action="store_true",
                    help="do not announce the next batch's ids to train_step (its sort then runs at the head of the next step "
                         "instead of on a side stream beside this step's catch-up)")
    ap.add_argument("--gemm", choices=["f16x2", "bf16x3", "fp32"], default="f16x2", help="matrix-pipe path of the MLP GEMMs")
    ap.add_argument("--catchup", choices=["exact", "bounded"], default="bounded",
                    help="lazy Adam replay of the steps a row sat out: TF's fp32 op sequence bit for bit, or the bounded-error "
                         "form (every variable within 3 ulp + 2e-6 of the replayed movement of the sweep; include/mi355x_rec.h MI_CATCHUP_BOUNDED)")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="NAME=0|1",
                    help="A/B runs: set a scheduling attribute of engine.DeepFM (WSPLIT_AHEAD, LIN_SIDE, BYGAP_AHEAD, WGRAD_BATCH, ...)")
    ap.add_argument("--route-ahead", type=int, choices=[0, 1], default=None,
                    help="row-sharded step: 0 = the whole step on ONE RCCL communicator (no routing of the next batch ahead on a "
                         "second one) — the default under a launcher for N > 1: two communicators have never run on more than one "
                         "GPU; 1 = the next batch routed ahead on a second communicator (faster by ~0.25 ms with one rank; "
                         "--force-shard's default).  `python bench.py --gpus N` without a launcher tries both, in fresh processes")
    ap.add_argument("--cpu-baseline-json", default=None, help=argparse.SUPPRESS)     # (parent -> rank 0: the baseline it timed before starting the ranks)
    ap.add_argument("--packed-exchange", type=int, choices=[0, 1], default=None,
                    help="row-sharded step: 1 = rows + wide weights (and their gradients) travel as one record of E + 4 floats per "
                         "request: one collective per chunk and direction instead of two (default: parallel.RowShard's, 0)")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="seconds after which a stuck collective aborts the process (non-zero exit, rank and collective named by "
                         "torch.distributed's watchdog)")
    return ap.parse_args()


def _run_rank_set(n, extra, label, timeout_s):
    """N fresh rank processes through torch.distributed.run — children of this process, which never touches the GPU.
    Returns (exit status, rank 0's JSON line or None)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + extra
    log("%s: starting %d ranks: %s" % (label, n, " ".join(cmd)))
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout_s)
        rc, text = p.returncode, p.stdout
    except subprocess.TimeoutExpired as e:
        rc, text = -9, (e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or ""))
    line = None
    for ln in text.splitlines():
        if ln.startswith("{"):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    return rc, line


def merge_mode_lines(line1, line2, rc2):
    """The record of a parent-mode run: the faster of the two sets' lines, both modes' numbers in config.communicator_modes;
    the second set's failure (no line) named there.  Legs only the first set ran (cpu_baseline, other_distribution) are
    carried over when the second set's line wins."""
    d1 = json.loads(line1)
    modes = {"one_communicator": {"value": d1["value"], "ms_per_step": d1["ms_per_step"]}}
    best = d1
    if line2:
        d2 = json.loads(line2)
        modes["route_ahead_second_communicator"] = {"value": d2["value"], "ms_per_step": d2["ms_per_step"]}
        if d2["value"] > d1["value"]:
            for k_ in ("cpu_baseline", "other_distribution"):
                if k_ in d1 and k_ not in d2:
                    d2[k_] = d1[k_]
            best = d2
    else:
        modes["route_ahead_second_communicator"] = {"failed": "exit status %d, no line (a collective timeout ends the ranks non-zero)" % rc2}
    best["config"]["communicator_modes"] = modes
    best["config"]["communicator_mode_of_value"] = "one_communicator" if best is d1 else "route_ahead_second_communicator"
    return best


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher.  This process has not touched the GPU and never will:
      1. the CPU baseline, timed HERE before any rank exists (the ranks' host threads then do not compete with it), handed to
         rank 0 as a file;
      2. a set of fresh ranks in the conservative mode — every collective of a step on ONE communicator (--route-ahead 0): its
         line is the record if nothing else works;
      3. a second set of fresh ranks with the next batch routed ahead on a second communicator (--route-ahead 1).  If it
         finishes, the faster of the two lines is printed, with both modes' numbers in config; if it dies (a collective
         timeout ends the ranks non-zero) the first line is printed with the failure named, and the exit status is 0."""
    import tempfile
    n = args.gpus
    extra = []
    if not args.no_cpu_baseline:
        log("cpu baseline (before the ranks start)")
        cb = cpu_baseline()
        f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(cb, f)
        f.close()
        extra = ["--cpu-baseline-json", f.name]
    budget = 3.0 * args.collective_timeout + 900.0
    if args.route_ahead is not None:                       # the caller chose a mode: one set
        rc, line = _run_rank_set(n, extra, "ranks", budget)
        if line:
            print(line)
        raise SystemExit(rc if not line else 0)
    rc1, line1 = _run_rank_set(n, extra + ["--route-ahead", "0"], "set 1 (one communicator)", budget)
    if not line1:
        raise SystemExit(rc1 or 1)
    rc2, line2 = _run_rank_set(n, extra + ["--route-ahead", "1", "--no-second-dist", "--no-cpu-baseline"], "set 2 (routing ahead on a second communicator)", budget)
    best = merge_mode_lines(line1, line2, rc2)
    print(json.dumps(best))
    raise SystemExit(0)


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
        peak, kern, prod = MFMA_F16_PEAK_TFLOPS / SPLIT_PRODUCTS, ("gemm_pl_k (forward, data gradient) + wgrad_pl_k (weight gradient): operands as pre-split "
                                                                   "fp16 high/low planes with per-row exponents; v_mfma_f32_32x32x16_f16, 3 products"), SPLIT_PRODUCTS
    return {"kernel": kern + "; all dense fwd/bwd launches incl. the N=1 logits layer (matrix-vector kernels) AND the path's own "
                             "overhead launches (weight split, abs-max, row splits, slab folds)", "bound": "mfma",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s (fp32-equivalent)", "frac": ach / peak,
            "mfma_products_per_fp32_product": prod, "executed_mfma_tflops": ach * prod,
            "fp32_input_mfma_peak": MFMA_F32_PEAK_TFLOPS, "flops_per_step": flops, "gemm_ms_per_step": gemm_ms,
            # (f16 split paths) what this chip sustains on live operands under its power management — a constant from
            # profiles/r03_gemm_power_limit.md (tools/probe/gemm4w_probe.hip: the same loop at 82 % matrix duty, the clock
            # falls to 1.3 GHz; 2.0 GHz and 1.67 PF on all-zero operands), not a measurement of this run
            "sustained_executed_peak_tflops": SUSTAINED_F16_EXECUTED_TFLOPS if prod in (3, 6) else None,
            "frac_of_sustained": (ach * prod / SUSTAINED_F16_EXECUTED_TFLOPS) if prod in (3, 6) else None}


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
    if int(os.environ.get("RANK", "0")) == 0:
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


ML100K_VOCAB = [2] * 19 + [1000, 2000, 50, 1000, 7, 8, 3]


def _time_steps(step, batches, warm, n, sync):
    for i in range(warm):
        step(*batches[i % len(batches)])
    sync()
    t0 = time.perf_counter()
    out = None
    for i in range(n):
        out = step(*batches[i % len(batches)])
    sync()
    return (time.perf_counter() - t0) / n, out


def other_configs(device, sync):
    """The other BASELINE.json workloads as short legs on this GPU: step time and examples/sec each,
    4 rotating batches (so little catch-up work: these are shape legs, not steady-state headlines)."""
    from mi355x_rec.engine import DeepFM, OptimizerSpec
    out = {}
    g = torch.Generator(device=device)
    g.manual_seed(SEED + 7)

    def batches(vocab, B, n_num, n=4):
        bs = []
        for _ in range(n):
            ids = torch.stack([torch.randint(0, v, (B,), device=device, generator=g) for v in vocab], 1).to(torch.int32).contiguous()
            y = (torch.rand(B, device=device, generator=g) < 0.25).to(torch.uint8)
            x = torch.log1p(torch.empty(B, n_num, device=device).exponential_(generator=g)) if n_num else None
            bs.append((ids, y, x) if n_num else (ids, y))
        return bs

    # config 2: trainers.deep_fm defaults on the MovieLens schema — launch-bound: us per step
    m = DeepFM(ML100K_VOCAB, embedding_size=4, hidden_units=[16, 16], dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001), device=device)
    m.init_variables(g, lin_scale=1e-3)
    bs = batches(ML100K_VOCAB, 32, 0, n=1)
    te, _ = _time_steps(m.train_step, bs, 20, 200, sync)
    tg, _ = _time_steps(m.graph_train_step, bs, 20, 200, sync)
    out["c2"] = {"workload": "config 2: trainers.deep_fm defaults (E=4, hidden [16,16], B=32, dropout 0.1, Adam), 26 MovieLens fields "
                             "(4,106 rows), synthetic ids", "us_per_step_eager": te * 1e6, "us_per_step_hip_graph": tg * 1e6,
                 "examples_per_sec_hip_graph": 32 / tg}
    del m
    # config 4: Wide&Deep (linear_deep), 26 fields x 1M ids + 13 dense columns, Ftrl (wide) + Adagrad (deep), SUM loss
    for key, Ec, hid in (("c4_defaults", 4, [16, 16]), ("c4_c3sizes", E, HIDDEN)):
        vocab = [V] * F
        m = DeepFM(vocab, n_numeric=13, numeric="raw", embedding_size=Ec, hidden_units=hid, use_mf=False, dropout=0.1,
                   optimizer=OptimizerSpec("Adagrad", 0.05), linear_optimizer=OptimizerSpec("Ftrl", 0.1961), reduction="sum", device=device)
        m.init_variables(g, lin_scale=1e-3)
        t, (loss, _) = _time_steps(m.train_step, batches(vocab, B_FULL, 13), 4, 12, sync)
        out[key] = {"workload": "config 4: trainers.linear_deep (DNNLinearCombined: Ftrl + Adagrad, SUM loss) B=65536, 26 fields x 1M ids + "
                                "13 dense columns, E=%d hidden %s, one GPU's step" % (Ec, hid),
                    "ms_per_step": t * 1e3, "examples_per_sec": B_FULL / t, "final_loss": float(loss.item())}
        del m
        torch.cuda.empty_cache()
    # config 5: one rank's share — 40 fields x 1.25M rows (50M rows x E=128 = 25.6 GB, 76.8 GB with Adam slots), B = 131072 / 8
    vocab = [1_250_000] * 40
    m = DeepFM(vocab, embedding_size=128, hidden_units=HIDDEN, dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001), device=device,
               catchup="bounded")
    m.init_variables(g, lin_scale=1e-3)
    t, (loss, _) = _time_steps(m.train_step, batches(vocab, 16384, 0), 4, 12, sync)
    out["c5_rank_share"] = {"workload": "config 5, ONE rank's share as a single-GPU step (no exchange): 40 fields x 1.25M rows (a 25.6 GB table, "
                                        "76.8 GB with Adam slots), E=128, hidden [512,256,128], B=16384 (131072 / 8)",
                            "ms_per_step": t * 1e3, "examples_per_sec": 16384 / t, "final_loss": float(loss.item()),
                            "hbm_allocated_GB": torch.cuda.max_memory_allocated() / 1e9}
    del m
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            spawn_ranks(args)                        # (never returns)
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import datetime
    # a stuck collective must END the process (non-zero, the rank and the collective in the watchdog's message), not hang
    # the node: a finite timeout on the group, and RCCL's watchdog told to tear the process down when it fires
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    tmo = datetime.timedelta(seconds=args.collective_timeout)
    from mi355x_rec.parallel import rccl_options
    if world == 1 and args.force_shard:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 300), rank=0,
                                world_size=1, timeout=tmo, **(dict(device_id=device, **rccl_options(tmo)) if args.backend == "nccl" else {}))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo, **rccl_options(tmo))    # (RCCL's stream: the high-priority pool of hardware queues, where the engine keeps nothing)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    from mi355x_rec.engine import DeepFM, OptimizerSpec
    if os.environ.get("MI_TUNING_LIB"):          # A/B runs of the tools' build (make -C csrc tuning: it reads the MI_* switches)
        from mi355x_rec import _lib
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libmi355x_rec_tuning.so")
    for opt in args.engine_opt:
        name, _, val = opt.partition("=")
        if not hasattr(DeepFM, name):
            raise SystemExit("--engine-opt %s: engine.DeepFM has no attribute %s" % (opt, name))
        setattr(DeepFM, name, type(getattr(DeepFM, name))(int(val)))
    B = B_FULL if args.scaling == "weak" else B_FULL // world      # examples per GPU and step
    shard = None
    if world > 1 or args.force_shard:
        from mi355x_rec.parallel import RowShard
        # N > 1 under a launcher: ONE communicator unless asked otherwise (the conservative mode for hardware this path has never
        # run on); one rank (--force-shard): parallel.RowShard's default, the routing ahead on a second communicator
        ra = args.route_ahead if args.route_ahead is not None else (0 if world > 1 else None)
        shard = RowShard(rank, world, chunks=args.chunks,
                         chunk_compute=None if args.chunk_compute is None else bool(args.chunk_compute),
                         route_ahead=None if ra is None else bool(ra),
                         **({} if args.packed_exchange is None else {"packed": bool(args.packed_exchange)}))
    m = DeepFM([V] * F, embedding_size=E, hidden_units=HIDDEN, dropout=DROPOUT,
               optimizer=OptimizerSpec("Adam", 0.001), device=device, seed=SEED, shard=shard, gemm=args.gemm,
               catchup=args.catchup)
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
    # (row-sharded step: the announced ids start the next batch's ROUTING — sorts, count exchange, split sizes to the host —
    # beside this step's sparse apply instead: parallel._route_ahead)
    presort = [not args.no_presort]

    def run(nsteps):
        out = None
        for _ in range(nsteps):
            ids, y = batches[cursor[0] % len(batches)]
            cursor[0] += 1
            out = m.train_step(ids, y, next_ids=batches[cursor[0] % len(batches)][0] if presort[0] else None)
        return out

    enqueue_s = [0.0]

    def timed(nsteps):
        """barrier + synchronize, nsteps steps, barrier + synchronize; max over ranks"""
        sync()
        t0 = time.perf_counter()
        out = run(nsteps)
        enqueue_s[0] = time.perf_counter() - t0          # (host time to enqueue the steps: far below dt = the host runs ahead)
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
    log("steady state: %.3f ms/step (host enqueue %.3f ms/step)" % (dt / args.steps * 1e3, enqueue_s[0] / args.steps * 1e3))
    steps_before = cursor[0] - args.steps
    m.timers = {}
    idt, _ = timed(args.steps)
    timers, m.timers = m.timers, None
    timers.update(roof_timers)                           # the roofline kernel's launches: those of the timed region
    log("instrumented pass: %.3f ms/step" % (idt / args.steps * 1e3))
    # The roofline kernel once more WITHOUT the next batch's sort running beside it: in the timed region that sort (a side
    # stream's small kernels, started beside the catch-up) ends under the gather and shares HBM with it; a few steps with
    # the batches handed over one by one (the sort then runs at the head of the step, on the step's stream) give the
    # kernel's own launch time.  Not part of `value`.
    alone_ms = None
    if presort[0] and shard is None:
        presort[0] = False
        run(3)
        m.timers, m.k.timer_only = {}, roof_keys
        timed(min(args.steps, 20))
        at, m.timers, m.k.timer_only = m.timers, None, None
        presort[0] = True
        ak = kernel_ms(at)
        alone_ms = next((v[0] for k, v in ak.items() if k in roof_keys), None)

    # What the sparse kernels of a steady-state step work on (a statistic for their rooflines, taken outside every timed
    # region with torch ops on the NEXT batch): distinct rows U, and over how many steps each has to be replayed.
    sparse_stats = None
    if shard is None and m.last_step is not None:
        ids_n = batches[cursor[0] % len(batches)][0]
        rows_n = torch.unique(ids_n.long() + m.field_off[None, :])
        st_n = m.last_step[rows_n]
        gaps_n = (m.step - st_n)[st_n > 0].double()
        sparse_stats = {"unique_rows": int(rows_n.numel()), "rows_with_state": int(gaps_n.numel()),
                        "mean_replayed_steps": float(gaps_n.mean().item()) if gaps_n.numel() else 0.0,
                        "element_steps": float(gaps_n.sum().item()) * E}
        del rows_n, st_n, gaps_n

    # N > 1: the OTHER scaling mode as a short leg in the same line (SURVEY 8e asks for both, labelled): weak = 65536
    # examples per GPU, strong = 65536 examples over all GPUs.  Same protocol (barrier + synchronize, max over ranks).
    other_scaling = None
    if world > 1:
        o_mode = "strong" if args.scaling == "weak" else "weak"
        Bo = B_FULL // world if o_mode == "strong" else B_FULL
        keep_b, keep_c = batches, cursor[0]
        batches = make_batches(24, gen, device, args.dist == "zipf", Bo)
        cursor[0] = 0
        run(4)
        sdt_, _ = timed(10)
        other_scaling = {"scaling": o_mode, "per_gpu_batch": Bo, "global_batch": Bo * world, "value": world * Bo * 10 / sdt_,
                         "unit": "examples/sec", "ms_per_step": sdt_ / 10 * 1e3, "steps": 10, "warmup": 4,
                         "note": "short leg on a fresh pool of 24 batches right after the headline (more catch-up work than in "
                                 "steady state)"}
        batches, cursor[0] = keep_b, keep_c
        run(2)

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
        # A/B/A on the same batch pool (right after a pool switch the catch-up has more to replay than in the headline and
        # gets lighter as the pool cycles: the replay is timed BETWEEN two legs of the plain eager sequence — no announced
        # next batch: a captured step cannot use one)
        keep_pre, presort[0] = presort[0], False
        run(3)
        e1dt, _ = timed(10)
        run_graph(4)                                      # eager step (sizes), capture, two replays
        sync()
        t0 = time.perf_counter()
        run_graph(10)
        sync()
        gdt = time.perf_counter() - t0
        run(2)
        e2dt, _ = timed(10)
        edt = 0.5 * (e1dt + e2dt)
        presort[0] = keep_pre
        extras["hip_graph"] = {"value": B * 10 / gdt, "unit": "examples/sec", "ms_per_step": gdt / 10 * 1e3,
                               "eager_same_leg_ms_per_step": edt / 10 * 1e3,
                               "note": "the whole train step as one hipGraph launch (bitwise the same step; global step, lr_t and "
                                       "dropout seeds in a device-resident step state), timed back to back with the plain eager "
                                       "sequence on the same batch pool, eager / replay / eager (eager_same_leg = mean of the two; right "
                                       "after a pool switch: more catch-up work than `value`).  At this batch size nothing is "
                                       "launch-bound: the replay pays two input copies into the captured buffers and has no weight split "
                                       "ahead; the capture pays at small batches (configs.c2: B = 32)"}
        m.drop_graphs()
        if args.gemm != "fp32":
            keep = (m.gemm, m.planes, m.gather_mlp)
            m.gemm, m.planes, m.gather_mlp = "fp32", False, True
            run(3)
            fdt, _ = timed(10)
            extras["gemm_fp32"] = {"value": B * 10 / fdt, "unit": "examples/sec", "ms_per_step": fdt / 10 * 1e3,
                                   "note": "same steps with every MLP GEMM on the fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact "
                                           "products): the un-emulated number beside the headline's fp16 high/low operand split"}
            m.gemm, m.planes, m.gather_mlp = keep
        # the other catch-up mode, same steps (A/B partner of nothing: run on the fresh 24-batch pool like the legs above)
        other_mode = "exact" if m.catchup == "bounded" else "bounded"
        keep_mode, m.catchup = m.catchup, other_mode
        run(3)
        cdt, _ = timed(10)
        m.catchup = keep_mode
        run(3)
        sdt, _ = timed(10)
        extras["catchup_" + other_mode] = {
            "value": B * 10 / cdt, "unit": "examples/sec", "ms_per_step": cdt / 10 * 1e3,
            "same_leg_in_headline_mode_ms_per_step": sdt / 10 * 1e3,
            "note": ("the same steps with the lazy Adam replay bit-exact with TF's dense-equivalent sweep (exactly rounded sqrt and divide "
                     "per element and replayed step)" if other_mode == "exact" else
                     "the same steps with the bounded-error replay (MI_CATCHUP_BOUNDED)") +
                    "; timed back to back with 10 steps in the headline's mode on the same batch pool"}
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

    mode_catchup = m.catchup
    configs = None
    if world == 1 and not args.no_extras and not args.force_shard and not args.no_configs:
        log("the other BASELINE configs")
        del m
        batches = None
        torch.cuda.empty_cache()
        configs = other_configs(device, sync)

    if world > 1:
        # every collective of the run is done: the group goes away HERE, before rank 0 assembles the line (and times the CPU
        # baseline for half a minute) — no rank waits inside a collective meanwhile
        dist.barrier()
        dist.destroy_process_group()
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
        gemm_keys = ("mi_dense_fwd", "mi_dense_fwd_gathered", "mi_dense_bwd_data", "mi_dense_bwd_weight", "mi_dense_bwd_weight_gathered",
                     "mi_dense_fwd_planes", "mi_dense_bwd_data_planes", "mi_dense_bwd_weight_planes",
                     # (round 4: the logits layer's forward and backward run inside the fused logits + head launch — counted
                     # here in full, the head's own ~10 us included)
                     # ... and, where the last hidden layer runs there too (mi_hidden_logits_head_fused), that launch)
                     "mi_logits_head_fused", "mi_hidden_logits_head_fused", "mi_dense_bwd_weight_planes_batch")
        # (the planes path's own overhead launches count against it: they exist only because of it)
        gemm_overhead_keys = ("mi_dense_bwd_data_vec_planes", "mi_split_weights", "mi_absmax", "mi_split_rows")
        gemm_ms = sum(v[2] for k, v in km.items() if k in gemm_keys + gemm_overhead_keys) / args.steps
        gemm_overhead_ms = sum(v[2] for k, v in km.items() if k in gemm_overhead_keys) / args.steps
        dims = [F * E] + HIDDEN + [1]
        flops = 3 * 2 * B * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        # PMC traffic: NOT measured by this run (hardware counters need rocprofv3 around the process) — the committed
        # per-launch byte counts of the same kernels on the same workload, profiles/traffic.json (its _comment names the
        # profile each entry comes from)
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            pass
        tr = lambda name: pmc.get(name, {}).get("bytes_per_launch")
        traffic = tr("embed_fm_planes_fwd_k" if planes_gather else "embed_fm_linear_fwd_k")
        if traffic is not None and int(Bl) != B_FULL:
            traffic = traffic * Bl / B_FULL               # (counted on the single-GPU launch of 65536 examples: scaled to this launch's)
        # the two other bandwidth / issue-bound kernels of the step (lazy sparse Adam = 28E + 8 bytes per distinct row)
        roof_apply = roof_catchup = None
        if sparse_stats is not None and "mi_sparse_apply_fused" in km:
            U = sparse_stats["unique_rows"]
            a_ms = km["mi_sparse_apply_fused"][0]
            a_bytes = U * (28 * E + 8)
            roof_apply = {"kernel": "sparse_apply_k<fused> (+ sparse_apply_long_k): duplicate-summing of the entry gradients rebuilt from "
                                    "d_concat / sumv / dlogit, TF-form Adam on the distinct rows of the batch, wide-part record, stamps",
                          "bound": "hbm", "achieved": a_bytes / (a_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": a_bytes / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": a_ms, "unique_rows": U,
                          "algorithmic_bytes_per_launch": int(a_bytes),
                          "algorithmic_bytes_note": "SURVEY 8d: per distinct row 4E (gradient) + 24E (w, m, v read and written) + 8 = 28E + 8 = %d B" % (28 * E + 8),
                          "traffic": tr("sparse_apply_k"),
                          "traffic_note": "PMC bytes per launch (profiles/traffic.json); above the algorithmic bytes by the per-ENTRY reads of "
                                          "sumv (served by L2 / Infinity Cache, counted at the fabric) and the wide-part records"}
        if sparse_stats is not None and "mi_sparse_catchup" in km and sparse_stats["rows_with_state"]:
            # the ROW kernel's launches only: the wide part's call (engine._catchup, LIN_SIDE: its own stream) is timed under its
            # own key — round 4 divided the rows' bytes by the mean over BOTH kernels' launches
            c_ms = km["mi_sparse_catchup"][0]
            Us = sparse_stats["rows_with_state"]
            wide_apart = "mi_sparse_catchup/wide" in km
            # w, m, v read + w written (deferred slots); + the 16-byte wide record read and written when ONE call does both
            c_bytes = Us * (16 * E + (0 if wide_apart else 32))
            cyc = 236.0 if mode_catchup == "exact" else 73.0
            issue_ms = sparse_stats["element_steps"] / 256.0 * cyc / 1024.0 / 2.1e9 * 1e3
            roof_catchup = {"kernel": "sparse_catchup_%s: lazy replay of TF Adam's dense-equivalent update on the rows "
                                      "about to be read (w only: the apply decays m, v)" % ("bounded_k" if mode_catchup == "bounded" else "k"),
                            "mode": mode_catchup, "bound": "hbm" if mode_catchup == "bounded" else "valu issue",
                            "achieved": c_bytes / (c_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": c_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": c_ms,
                            "algorithmic_bytes_per_launch": int(c_bytes),
                            "algorithmic_bytes_note": "per row with state: w, m, v read (12E) + w written (4E) = 16E" +
                                                      (" (the wide part's records: roofline_catchup.wide)" if wide_apart else " + the 16-byte wide record read and written = 16E + 32"),
                            "wide": None if not wide_apart else {
                                "kernel": "catchup_lin_k: the wide part's {w, m, v, stamp} records of the same rows, on the wide part's stream beside the row kernel",
                                "avg_launch_ms": km["mi_sparse_catchup/wide"][0], "algorithmic_bytes_per_launch": int(Us * 32),
                                "achieved": Us * 32 / (km["mi_sparse_catchup/wide"][0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": Us * 32 / (km["mi_sparse_catchup/wide"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": tr("catchup_lin_k"),
                                "note": "16-byte records scattered over 128-byte lines: the traffic is 8x the algorithmic bytes by construction"},
                            "rows_with_state": Us, "mean_replayed_steps": sparse_stats["mean_replayed_steps"],
                            "element_steps_per_launch": sparse_stats["element_steps"],
                            "issue_model": {"cycles_per_wave_and_replayed_step": cyc, "simds": 1024, "clock_GHz": 2.1,
                                            "issue_bound_ms": issue_ms, "frac_of_issue_bound": issue_ms / c_ms,
                                            "note": "instruction issue of the replay loop alone (ISA count x the gfx950 prices measured by "
                                                    "tools/probe/valu_cost_probe.hip); exact: 38 packed + 8 transcendental + 3; bounded: "
                                                    "14 packed + 2 single + 1.5 per 4 elements and step (no transcendental: the reciprocal is carried from step to step; "
                                                    "the launch does not follow this bound — the kernel waits for rows, profiles/r05_catchup_reciprocal.md)"},
                            "traffic": tr("sparse_catchup_k") if mode_catchup == "exact" else tr("sparse_catchup_bounded_k"),
                            "traffic_note": "PMC bytes per launch of the row kernel (profiles/traffic.json)"}
        out = {
            "metric": "examples/sec DeepFM batch=65536 (full train step)",
            "value": world * B * args.steps / dt,
            "unit": "examples/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            # fp32 variables, activations, gradients, accumulation and results; the MLP's GEMM operands go to the matrix pipe
            # as fp16 high + low parts (3 exact products per fp32 product, lo*lo dropped: ~22-bit operands, row-relative error
            # < 1e-5) unless --gemm fp32.  The un-emulated number is config.gemm_fp32_examples_per_sec.
            "dtype": {"f16x2": "f32 (MLP GEMM operands: fp16 hi+lo split, 3 products, fp32 accumulate)",
                      "bf16x3": "f32 (MLP GEMM operands: bf16 x3 split, 6 products, fp32 accumulate)", "fp32": "f32"}[args.gemm],
            "data": "synthetic (%s ids, a fresh batch every step from a pool of up to %d, random-init weights)" % (args.dist, POOL),
            "config": {"workload": "config 3: trainers.deep_fm --embedding-size 64 --hidden-units 512 256 128 "
                                   "--batch-size 65536 --dropout 0.1, 26 fields x 1M ids (Criteo-shaped), Adam(1e-3)",
                       "per_gpu_batch": B, "global_batch": world * B, "fields": F, "vocab_per_field": V,
                       "embedding_size": E, "hidden_units": HIDDEN,
                       "gemm": {"f16x2": "fp32 GEMMs via scaled fp16 high+low operand split, fp32 accumulate",
                                "bf16x3": "fp32 GEMMs via 3-way bf16 operand split, fp32 accumulate",
                                "fp32": "fp32-input MFMA"}[args.gemm],
                       "parallelism": "dp%d + row-sharded embeddings (all-to-all)" % world if (world > 1 or args.force_shard) else "single GPU",
                       "communicators": (None if shard is None else ("two: the next batch is routed ahead on a second communicator" if shard.route_ahead
                                                                     else "one: every collective of a step in program order on one communicator")),
                       "catchup": ("bounded-error lazy Adam replay (MI_CATCHUP_BOUNDED: every variable within 3 ulp + 2e-6 of the replayed "
                                   "movement of TF's sweep, 98.7 % within 1e-7 relative; the exact mode is the extra catchup_exact)"
                                   if mode_catchup == "bounded" else "lazy Adam replay bit-exact with TF's dense-equivalent sweep"),
                       "input_pipeline": (("next batch's ids announced one step ahead (train_step(next_ids=...)): " +
                                           ("their routing (sorts, count exchange, split sizes to the host) runs on a side stream "
                                            "beside this step's sparse apply" if shard is not None else
                                            "their sort runs on a side stream beside this step's catch-up"))
                                          if presort[0] else "ids handed over step by step")},
            "roofline": {"kernel": ("embed_fm_planes_fwd_k: embedding gather + FM second order, writes the input_layer concat as fp16 "
                                    "high/low planes with one exponent per example (the operand of the layer-1 GEMMs); the wide part's "
                                    "4-byte gathers run as linear_only_fwd_k on a side stream" if planes_gather else
                                    "embed_fm_linear_fwd_k (embedding gather + FM second order; read-only form)"), "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of_measured_copy_peak": achieved / HBM_MEASURED_GBS,
                         "algorithmic_bytes_per_launch": int(gather_bytes), "examples_per_launch": int(Bl), "avg_launch_ms": g_ms,
                         "row_read_GBs": row_bytes / (g_ms * 1e-3) / 1e9, "row_read_frac": row_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms_without_side_stream": alone_ms,
                         "frac_without_side_stream": (gather_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if alone_ms else None,
                         "side_stream_note": ("in the timed region the next batch's sort (side stream, started beside the catch-up) ends "
                                              "under this kernel and shares HBM with it; *_without_side_stream: the same launches over a "
                                              "few steps whose batches were handed over one by one — the kernel by itself") if alone_ms else None,
                         "traffic": traffic,
                         "traffic_note": "HBM bytes/launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) of the same kernel on this workload, "
                                         "committed in profiles/traffic.json — a constant read from that file, not a counter of this run",
                         "launch_time_note": "avg_launch_ms: HIP events around the launches of the timed region; a long rocprofv3 trace of "
                                             "the same build averages ~4 % less (boxes of the pool and cold starts differ by that much)"},
            "roofline_mlp": dict(mlp_roofline(args.gemm, flops, gemm_ms), overhead_launches_ms_per_step=gemm_overhead_ms),
            "roofline_sparse_apply": roof_apply,
            "roofline_catchup": roof_catchup,
            "configs": configs,
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
        if other_scaling is not None:
            out["other_scaling"] = other_scaling
        if args.cpu_baseline_json:
            with open(args.cpu_baseline_json) as f:
                out["cpu_baseline"] = json.load(f)
        elif not args.no_cpu_baseline:
            # (N > 1 under a launcher: every collective of the run is done — the other ranks wait in the final barrier)
            log("cpu baseline")
            out["cpu_baseline"] = cpu_baseline()
            if world > 1:
                out["cpu_baseline"]["sample"] += "; timed on rank 0's host threads after the GPU legs, the process group gone and the other %d ranks leaving" % (world - 1)
        # Numbers a record that truncates long strings and keeps only the line's tail must still show: flat numeric
        # copies in `config` (the strict legs beside the headline's two disclosed modes; the other rooflines; the other
        # BASELINE configs), and the same as a compact `summary` object at the very END of the line.
        num = {"ms_per_step": ms_step, "gather_frac": achieved / HBM_PEAK_GBS,
               "gather_frac_without_side_stream": out["roofline"]["frac_without_side_stream"],
               "gemm_ms_per_step": gemm_ms, "mlp_frac": out["roofline_mlp"]["frac"],
               "mlp_frac_of_sustained": out["roofline_mlp"]["frac_of_sustained"],
               "sparse_apply_frac": roof_apply["frac"] if roof_apply else None,
               "sparse_apply_ms": roof_apply["avg_launch_ms"] if roof_apply else None,
               "catchup_frac": roof_catchup["frac"] if roof_catchup else None,
               "catchup_ms": roof_catchup["avg_launch_ms"] if roof_catchup else None,
               "cold_start_examples_per_sec": cold["value"]}
        for k_, v_ in extras.items():
            num[k_ + "_examples_per_sec"] = v_["value"]
            num[k_ + "_ms_per_step"] = v_["ms_per_step"]
        if other is not None:
            num["other_distribution_examples_per_sec"] = other["value"]
        if other_scaling is not None:
            num[other_scaling["scaling"] + "_scaling_examples_per_sec"] = other_scaling["value"]
            num[other_scaling["scaling"] + "_scaling_ms_per_step"] = other_scaling["ms_per_step"]
        if configs:
            num.update(c2_us_eager=configs["c2"]["us_per_step_eager"], c2_us_hip_graph=configs["c2"]["us_per_step_hip_graph"],
                       c4_defaults_ms=configs["c4_defaults"]["ms_per_step"], c4_c3sizes_ms=configs["c4_c3sizes"]["ms_per_step"],
                       c5_rank_share_ms=configs["c5_rank_share"]["ms_per_step"])
        num = {k_: (round(v_, 4) if isinstance(v_, float) and abs(v_) < 1e5 else (int(v_) if isinstance(v_, float) else v_))
               for k_, v_ in num.items() if v_ is not None}
        out["config"].update(num)
        top = {k_: v_ for k_, v_ in km.items()}
        out["summary"] = dict(num, value=int(out["value"]),
                              kernel_ms_per_step={k_: round(v_[2] / args.steps, 4) for k_, v_ in sorted(top.items(), key=lambda kv: -kv[1][2])[:14]})
        print(json.dumps(out))


if __name__ == "__main__":
    main()
