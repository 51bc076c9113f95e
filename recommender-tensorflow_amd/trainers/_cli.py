"""Shared argument table and run loop of the four trainer CLIs (flags and defaults of the
reference: trainers/deep_fm.py:182-207, linear.py:50-65, deep.py:54-73, linear_deep.py:55-74)."""
import os
import shutil
from argparse import ArgumentParser

from mi355x_rec.estimator import ModeKeys, train_and_evaluate as _tae
from trainers.conf_utils import get_eval_spec, get_exporter, get_run_config, get_train_spec
from trainers.ml_100k import get_feature_columns, get_input_fn, serving_input_fn

# flag -> (kwargs); every trainer takes the common ones, the model-specific ones are opted in by name
_COMMON = [
    ("--train-csv", dict(default="data/ml-100k/train.csv", help="path to the training csv data (default: %(default)s)")),
    ("--test-csv", dict(default="data/ml-100k/test.csv", help="path to the test csv data (default: %(default)s)")),
    ("--restore", dict(action="store_true", help="whether to restore from job_dir")),
    ("--embedding-size", dict(type=int, default=4, help="embedding size (default: %(default)s)")),
    ("--batch-size", dict(type=int, default=32, help="batch size (default: %(default)s)")),
    ("--train-steps", dict(type=int, default=20000, help="number of training steps (default: %(default)s)")),
    ("--device", dict(default="cuda", help="torch device of the MI355X to run on (default: %(default)s)")),
    ("--warm-start-from", dict(default=None, help="a TensorFlow checkpoint of the reference — its model_dir, a model.ckpt-N prefix, or "
                                                   "an .npz dump of its variables by name — to start from when job_dir has no checkpoint")),
    ("--world-size", dict(type=int, default=None, help="number of MI355X (one process each): launch with `python -m "
                                                       "torch.distributed.run --nproc-per-node N -m trainers.<model> ...`; the flag "
                                                       "only checks the launch (default: WORLD_SIZE of the launcher, else 1)")),
    ("--single-communicator", dict(action="store_true", help="N > 1 GPUs: run every collective of a step on ONE RCCL communicator (no "
                                                             "routing of the next batch ahead on a second one)")),
    ("--collective-timeout", dict(type=float, default=600.0, help="N > 1 GPUs: seconds after which a stuck collective aborts the process")),
    ("--hip-graph", dict(nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                         help="replay the train step as one hipGraph launch (bit for bit the eager step): auto = for batches of at most "
                              "1,024 examples on one GPU, where a step is launch-bound (default: %(default)s)")),
    ("--catchup", dict(choices=["exact", "bounded"], default="bounded",
                       help="Adam only: how the steps a table row sat out are replayed when it is next read — bounded: every variable "
                            "within 3 ulp + 2e-6 of the movement the replay covers (98.7 %% of them within 1e-7 relative of TensorFlow's "
                            "sweep; logits / loss stay inside 1e-5), a third of the instructions; exact: TensorFlow's fp32 sequence bit "
                            "for bit, ~8 %% slower at large vocabularies (default: %(default)s — the mode bench.py's headline is timed in)")),
    ("--synthetic", dict(type=int, default=None, metavar="N", help="train on N generated MovieLens-shaped examples (and evaluate on "
                                                                   "N/10) instead of --train-csv / --test-csv")),
]
_OPTIONAL = {
    "hidden_units": ("--hidden-units", dict(type=int, nargs="+", default=[16, 16],
                                            help="hidden layer specification (default: %(default)s)")),
    "dropout": ("--dropout", dict(type=float, default=0.1, help="dropout rate (default: %(default)s)")),
    "exclude_linear": ("--exclude-linear", dict(action="store_true", help="flag to exclude linear component (default: %(default)s)")),
    "exclude_mf": ("--exclude-mf", dict(action="store_true", help="flag to exclude mf component (default: %(default)s)")),
    "exclude_dnn": ("--exclude-dnn", dict(action="store_true", help="flag to exclude dnn component (default: %(default)s)")),
}


def make_parser(model, extra=()):
    p = ArgumentParser()
    p.add_argument("--job-dir", default="checkpoints/" + model, help="job directory (default: %(default)s)")
    for flag, kw in _COMMON:
        p.add_argument(flag, **kw)
    for name in extra:
        flag, kw = _OPTIONAL[name]
        p.add_argument(flag, **kw)
    return p


def init_distributed(args):
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE; the reference's multi-worker
    mode is TF_CONFIG parameter servers, distributed.md:58-82).  Returns (rank, world, device, shard).  Runs before any
    GPU work: the process group and the device choice come first, nothing is re-executed afterwards."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if getattr(args, "world_size", None) not in (None, world):
        raise SystemExit("--world-size %d but the launcher started %d process(es): use python -m torch.distributed.run "
                         "--nproc-per-node %d -m trainers.<model> ..." % (args.world_size, world, args.world_size))
    device = getattr(args, "device", "cuda")
    if world == 1:
        return 0, 1, device, None
    import torch
    import torch.distributed as dist
    from mi355x_rec.parallel import RowShard
    import datetime
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a stuck collective ends the process non-zero (the watchdog names the rank and the collective) instead of hanging the node
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    tmo = datetime.timedelta(seconds=float(getattr(args, "collective_timeout", None) or 600.0))
    if device.startswith("cuda"):
        device = "cuda:%d" % local
        torch.cuda.set_device(local)
        from mi355x_rec.parallel import rccl_options
        dist.init_process_group("nccl", device_id=torch.device(device), timeout=tmo, **rccl_options(tmo))
    else:
        dist.init_process_group("gloo", timeout=tmo)
    return rank, world, device, RowShard(rank, world, route_ahead=not getattr(args, "single_communicator", False))


def run(args, make_estimator):
    """Common body of train_and_evaluate(args): wipe job_dir unless --restore, build the estimator
    from the MovieLens feature columns, train with periodic eval + export.  With N processes (one per GPU) the
    embedding tables are row-sharded, every rank trains on its own shuffled stream of --batch-size examples
    (global batch N x batch size, synchronous), all ranks evaluate the same test batches, and rank r keeps its
    shard in model.ckpt-<step>.rank<r>.pt."""
    rank, world, device, shard = init_distributed(args)
    if getattr(args, "synthetic", None):
        args.train_csv, args.test_csv = "synthetic:%d:1" % args.synthetic, "synthetic:%d:2" % max(args.synthetic // 10, 1)
    if not args.restore and rank == 0:
        shutil.rmtree(args.job_dir, ignore_errors=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    columns = get_feature_columns(embedding_size=args.embedding_size)
    config = get_run_config()
    config.device = device
    estimator = make_estimator(columns, config)
    estimator.warm_start_from = getattr(args, "warm_start_from", None)
    estimator.params["_shard"] = shard
    estimator.params["hip_graph"] = getattr(args, "hip_graph", "auto")
    estimator.params["catchup"] = getattr(args, "catchup", "bounded")
    train_spec = get_train_spec(get_input_fn(args.train_csv, batch_size=args.batch_size, seed=rank if world > 1 else None),
                                args.train_steps)
    eval_spec = get_eval_spec(get_input_fn(args.test_csv, ModeKeys.EVAL, batch_size=args.batch_size),
                              get_exporter(serving_input_fn))
    return _tae(estimator, train_spec, eval_spec)
