"""-m gpu: the multi-rank step with the REAL HIP kernels.  The GPU box has one MI355X, so
(a) two processes share cuda:0 and exchange through gloo (host-staged collectives): kernels +
    routing + sharded optimizer state are the shipped ones, only the transport differs;
(b) a single rank runs the RCCL ("nccl") code path end to end (all_to_all_single / all_reduce on
    device tensors with world_size 1).
The 8-GPU RCCL run itself is the driver's (bench.py --gpus N)."""
import pytest

from tests.test_distributed_cpu import CASES, _run, check_against_big_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [CASES[0], CASES[1], ([50, 30, 20, 40], 64, [64, 32], 128, 0, "Adam", 0.001, 3, (True, True, True)),
                                 CASES[4],                                                                        # pipelined, 4 chunks
                                 ([50, 30, 20, 40], 64, [64, 32], 256, 0, "Adam", 0.001, 3, (True, True, True), 2),   # 2 chunks of 128
                                 # BASELINE config 5's model shape: 40 fields, E=128, [512,256,128], row-sharded, pipelined
                                 ([13] * 40, 128, [512, 256, 128], 128, 0, "Adam", 0.001, 2, (True, True, True), 2),
                                 # BASELINE config 4's model: Wide&Deep + raw numeric columns, Ftrl + Adagrad, SUM loss
                                 CASES[7],
                                 # chunked exchanges, one forward / backward (the default from 8 ranks on), config-5 shape
                                 ([13] * 40, 128, [512, 256, 128], 128, 0, "Adam", 0.001, 2, (True, True, True), 2, dict(chunk_compute=False)),
                                 ([50, 30, 20, 40], 64, [64, 32], 256, 0, "Adam", 0.001, 3, (True, True, True), 2,
                                  dict(chunk_compute=False, announce=True)),
                                 # config 4's model with the numeric columns written into the planes by the gather (E = 32),
                                 # the embedding-side kernels in two pieces
                                 ([50, 30, 20, 40], 32, [64, 32], 128, 3, "Adagrad", 0.05, 2, (True, False, True), 2,
                                  dict(numeric="raw", lin_opt=("Ftrl", 0.2), reduction="sum", chunk_compute=False)),
                                 # the whole step on ONE communicator (RowShard(route_ahead=False)): announced batches are ignored
                                 ([50, 30, 20, 40], 64, [64, 32], 256, 0, "Adam", 0.001, 3, (True, True, True), 2,
                                  dict(chunk_compute=False, announce=True, route_ahead=False)),
                                 # the packed exchange: rows + wide weights (and their gradients) as records of E + 4 floats
                                 ([50, 30, 20, 40], 64, [64, 32], 256, 0, "Adam", 0.001, 3, (True, True, True), 2,
                                  dict(chunk_compute=False, announce=True, packed=True)),
                                 ([50, 30, 20, 40], 32, [64, 32], 128, 3, "Adagrad", 0.05, 2, (True, False, True), 2,
                                  dict(numeric="raw", lin_opt=("Ftrl", 0.2), reduction="sum", packed=True)),
                                 # wide and deep parts on different columns, per-column embedding dimensions, numeric columns
                                 # that one part reads (linear_deep.py:32-39 with two different lists) on two ranks
                                 ([50, 30, 20, 40, 7, 11], 32, [64, 32], 128, 3, "Adagrad", 0.05, 3, (True, False, True), 2,
                                  dict(numeric="raw", lin_opt=("Ftrl", 0.2), reduction="sum", field_dims=[32, 16, 32, 0, 8, 32],
                                       wide_fields=[True, False, True, True, False, True], deep_numeric=[True, False, True],
                                       wide_numeric=[True, True, False])),
                                 # 4,096 examples per rank and a 128-unit last hidden layer: it runs inside the fused logits + head
                                 # launch (engine._top_fusable), one forward / backward (chunk_compute=False) and a chunk of 4,096 each
                                 # (SUM loss: with the mean over 8,192+ examples many gradient elements lie below Adam's epsilon, where
                                 # an update amplifies a 1e-10 difference of the gradient to 1e-5 of the weight)
                                 ([50, 30, 20, 40], 32, [128, 128], 4096, 0, "Adam", 0.001, 2, (True, True, True), 2,
                                  dict(chunk_compute=False, reduction="sum")),
                                 ([50, 30, 20, 40], 32, [128, 128], 8192, 0, "Adam", 0.001, 2, (True, True, True), 2,
                                  dict(chunk_compute=True, reduction="sum"))])
def test_two_ranks_one_gpu_gloo(cfg):
    check_against_big_batch(cfg, _run(cfg, 2, device="cuda", backend="gloo"), 2, tol=3.0)


@pytest.mark.parametrize("chunks,chunk_compute,route_ahead", [(1, True, True), (3, True, True), (3, False, True), (3, False, False)])
def test_single_rank_rccl_path(chunks, chunk_compute, route_ahead):
    """chunks = 3: the asynchronous all_to_all handles of the pipelined step on RCCL's own stream; both route modes (a
    second RCCL communicator for the batch routed ahead / one communicator for the whole step)"""
    cfg = ([50, 30, 20, 40], 16, [32, 16], 96, 0, "Adam", 0.001, 3, (True, True, True), chunks,
           dict(chunk_compute=chunk_compute, announce=True, route_ahead=route_ahead))
    check_against_big_batch(cfg, _run(cfg, 1, device="cuda", backend="nccl"), 1, tol=3.0)


@pytest.mark.parametrize("chunks,chunk_compute", [(1, True), (2, True), (2, False)])
def test_modelled_link_time_changes_no_number(chunks, chunk_compute):
    """RowShard(sim_links=...) — tools/sim_ranks.py's rehearsal of an N-rank job on one GPU: a spin kernel on a stream of its
    own behind every exchange (row and gradient all-to-alls, id and count exchanges, the dense all-reduce).  It may only cost
    time: the step's results equal the oracle's big-batch step like the plain one-rank RCCL path's."""
    cfg = ([50, 30, 20, 40], 16, [32, 16], 96, 0, "Adam", 0.001, 3, (True, True, True), chunks,
           dict(chunk_compute=chunk_compute, announce=True, route_ahead=False,
                sim_links={"world": 8, "gbs": 7 * 45.0, "latency_us": 40.0}))
    check_against_big_batch(cfg, _run(cfg, 1, device="cuda", backend="nccl"), 1, tol=3.0)
