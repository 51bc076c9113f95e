"""CPU: the host-side logic of bench.py that no GPU run on this pool exercises — the record a parent-mode `python bench.py --gpus N`
assembles from its two sets of ranks."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _line(value, ms, **extra):
    return json.dumps(dict({"metric": "examples/sec DeepFM batch=65536 (full train step)", "value": value, "ms_per_step": ms, "n_gpus": 8,
                            "config": {"workload": "config 3"}}, **extra))


def test_parent_mode_prints_the_faster_set_with_both_modes_in_config():
    b = _bench()
    one = _line(100.0, 4.2, cpu_baseline={"value": 11000.0}, other_distribution={"value": 90.0})
    two = _line(110.0, 3.9)
    best = b.merge_mode_lines(one, two, 0)
    assert best["value"] == 110.0 and best["config"]["communicator_mode_of_value"] == "route_ahead_second_communicator"
    assert best["config"]["communicator_modes"] == {"one_communicator": {"value": 100.0, "ms_per_step": 4.2},
                                                    "route_ahead_second_communicator": {"value": 110.0, "ms_per_step": 3.9}}
    assert best["cpu_baseline"] == {"value": 11000.0} and best["other_distribution"] == {"value": 90.0}      # carried over from set 1
    best = b.merge_mode_lines(_line(120.0, 3.5, cpu_baseline={"value": 1.0}), two, 0)
    assert best["value"] == 120.0 and best["config"]["communicator_mode_of_value"] == "one_communicator"


def test_parent_mode_survives_a_dead_second_set():
    b = _bench()
    best = b.merge_mode_lines(_line(100.0, 4.2, cpu_baseline={"value": 11000.0}), None, -6)
    assert best["value"] == 100.0 and best["config"]["communicator_mode_of_value"] == "one_communicator"
    assert "failed" in best["config"]["communicator_modes"]["route_ahead_second_communicator"]
    assert "-6" in best["config"]["communicator_modes"]["route_ahead_second_communicator"]["failed"]


def test_default_flags_match_the_driver_contract():
    """`python bench.py` with no flags: N = 1 and a K / W that finish within minutes; the catch-up mode of the headline is the
    library's and the CLIs' default (one headline, one default)."""
    import sys
    b = _bench()
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = b.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup) == (1, 100, 20) and a.catchup == "bounded" and a.route_ahead is None
    sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
    import inspect
    from mi355x_rec.engine import DeepFM
    from trainers import _cli
    assert inspect.signature(DeepFM.__init__).parameters["catchup"].default == a.catchup
    opt = ("exclude_linear", "exclude_mf", "exclude_dnn", "hidden_units", "dropout")
    assert _cli.make_parser("deep_fm", opt).parse_args([]).catchup == a.catchup


def test_every_tool_parses():
    """tools/*.py run on the GPU box only: at least they must be Python"""
    import ast
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        ast.parse(open(p).read(), p)


def test_lookahead_loop_yields_the_same_batches_in_the_same_order():
    """Estimator._with_lookahead: large batches are read one ahead — every batch still comes out once, in order, and
    params['_lookahead'] names the batch that follows (None behind the last); small batches and row-sharded runs pass through."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
    import numpy as np
    from mi355x_rec.estimator import Estimator

    class E(Estimator):
        def __init__(self):
            self.params = {}
    for B, expect_lookahead in ((4096, True), (32, False)):
        e = E()
        batches = [({"user_id": np.full(B, i)}, np.full(B, i % 2)) for i in range(5)]
        seen, ahead = [], []
        for f, l in e._with_lookahead(iter(batches)):
            seen.append(int(f["user_id"][0]))
            la = e.params.get("_lookahead")
            ahead.append(None if la is None else int(la["features"]["user_id"][0]))
        assert seen == [0, 1, 2, 3, 4]
        assert ahead == ([1, 2, 3, 4, None] if expect_lookahead else [None] * 5)
        assert "_lookahead" not in e.params
    e = E()
    e.params["_shard"] = object()                        # (a row-sharded run: Estimator._shard)
    assert [int(f["user_id"][0]) for f, _ in e._with_lookahead(iter(batches))] == [0, 1, 2, 3, 4] and "_lookahead" not in e.params
    assert list(E()._with_lookahead(iter([]))) == []
