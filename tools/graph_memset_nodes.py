#!/usr/bin/env python3
"""Evidence for the captured-memset fault of round 2 (commit 8038430), WITHOUT re-triggering it: with
MI_SORT_MEMSET=1 the sorts zero their counters with hipMemsetAsync again; the B = 1024 MovieLens-shaped step (a linear
graph: no side stream below 4096 examples) is captured and NOT replayed; hipGraphDebugDotPrint lists every node, and for
each memset node its destination and extent are compared with the live torch allocations (the engine's workspaces were
allocated by the eager sizing step, before the capture opened its private pool).
usage: MI_SORT_MEMSET=1 python tools/graph_memset_nodes.py"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
os.environ.setdefault("MI_SORT_MEMSET", "1")
import numpy as np
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec

VOCAB = [2, 2, 7, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 2000, 2, 2, 50, 8, 2, 2, 2, 2, 1000, 2, 2, 1000]
B = 1024
m = DeepFM(VOCAB, embedding_size=4, hidden_units=[16, 16], dropout=0.25, seed=5, optimizer=OptimizerSpec("Adam", 0.001))
g = torch.Generator(device="cuda"); g.manual_seed(0)
m.init_variables(g, lin_scale=1e-3)
ids = torch.stack([torch.randint(0, v, (B,), device="cuda", generator=g) for v in VOCAB], 1).to(torch.int32).contiguous()
y = (torch.rand(B, device="cuda", generator=g) < 0.3).to(torch.uint8)
for _ in range(2):
    m.train_step(ids, y)                      # eager: sizes every workspace
torch.cuda.synchronize()
before = {k: (t.data_ptr(), t.numel() * t.element_size()) for k, t in m._ws.items() if isinstance(t, torch.Tensor)}
state = torch.zeros(16, dtype=torch.uint8, device="cuda")
m._write_step_state(state)
graph = torch.cuda.CUDAGraph(keep_graph=True)        # the captured hipGraph_t stays inspectable; nothing is instantiated
m.k.query("mi_set_step_state", state.data_ptr())
m._capturing = True
try:
    with torch.cuda.graph(graph):
        m.k.mi_step_advance(state, m.sched.table)
        m.train_step(ids.clone(), y.clone())
finally:
    m._capturing = False
    m.k.query("mi_set_step_state", None)
after = {k: (t.data_ptr(), t.numel() * t.element_size()) for k, t in m._ws.items() if isinstance(t, torch.Tensor)}
moved = [k for k in before if before[k] != after.get(k)]
new = [k for k in after if k not in before]
print("workspaces that moved between the eager sizing step and the capture:", moved or "none", "| created during the capture:", new or "none")
import ctypes as C
hip = C.CDLL("libamdhip64.so")                  # (the copy torch already mapped)
h = C.c_void_p(graph.raw_cuda_graph())
n = C.c_size_t(0)
assert hip.hipGraphGetNodes(h, None, C.byref(n)) == 0
arr = (C.c_void_p * n.value)()
assert hip.hipGraphGetNodes(h, arr, C.byref(n)) == 0
ne = C.c_size_t(0)
hip.hipGraphGetEdges(h, None, None, C.byref(ne))


class MemsetParams(C.Structure):                # hipMemsetParams
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint), ("height", C.c_size_t), ("pitch", C.c_size_t),
                ("value", C.c_uint), ("width", C.c_size_t)]


NAMES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}
kinds, memsets = {}, []
for node in arr:
    t = C.c_int(-1)
    assert hip.hipGraphNodeGetType(C.c_void_p(node), C.byref(t)) == 0
    kinds[NAMES.get(t.value, t.value)] = kinds.get(NAMES.get(t.value, t.value), 0) + 1
    if t.value == 2:
        mp = MemsetParams()
        assert hip.hipGraphMemsetNodeGetParams(C.c_void_p(node), C.byref(mp)) == 0
        memsets.append(mp)
print("captured graph: %d nodes %s, %d edges (a linear graph has nodes - 1)" % (n.value, kinds, ne.value))
allocs = sorted((p, p + nb, k) for k, (p, nb) in after.items())
allocs += [(m.dense.data_ptr(), m.dense.data_ptr() + m.dense.numel() * 4, "dense"), (m._amax.data_ptr(), m._amax.data_ptr() + m._amax.numel() * 4, "_amax")]
for mp in memsets:
    nbytes = mp.width * mp.elementSize * max(mp.height, 1)
    hit = [(k, mp.dst - a, b - a) for a, b, k in allocs if a <= mp.dst < b]
    where = ("workspace %r at offset %d of %d bytes, extent ends %d bytes before the allocation's end" %
             (hit[0][0], hit[0][1], hit[0][2], hit[0][2] - hit[0][1] - nbytes)) if hit else "NOT inside any engine workspace"
    print("memset node: dst 0x%x value %d elementSize %d width %d height %d pitch %d -> %d bytes; %s" %
          (mp.dst, mp.value, mp.elementSize, mp.width, mp.height, mp.pitch, nbytes, where))
print("(the graph was NOT instantiated and NOT replayed)")
