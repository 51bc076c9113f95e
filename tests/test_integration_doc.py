"""INTEGRATION.md's Level-2 stub — the binding a maintainer of the reference would paste into trainers/ — is EXECUTED here:
the code block is cut out of the document and run against the built library (the stub had gone stale
against the header once, and nothing noticed)."""
import os
import re

import numpy as np
import pytest

from mi355x_rec import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    stub = [b for b in blocks if b.startswith("# trainers/_mi355x.py")]
    assert len(stub) == 1, "INTEGRATION.md must hold exactly one '# trainers/_mi355x.py' code block"
    return stub[0]


def _header_arg_count(name):
    src = open(os.path.join(ROOT, "include", "mi355x_rec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"\b%s\s*\(([^;{}]*?)\)\s*;" % name, src, flags=re.S)
    return m.group(1).count(",") + 1


def _load_stub():
    os.environ["MI355X_REC_LIB"] = _lib.LIB_PATH
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md:trainers/_mi355x.py", "exec"), ns)
    return ns


def test_stub_binds_the_header_as_it_is_now(lib):
    """CPU: the block runs (the library loads, the ABI version it asserts is the library's) and the argument list it
    declares has the header's length and the shipped binding's types."""
    ns = _load_stub()
    fn = ns["lib"].mi_embed_fm_linear_fwd
    assert len(fn.argtypes) == _header_arg_count("mi_embed_fm_linear_fwd") == len(_lib.SIGNATURES["mi_embed_fm_linear_fwd"][1])
    assert list(fn.argtypes) == list(_lib.SIGNATURES["mi_embed_fm_linear_fwd"][1])
    assert "== %d" % _lib.ABI_VERSION in _stub_source()


@pytest.mark.gpu
def test_stub_computes_what_the_reference_lines_compute():
    """GPU: the stub's embed_fm_linear on plain [R, E] / [R] arrays (what a reference-side caller holds) against
    oracle.deepfm.forward (deep_fm.py:39-90): sumv and the wide sum to fp32 rounding, the FM term to 1e-5 of its scale."""
    import torch
    from oracle import deepfm as O
    ns = _load_stub()
    rng = np.random.default_rng(5)
    vocab, E, B = [13, 7, 29, 5], 16, 67
    p = O.init_params(rng, vocab, E, [8], dtype=np.float32, lin_scale=0.1)
    ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
    c = O.forward(p, ids, use_dnn=False)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    table, lin_w = dev(np.concatenate(p.emb, 0)), dev(np.concatenate(p.lin_w, 0))
    off = dev(np.concatenate([[0], np.cumsum(vocab)[:-1]]).astype(np.int64))
    d_ids = dev(ids)
    sumv, fm, lin = (torch.empty(B, E, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda"))
    ns["embed_fm_linear"](table.data_ptr(), lin_w.data_ptr(), off.data_ptr(), d_ids.data_ptr(), B, len(vocab), E,
                          sumv.data_ptr(), fm.data_ptr(), lin.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.allclose(sumv.cpu().numpy(), c["sumv"], rtol=0, atol=2e-6)
    assert np.allclose(lin.cpu().numpy() + p.lin_bias[0], c["lin"], rtol=0, atol=1e-6)
    scale = np.sqrt(np.mean(c["fm"] ** 2))
    assert np.max(np.abs(fm.cpu().numpy() - c["fm"])) < 1e-5 * scale
    # error behaviour: a non-zero status and a message, never an exception across the boundary
    with pytest.raises(RuntimeError):
        ns["embed_fm_linear"](table.data_ptr(), lin_w.data_ptr(), off.data_ptr(), d_ids.data_ptr(), B, len(vocab), 6,
                              sumv.data_ptr(), fm.data_ptr(), lin.data_ptr(), torch.cuda.current_stream().cuda_stream)
