"""CPU oracle for the DeepFM / Wide&Deep training hot path  — TEST INFRASTRUCTURE ONLY.

This package restates, op for op, the arithmetic that leotimus/recommender-tensorflow's
``trainers/deep_fm.py:36-125`` asks TensorFlow 1.12 to perform (and the canned estimators behind
``trainers/{linear,deep,linear_deep}.py``).  It exists so that ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` have something to check
the HIP path against.  Nothing under ``recommender-tensorflow_amd/`` may import it: the product
path is the HIP library and fails loudly without it.

PARITY UNPINNED.  The algorithm lives in the un-vendored third-party dependency
``tensorflow=1.12`` (reference ``environment.yml:10``), which is absent from the reference tree,
not installed here, not installable offline, and has no Python 3.10 build.  The reference holds
no tests, golden vectors or fixtures for this path, so the oracle cannot be
pinned on reference-held data.  What pins it instead:

* FarmHash Fingerprint64 known answers recalled from TensorFlow's own
  ``string_to_hash_bucket`` tests / docs (``tests/test_fingerprint.py``), cross-checked between
  two independent restatements (this package's pure Python and the C in ``csrc/host_ids.cpp``);
* independent implementations of the same mathematics (``tests/test_oracle_vs_torch.py``): the graph written again in
  plain torch ops and differentiated by autograd (every gradient to 1e-10), ``torch.optim`` Adam / Adagrad / SGD beside
  the dense rules, scikit-learn's metrics beside the streaming ones — the calculus and the update algebra, not TF's bits;
* mathematical identities the reference's formulas must satisfy (FM == sum of pairwise dot
  products, stable sigmoid-CE == naive form, finite-difference gradients in fp64, Adam step-1
  closed form, sync-DP == big batch);
* self-generated golden vectors under ``tests/golden`` (``tests/golden/make_golden.py``) — they
  detect drift of this package, they are NOT reference outputs.

Every function cites the reference file:line it follows; TensorFlow-internal semantics are cited
as ``SURVEY A.n`` (recalled from TF 1.12 sources).
"""

