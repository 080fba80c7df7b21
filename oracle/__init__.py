"""CPU oracle for the R-GCN-VAE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain torch-CPU / numpy restatement of the arithmetic that
karenyang/GCN-VAE executes on its link-prediction hot path.  It exists to *check*
the HIP implementation in ``gcn-vae_amd/``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; nothing under ``gcn-vae_amd/`` does (a test enforces that).

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.flows``, ``oracle.prob``, ``oracle.kgvae`` (KGVAE wiring, KL, MMD,
  DistMult scorer, loss), ``oracle.graphs`` (graph build, sampling, negative
  sampling) and ``oracle.ranking`` are PINNED: they are checked against golden
  vectors produced by importing the reference's own Python
  (``/root/reference/kgvae/{flow_network,utils,model,link_predict}.py``) in the build
  container -- generator script ``tests/golden/make_golden.py``, vectors committed
  under ``tests/golden/``.
* ``oracle.rgcn`` (the ``dgl.nn.pytorch.RelGraphConv`` layer) is **PARITY
  UNPINNED**: DGL is a third-party, un-vendored, un-pinned dependency of the
  reference (API evidence dates it to DGL 0.4.x) that is neither in
  ``/root/reference`` nor installable here, and the reference holds no test or
  golden vector at that boundary.  The restatement follows DGL 0.4.x's published
  ``RelGraphConv`` (bdd / basis message functions, sum reduce, bias, self loop,
  activation, dropout) and is cross-checked three independent ways in
  ``tests/test_oracle_rgcn.py`` (dense per-relation adjacency formulation, scalar
  triple loop, fp64 gradcheck).

Everything here is written functionally: functions take plain tensors plus a
``state`` dict that uses the reference's ``state_dict`` key names, so the same
weights can be fed to the reference (fixture generation), to this oracle, and to the
HIP modules.
"""

__all__ = ["rgcn", "flows", "prob", "kgvae", "graphs", "ranking"]
