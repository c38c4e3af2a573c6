"""CPU oracle for the ST-GCN block stack of UttaranB127/STEP.

TEST INFRASTRUCTURE ONLY.  Nothing under ``step_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline, never as the shipped compute path.

The functions here restate, from scratch and in plain PyTorch/numpy on the CPU,
the algorithm of the reference's hot path (each function cites the reference
file:line it follows).  Parity is PINNED: ``tests/golden/make_golden.py``
imported the real reference modules from ``/root/reference`` in the build
container and wrote input/output vectors (``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks this oracle against every one of them.
"""
from .graph import graph_adjacency, hop_distance, normalize_digraph  # noqa: F401
from . import stgcn  # noqa: F401
