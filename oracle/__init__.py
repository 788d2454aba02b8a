"""CPU oracle for the RAG-Gesture inference hot path.  TEST INFRASTRUCTURE ONLY.

A plain torch-fp32 (CPU) restatement of the reference's algorithm for the path named
by BASELINE.json (VAE encode -> [retrieval -> DDIM inversion] -> 50-step DDIM with CFG
[+ insertion guidance] -> VAE decode); every function cites the reference file:line it
follows.  It recomputes everything per step exactly like the reference does (no
hoisting, no algebraic shortcuts), so it doubles as the "port" CPU baseline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker.  The product (rag-gesture_amd/) never imports it.

Parity pinning: the reference has no tests or golden vectors of its own;
this oracle is pinned against outputs of the reference itself, imported in the build
container by tests/golden/make_goldens.py (fixtures in tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  Unpinned pieces are listed in DESIGN.md.
"""
