"""CPU oracle for the YOLOv7 hot path (TEST INFRASTRUCTURE ONLY).

Plain numpy / torch-CPU restatements of the reference algorithms, used as the
checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under objectdetection_amd/ may import this package: the product path
runs on the HIP library alone and fails loudly without it.

Parity status: pinned by the fixtures in tests/golden/ which were produced by
importing the reference (torch 2.10 CPU) with tests/golden/make_golden.py.
The reference ships no golden vectors of its own, so nothing stronger exists.
Tied NMS scores are unpinned (np.argsort in the reference is unstable); the
oracle defines ties as "descending score, then descending original index".
"""
