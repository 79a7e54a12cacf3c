"""CPU oracle for the c2d collision hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (libc2d.so, the host mirror in
``convex-2d-gpu-collision-detection_amd/``, the CLI drivers) never does.

PARITY UNPINNED by the reference's own tests (it has none) — see the header of
``c2d_oracle.c``.
"""
