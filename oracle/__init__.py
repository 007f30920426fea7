"""CPU oracle for the lshrs hot path (TEST INFRASTRUCTURE — never shipped, never timed as product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product package ``lshrs_amd`` must never do so.
"""
