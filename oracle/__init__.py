"""CPU oracle for the CARL hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``carl_amd`` never does.
"""
