"""CPU oracle for the LoIK hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``loik_amd``) never does.  PARITY UNPINNED: see ``oracle/loik_ref.h``.
"""
