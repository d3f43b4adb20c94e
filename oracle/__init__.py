"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU checkers for the voxel graph-cut hot path.  Nothing under ``medpy_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may.  See ``oracle/README.md``.
"""
