"""CPU oracle for the PyRHE streaming trace-sketch path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pyrhe_amd/`` may import this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.  See ``oracle/README.md``.
"""
