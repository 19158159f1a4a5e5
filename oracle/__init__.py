"""CPU oracle of the filtered vector-search hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; nothing under ``search_engine_amd/`` does.

Modules
  scan      ctypes front end of ``scan_oracle.c`` (arroy 0.2.0 Angular +
            exhaustive nns_by_vector, Hamming top-k) and an independent numpy
            implementation of the same definitions
  host      pure-Python restatement of the reference's host logic around the
            scan (VectorSort, DistributionShift, score_details, hybrid merge)
  roaring   pure-Python restatement of the bitmap wire formats

PARITY UNPINNED at the arroy boundary: the reference holds no golden vector for
this path and cannot be built here; see the header of ``scan_oracle.c``.
"""
