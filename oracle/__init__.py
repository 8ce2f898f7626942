"""CPU oracle for the pos-evolution hot path -- TEST INFRASTRUCTURE ONLY.

``oracle.spec``   L0: literal pyspec restatement (pure Python, small cases).
``oracle.g1``     L0: exact BLS12-381 G1 arithmetic on Python ints.
``oracle.cport``  L1: ctypes binding of ``posevo_oracle.c`` (plain C restatement,
                  same results as L0, finishes the BASELINE.json sizes in seconds;
                  this is what ``bench.py`` times as ``cpu_baseline`` kind "port").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path never does.
"""
