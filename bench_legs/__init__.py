"""The legs of bench.py: the workload and the step (workload), the CPU baselines (cpu), the checks against the oracle
and the synchronous replay (verify), the N > 1 steps and their checks (sharded), the signed steps (signed) and the
per-slot cadence (slots).  bench.py itself holds the argument parsing, the timed region and the JSON line."""
