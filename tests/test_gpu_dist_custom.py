"""-m gpu: the engine-owned sharded step with TWO ranks on one GPU (VERDICT r2 "make the N > 1 path
verifiable"): pe_dist_init_custom carries the two exchange steps over gloo, staged through the host; every rank checks
every step against an unsharded twin (tests/dist_worker.py).  "committee" = the committee-sharded step (pe_aggregate over
the rank's own committees + pe_aggregate_exchange, SURVEY.md 8e Option B): no G1 collective, no weight all-reduce, every rank
ends each step with the unsharded store."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rows_mode,pipe_mode", [("device", "lagged"), ("device", "plain"), ("host", "lagged"),
                                                 ("host", "pipelined"), ("committee", "lagged"), ("committee", "plain")])
def test_engine_owned_sharded_step_two_ranks_one_gpu(rows_mode, pipe_mode):
    port = 29650 + (hash((rows_mode, pipe_mode)) % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), rows_mode, pipe_mode]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.count("DIST_WORKER_OK") == 2, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
