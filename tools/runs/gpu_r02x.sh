#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02x_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02x_tests.log
tail -n 30 gpurun_out/r02x_tests.log
