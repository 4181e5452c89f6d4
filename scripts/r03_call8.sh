#!/bin/bash
# GPU call 8 (round 3): the whole -m gpu suite after the round's changes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r03_h_pytest_gpu.log
