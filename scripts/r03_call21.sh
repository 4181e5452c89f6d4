#!/bin/bash
# GPU call 21 (round 3): tap skipping == dense tap loop, bit for bit
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "tap_skip" 2>&1 | tail -5
