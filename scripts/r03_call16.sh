#!/bin/bash
# GPU call 16 (round 3): config 5 over a horizon: 3-step DDIM chain at full resolution vs the oracle loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_configs.py -m gpu -q -x -s -k "fullres_ddim_chain" 2>&1 | grep -E "fullres|passed|failed|Error|assert" | tail -10
