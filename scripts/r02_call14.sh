#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_alignment.py -m gpu -x -q -s 2>&1 | grep -v "^$\|MIOpen\|amdgpu" | tail -8
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -x -q -s -k "aligned" 2>&1 | grep "aligned\|passed\|failed" | tail -8
python scripts/time_alignment.py 32 2>&1 | grep -v "MIOpen\|amdgpu" | tail -4 | tee gpurun_out/time_alignment.log
python scripts/time_alignment.py 8 2>&1 | grep -v "MIOpen\|amdgpu" | tail -4 | tee -a gpurun_out/time_alignment.log
