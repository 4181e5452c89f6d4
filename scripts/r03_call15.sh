#!/bin/bash
# GPU call 15 (round 3): where the guidance gradient's time goes now (torch profiler), fp32-class and bf16 convolution operands
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/exp_alignment_profile.py 32 2>&1 | grep -v amdgpu | tail -40 | cut -c1-220 | tee gpurun_out/r03_n_alignment_profile.log
