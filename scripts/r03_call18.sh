#!/bin/bash
# GPU call 18 (round 3): small-grid variant of the attention block (6-slot ring) with all four k-steps' weight fragments in flight
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scripts/bench_fused_opts.py 4,6 2>&1 | grep -v amdgpu | tee gpurun_out/r03_o_deep_pf3.log
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "fused_engine_switches or attn_block" 2>&1 | tail -3
