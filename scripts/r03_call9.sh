#!/bin/bash
# GPU call 9 (round 3): VAE ResBlock fused GroupNorm -> SiLU -> Conv2d kernel: parity, two-workgroups-per-CU stress, timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -s -k "conv2d_gn" 2>&1 | grep -v "^$" | tail -12
timeout 900 python -m pytest tests/test_hip_vae.py tests/test_training_side.py -m gpu -q -x -s 2>&1 | grep -E "rel-L2|passed|failed|Error|loss" | tail -12
timeout 600 python scripts/stress_conv2d_gn.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_i_conv2d_gn_stress.log
timeout 600 python scripts/bench_vae.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_i_vae_fused.log
