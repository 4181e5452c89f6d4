#!/bin/bash
# GPU call 14 (round 3): repeat-determinism of the v1 denoiser at full occupancy (bf16 and fp8 engines)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_unet.py -m gpu -q -x -k "repeats_bit_equal" 2>&1 | tail -5
