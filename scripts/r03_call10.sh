#!/bin/bash
# GPU call 10 (round 3): fused level-0 kernels built with / without SLP vectorisation (packed-f32 VALU beside the MFMAs), interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2; do
  echo "== round $r: default build"
  timeout 600 python scripts/bench_fused_opts.py 4 2>&1 | grep -v amdgpu
  echo "== round $r: ffn.hip + attn_block.hip with -fno-slp-vectorize"
  PD_LIB_PATH=$PWD/prediff_amd/libprediff_hip_noslp.so timeout 600 python scripts/bench_fused_opts.py 4 2>&1 | grep -v amdgpu
done | tee gpurun_out/r03_j_noslp_ab.log
