#!/bin/bash
# GPU call 20 (round 3): whole-tile temporal tap skipping in the 256 x 256 Conv3d kernel: parity, then A/B against the dense loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "igemm or conv3d" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_hip_unet.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2 3; do
  for D in 0 8; do
    v=$(timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --igemm-debug $D 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'])")
    echo "round $r igemm-debug $D (8 = dense tap loop): $v"
  done
done | tee gpurun_out/r03_q_tap_skip_ab.log
