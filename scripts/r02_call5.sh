#!/bin/bash
# GPU call 5 (round 2): 64-row / 2-workgroups-per-CU attention block kernel: parity, timing, phase stamps, ablations, end-to-end
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attn_block" 2>&1 | tail -12 > gpurun_out/pytest_gpu5.log
tail -12 gpurun_out/pytest_gpu5.log
python scripts/bench_attn_block.py 0 2>&1 | tail -4 > gpurun_out/attn_bench5.log
for D in 16 48 112 1 2 4 8 15; do python scripts/bench_attn_block.py $D 2>&1 | grep "attn_block L0"; done >> gpurun_out/attn_bench5.log 2>&1
for B in 4 8 16; do python scripts/bench_attn_block.py 0 0 $B 2>&1 | grep "attn_block L0"; done >> gpurun_out/attn_bench5.log 2>&1
cat gpurun_out/attn_bench5.log
timeout 900 python -m pytest tests/test_hip_unet.py -m gpu -x -q 2>&1 | tail -3
for cfg in 32:1 64:2 4:1 4:2; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['trajectories_per_gpu'], d['config']['lanes'], d['value'], d['ms_per_step'], d['attention_block']['frac'], d['attention_block']['avg_launch_us'])"
done
