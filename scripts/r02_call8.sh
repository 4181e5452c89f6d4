#!/bin/bash
# GPU call 8 (round 2): 64-row FFN kernel; multi-tile / large-cuboid attention cores; tiny "full" / "divided_st" denoisers;
# guidance autocast option; aligned-step timing
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "ffn or attention or attn" 2>&1 | tail -12
for U in 1 0; do python scripts/bench_ffn.py 0 $U 2>&1 | grep "ffn_fused L0"; done | tee gpurun_out/ffn64_bench.log
for D in 1 2 4 8 14 16 48 112; do python scripts/bench_ffn.py $D 1 2>&1 | grep "ffn_fused L0 B=32"; done | tee -a gpurun_out/ffn64_bench.log
timeout 900 python -m pytest tests/test_hip_unet.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12
timeout 900 python -m pytest tests/test_hip_configs.py -m gpu -x -q -s -k "aligned" 2>&1 | grep "aligned\|passed\|failed" | tail -8
python scripts/time_alignment.py 32 2>&1 | tail -5 | tee gpurun_out/time_alignment.log
for cfg in 32:1 64:2 4:1 8:1 16:2; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['trajectories_per_gpu'], d['config']['lanes'], d['value'], d['ms_per_step'])"
done
