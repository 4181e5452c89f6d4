#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "ffn" 2>&1 | tail -3
python scripts/bench_ffn.py 0 1 2>&1 | grep "ffn_fused L0"
python scripts/bench_ffn.py 1 1 2>&1 | grep "ffn_fused L0 B=32"
for cfg in 32:1 64:2 4:1; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['trajectories_per_gpu'], d['config']['lanes'], d['value'], d['ms_per_step'])"
done
