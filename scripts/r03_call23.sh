#!/bin/bash
# GPU call 23 (round 3): lanes at small per-GPU batches (strong-scaling line): 4 and 8 trajectories, 1 vs 2 lanes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2; do for B in 4 8; do for S in 1 2; do
  v=$(timeout 600 python bench.py --batch $B --streams $S --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['lanes'])")
  echo "round $r: $B trajectories, --streams $S: $v"
done; done; done | tee gpurun_out/r03_s_small_batch_lanes.log
