#!/bin/bash
# GPU call 12 (round 3): do the two lanes run in lock step (Conv3d against Conv3d)?  Lane 1 starts its timed steps late by a fixed offset.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2; do
for MS in 0 0.4 1 2 5 12; do
  v=$(PD_LANE_STAGGER_MS=$MS timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $r stagger ${MS} ms: $v"
done
done | tee gpurun_out/r03_l_lane_stagger.log
