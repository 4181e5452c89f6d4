#!/bin/bash
# GPU call 24 (round 3): K = 512 level-1 linears on the 256 x 256 kernel instead of 128 x 128 (A/B of the tile choice), two lanes and one
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2; do for K in -1 512; do for S in 2 1; do
  v=$(timeout 600 python bench.py --streams $S --steps 30 --warmup 5 --no-cpu-baseline --no-extra --min-k-256 $K 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $r: min-k-256 $K (-1 = default), --streams $S: $v"
done; done; done | tee gpurun_out/r03_t_min_k_256.log
