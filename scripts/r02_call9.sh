#!/bin/bash
# GPU call 9 (round 2): pd_igemm tile sweep at the level-1 shapes (32 and 4 trajectories); split-K for the level-1 FFN-2 at small batch
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/igemm_tiles.log
for B in 32 4; do for T in 0 1 3 4 5 6 7 8 9 2; do
  echo "== batch $B tile $T" >> gpurun_out/igemm_tiles.log
  python scripts/bench_igemm.py --batch $B --tile $T --only qkv_l1,proj_l1,ffn1_l1,ffn2_l1 2>&1 | grep "TFLOP" >> gpurun_out/igemm_tiles.log
done; done
cat gpurun_out/igemm_tiles.log
for cfg in 4:1 8:1 2:1 1:1; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['trajectories_per_gpu'], d['config']['lanes'], d['value'], d['ms_per_step'])"
done
