#!/bin/bash
# A/B on the GPU box: steps/s per trajectory count with the level-1 (units 512) pairs on pd_attn_ffn_pair vs as separate launches
for B in ${BATCHES:-1 2 4 8 12 16 24 32}; do
  for U in 256 256,512; do
    v=$(PD_PAIR_UNITS=$U PD_PAIR_L1_MIN_TILES=0 python bench.py --batch $B --streams 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'])")
    echo "B=$B units=$U steps/s=$v"
  done
done
