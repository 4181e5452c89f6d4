#!/bin/bash
# Run on the GPU box through gpurun: sweep of bench.py over trajectories/GPU x lanes + rocprofv3 kernel traces of the headline
# configuration (default: 64 trajectories as 2 lanes of 32) and of one lane in isolation (the roofline object's measurement).
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
rm -f gpurun_out/sweep.jsonl
SWEEP=${SWEEP:-1:1 4:1 16:1 32:1 64:1 32:2 64:2 78:2 128:2 128:4}
for cfg in $SWEEP; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/sweep.jsonl
done
python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_headline.json
STEPS=10 OUT=prof bash scripts/prof.sh                                 # the default command: 2 lanes x 32, kernels overlap
EXTRA="--batch 32 --streams 1" STEPS=10 OUT=prof_lane bash scripts/prof.sh     # one lane alone: per-kernel durations without overlap
