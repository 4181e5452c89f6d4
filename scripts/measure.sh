#!/bin/bash
# Run on the GPU box through gpurun: batch sweep of bench.py + rocprofv3 kernel trace of the headline configuration.
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for B in ${SWEEP:-1 4 16 32}; do
  python bench.py --steps 20 --warmup 3 --batch $B --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/sweep.jsonl
done
python bench.py --steps 50 --warmup 5 --batch ${HB:-16} 2>&1 | tail -1 | tee gpurun_out/bench_headline.json
HB=${HB:-32} STEPS=10 bash scripts/prof.sh
