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
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 10 --warmup 2 --batch ${HB:-16} --no-cpu-baseline > gpurun_out/prof_run.log 2>&1
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-220
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
