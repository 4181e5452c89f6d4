#!/bin/bash
# GPU call 1 (round 3): new config tests (100-step aligned chain, lane split, tightened guard rails), baseline bench + B=4 kernel stats
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_configs.py -m gpu -x -q -s -k "chain or lane_split or ddim50 or fullres or fp8_conv" 2>&1 | grep -v "^$" | tail -20
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/r03_a_bench_headline.json; cat gpurun_out/r03_a_bench_headline.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_prof_b4 -o b4 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --batch 4 --streams 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_b4_run.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r03_prof_b4_run.log
find $GRAFT_REPO_ROOT/gpurun_out/r03_prof_b4 -name "*kernel_stats.csv" | head -1 | xargs head -40
