#!/bin/bash
# GPU call 10 (round 2): profiles of the current engine: rocprofv3 kernel stats of one lane (32 trajectories), the default two-lane
# command, 4 trajectories; whole sample() incl. VAE under rocprof; PMC HBM traffic of the Conv3d kernel
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
EXTRA="--batch 32 --streams 1 --no-extra" STEPS=10 OUT=prof_lane bash scripts/prof.sh
EXTRA="--no-extra" STEPS=10 OUT=prof bash scripts/prof.sh
EXTRA="--batch 4 --streams 1 --no-extra" STEPS=10 OUT=prof_b4 bash scripts/prof.sh
rm -rf gpurun_out/prof_sample
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sample -o trace -- python scripts/time_sample.py 32 > gpurun_out/prof_sample_run.log 2>&1
tail -3 gpurun_out/prof_sample_run.log
find gpurun_out/prof_sample -name "*kernel_trace.csv" -size +30M -delete
HB=32 bash scripts/pmc_bench.sh 2>&1 | tail -4
