#!/bin/bash
# GPU call 1 (round 2): full gpu test suite, default bench, small-batch lane sweep, rocprof of B=4
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_headline.json | cut -c1-300
rm -f gpurun_out/sweep_small.jsonl
for cfg in 1:1 2:1 2:2 4:1 4:2 4:4 8:1 8:2 8:4 16:1 16:2 16:4; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 >> gpurun_out/sweep_small.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/sweep_small.jsonl"):
    try:
        d = json.loads(l); print(d["config"]["trajectories_per_gpu"], d["config"]["lanes"], d["value"], d["ms_per_step"])
    except Exception as e:
        print("bad line", l[:200])
PY
EXTRA="--batch 4 --streams 1 --no-extra" STEPS=10 OUT=prof_b4 bash scripts/prof.sh
EXTRA="--batch 32 --streams 1 --no-extra" STEPS=10 OUT=prof_lane bash scripts/prof.sh
