#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/sweep_quant.jsonl
for cfg in 64:2 78:2 39:1 76:2 80:2 117:3 72:2 70:2; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 >> gpurun_out/sweep_quant.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/sweep_quant.jsonl"):
    d = json.loads(l); print(d["config"]["trajectories_per_gpu"], d["config"]["lanes"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["attention_block"]["frac"])
PY
