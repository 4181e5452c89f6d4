#!/bin/bash
# GPU call 7 (round 2): attention (tables behind the row loads) + FFN (bias folded into GEMM-1) parity and timing; full-res
# geometry (BASELINE config 5, bf16): parity test + bench line; lane sweep
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attn_block or ffn" 2>&1 | tail -12
python scripts/bench_attn_block.py 0 2>&1 | grep "attn_block L0\|phase"
python scripts/bench_ffn.py 0 2>&1 | tail -7
timeout 1200 python -m pytest tests/test_hip_configs.py -m gpu -x -q -s -k "fullres" 2>&1 | tail -8
timeout 900 python bench.py --config fullres --steps 5 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_fullres.json | cut -c1-1500
rm -f gpurun_out/sweep_lanes.jsonl
for cfg in 32:1 64:2 96:3 64:4 128:4 48:2 96:2; do
  B=${cfg%%:*}; S=${cfg##*:}
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --no-cpu-baseline --no-extra 2>&1 | tail -1 >> gpurun_out/sweep_lanes.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/sweep_lanes.jsonl"):
    d = json.loads(l); print(d["config"]["trajectories_per_gpu"], d["config"]["lanes"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["attention_block"]["frac"])
PY
