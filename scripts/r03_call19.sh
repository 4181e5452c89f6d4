#!/bin/bash
# GPU call 19 (re-run as call 22 after the tap-skipping change) (round 3): the whole -m gpu suite and the default bench at the final commit of the round
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03_r_pytest_gpu.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r03_r_bench_headline.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_r_bench_headline.json"))
print("headline", d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], "in_situ", d["roofline"]["in_situ"]["frac"], "attn", d["attention_block"]["frac"],
      "B4", d["small_batch"]["B4"]["value"], "fp32", d["precision_fp32"]["value"], "vae", d["vae"]["encode"]["frac"], d["vae"]["decode"]["frac"])
PY
