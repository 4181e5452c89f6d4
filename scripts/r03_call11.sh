#!/bin/bash
# GPU call 11 (round 3): the whole -m gpu suite, the headline bench, the Conv3d kernel per launch with one and with two lanes
# (rocprofv3 kernel trace), and its HBM traffic (separate PMC passes).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03_k_pytest_gpu.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r03_k_bench_headline.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_k_bench_headline.json"))
print("headline", d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], "in_situ", d["roofline"].get("in_situ"), "attn", d.get("attention_block", {}).get("frac"),
      "B4", d.get("small_batch", {}).get("B4", {}).get("value"), "fp32", d.get("precision_fp32", {}).get("value"), "vae", d.get("vae"))
PY
R=$GRAFT_REPO_ROOT
cd /tmp
for S in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_k_prof_lanes$S -o p -- python $R/bench.py --steps 10 --warmup 3 --streams $S --no-cpu-baseline --no-extra > $R/gpurun_out/r03_k_prof_lanes${S}_run.log 2>&1
  tail -1 $R/gpurun_out/r03_k_prof_lanes${S}_run.log | cut -c1-200
  find $R/gpurun_out/r03_k_prof_lanes$S -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-170
done
cd $R
HB=32 bash scripts/pmc_bench.sh 2>&1 | tail -4
