#!/bin/bash
# GPU call 7b (round 3): fp8 = e4m3 convolutions + K >= 512 token linears, fp8_conv = convolutions only: parity + bench + kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_configs.py -m gpu -q -x -s -k "ddim50 or fullres or fp8_conv" 2>&1 | grep -E "rel-L2|passed|failed|Error|assert" | tail -14
for P in fp8 fp8_conv; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $P --no-extra 2>&1 | tail -1 > gpurun_out/r03_g_bench_v1_$P.json
timeout 900 python bench.py --config fullres --steps 10 --warmup 2 --no-cpu-baseline --no-extra --precision $P 2>&1 | tail -1 > gpurun_out/r03_g_bench_fullres_$P.json
done
timeout 900 python bench.py --config fullres --steps 10 --warmup 2 --no-cpu-baseline --no-extra --precision bf16 2>&1 | tail -1 > gpurun_out/r03_g_bench_fullres_bf16.json
python - <<PY
import json
for f in ("v1_fp8","v1_fp8_conv","fullres_fp8","fullres_fp8_conv","fullres_bf16"):
    d=json.load(open(f"gpurun_out/r03_g_bench_{f}.json")); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_prof_fullres_fp8 -o fr -- python $GRAFT_REPO_ROOT/bench.py --config fullres --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_fullres_fp8_run.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r03_prof_fullres_fp8 -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-160
