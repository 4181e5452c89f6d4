#!/bin/bash
# GPU call 2 (round 3): fused-kernel engine switches (atomic epilogue, deep ring, arithmetic token ids): parity + A/B
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attn_block or ffn or fused_engine" 2>&1 | tail -8
timeout 600 python scripts/bench_fused_opts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_b_fused_opts_ab.log
for O in 0 7; do
PD_FUSED_OPTS=$O timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_b_bench_opts$O.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_b_bench_opts$O.json"))
print("opts $O:", d["value"], d["attention_block"]["frac"], d["attention_block"]["avg_launch_us"], d["small_batch"], d["ensemble_strong_scaling"]["value"])
PY
done
