#!/bin/bash
# GPU call 3 (round 2): persistent attention block kernel: parity + A/B timing + phase stamps; default bench
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attn_block" 2>&1 | tail -8 > gpurun_out/pytest_gpu3.log
tail -4 gpurun_out/pytest_gpu3.log
timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_sampler.py -m gpu -x -q 2>&1 | tail -4
for P in 256 0 512; do python scripts/bench_attn_block.py 0 $P 2>&1 | tail -4; done > gpurun_out/attn_bench.log 2>&1
for D in 16 48 112 2 4 8; do python scripts/bench_attn_block.py $D 256 2>&1 | grep "attn_block L0"; done >> gpurun_out/attn_bench.log 2>&1
cat gpurun_out/attn_bench.log
python bench.py --steps 20 --warmup 3 --batch 32 --streams 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | cut -c1-2000 > gpurun_out/bench_b32s1.json
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_default.json
python - <<'PY'
import json
for f in ("gpurun_out/bench_b32s1.json", "gpurun_out/bench_default.json"):
    d = json.loads(open(f).read()); print(f, d["value"], d["ms_per_step"], d.get("attention_block"), d.get("ensemble_strong_scaling"), d.get("small_batch"))
PY
