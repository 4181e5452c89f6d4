#!/bin/bash
# GPU call 3 (round 3): producer/consumer FFN kernel: parity + A/B vs ffn64 + ablations + end to end
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -s -k "ffn_fused_pc" 2>&1 | grep -v "^$" | tail -12
timeout 300 python scripts/bench_ffn_pc.py 2>&1 | grep FFN | tee gpurun_out/r03_c_ffn_pc_ab.log
for D in 1 2 4 8 14; do timeout 300 python scripts/bench_ffn_pc.py $D 2>&1 | grep "B=32:" | tee -a gpurun_out/r03_c_ffn_pc_ab.log; done
for V in 64 pc; do
PD_FFN_VARIANT=$V timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_c_bench_ffn$V.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_c_bench_ffn$V.json"))
print("ffn $V:", d["value"], d["small_batch"], d["ensemble_strong_scaling"]["value"], d.get("precision_fp32"), d["roofline"].get("in_situ"), d.get("vae"))
PY
done
timeout 900 python -m pytest tests/test_hip_unet.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
