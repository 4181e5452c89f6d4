#!/bin/bash
# GPU call 2 (round 2): split-K Conv3d + GroupNorm partial reduction: parity, small-batch sweep, B=4 profile
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_sampler.py tests/test_hip_vae.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest_gpu2.log
tail -4 gpurun_out/pytest_gpu2.log
rm -f gpurun_out/sweep_splitk.jsonl
for cfg in 1:1:128 1:1:0 2:1:128 4:1:128 4:1:0 4:2:128 4:1:64 8:1:128 8:1:64 8:2:128 8:1:0 16:1:128 16:1:0 16:2:128 16:2:0 32:2:128; do
  IFS=: read B S T <<< "$cfg"
  python bench.py --steps 20 --warmup 3 --batch $B --streams $S --splitk-max-tiles $T --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['splitk_max_tiles']=$T; print(json.dumps(d))" >> gpurun_out/sweep_splitk.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/sweep_splitk.jsonl"):
    d = json.loads(l); print(d["config"]["trajectories_per_gpu"], d["config"]["lanes"], d["splitk_max_tiles"], d["value"], d["ms_per_step"])
PY
EXTRA="--batch 4 --streams 1 --no-extra" STEPS=10 OUT=prof_b4 bash scripts/prof.sh
