#!/bin/bash
# GPU call 4 (round 2): LDS-DMA rate micro-benchmark; FFN with the hand-placed fragment pipeline (parity + timing); attention ablations
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/ubench/dma_rate > gpurun_out/dma_rate.log 2>&1; cat gpurun_out/dma_rate.log
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "ffn" 2>&1 | tail -4
python scripts/bench_ffn.py 0 2>&1 | tail -8 > gpurun_out/ffn_bench.log; cat gpurun_out/ffn_bench.log
for D in 1 2 8 6 10 14; do python scripts/bench_ffn.py $D 2>&1 | grep "ffn_fused L0 B=32"; done >> gpurun_out/ffn_bench.log 2>&1; tail -6 gpurun_out/ffn_bench.log
for D in 1 3 7 15; do python scripts/bench_attn_block.py $D 0 2>&1 | grep "attn_block L0"; done > gpurun_out/attn_bench2.log 2>&1; cat gpurun_out/attn_bench2.log
python bench.py --steps 20 --warmup 3 --batch 32 --streams 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['attention_block'])"
