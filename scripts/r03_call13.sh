#!/bin/bash
# GPU call 13 (round 3): smoke(), and bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, RCCL) with one rank
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 2>&1 | tail -2 | cut -c1-1500
echo "--- self-launch with more ranks than devices must fail:"
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1; echo "exit code $?"
