#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over the level-0 kernels (pair_kernel, attn_block_kernel, ffn64_kernel) at the v1 shapes, 32 trajectories: where do the wave
# cycles go (VALU / MFMA / LDS / VMEM issue, waits)?  -> gpurun_out/pmc_fused.log
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/pmcf
# (PMC_LEVEL1=1: the units-512 instantiation at the level-1 shapes instead)
CMD="python scripts/bench_pair.py 32 ${PMC_LEVEL1:+level1}"      # pair kernel (+ the two round-3 kernels at level 0) at the v1 shapes, 32 trajectories
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
         "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_BUSY_CU_CYCLES" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcf/$tag -o p -- $CMD > gpurun_out/pmcf/$tag.log 2>&1
  f=$(find gpurun_out/pmcf/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "attn_block" not in n and "ffn64" not in n and "pair_kernel" not in n:
        continue
    key = (n[:48], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
done 2>&1 | tee gpurun_out/pmc_fused${PMC_LEVEL1:+_level1}.log
