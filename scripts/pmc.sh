#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the igemm micro-benchmark: HBM traffic + L2 hit rate.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
CMD="python scripts/bench_igemm.py --batch ${HB:-16} --reps 3 --only ${ONLY:-conv3d_l0,conv3d_l1,qkv_l0,ffn2_l0} ${EXTRA}"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc/$tag -o p -- $CMD > gpurun_out/pmc/$tag.log 2>&1
  f=$(find gpurun_out/pmc/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "igemm" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in agg.items():
    print(key, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
done
python scripts/bench_igemm.py --batch ${HB:-16} --only ${ONLY:-conv3d_l0,conv3d_l1,qkv_l0,ffn2_l0} ${EXTRA}
