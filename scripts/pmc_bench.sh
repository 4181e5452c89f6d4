#!/bin/bash
# HBM traffic of the Conv3d kernel inside the real bench (separate PMC passes, kernel-trace only) -> profiles/conv3d_hbm_traffic.json
export TMPDIR=/tmp
B=${HB:-32}
PREC=${PREC:-bf16}          # bf16 | fp8 (e4m3 Conv3d operands): the result goes under key B<batch> / B<batch>_fp8
rm -rf gpurun_out/pmcb
mkdir -p gpurun_out/pmcb
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcb/$C -o p -- python bench.py --steps 3 --warmup 1 --batch $B --streams 1 --no-cpu-baseline --no-graph --no-extra --precision $PREC > gpurun_out/pmcb/$C.log 2>&1
done
python - $B $PREC <<'PY'
import csv, glob, json, sys
B = sys.argv[1] + ("" if sys.argv[2] == "bf16" else "_" + sys.argv[2])
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmcb/{c}/*counter_collection.csv")
    def is_conv3d(name):   # igemm256_kernel<2> or igemm_kernel<BM, BN, BK, NS, SPLIT, 2, ...>
        if "igemm256_kernel<2" in name:
            return True
        return "igemm_kernel<" in name and name.split("<")[1].split(">")[0].split(",")[5].strip() == "2"
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if is_conv3d(r["Kernel_Name"])]
    out[c] = sum(vals) / max(1, len(vals))
    print(c, "avg per conv3d launch (KB):", out[c], "n =", len(vals))
# guide (MI355X_MICROARCH.md §HBM): FETCH_SIZE is in KB and under-reports wide coalesced reads by 2x on gfx950; WRITE_SIZE in KB
traffic = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
json.dump({f"B{B}": round(traffic), "raw_kb": out, "note": "bytes per Conv3d launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction)"},
          open(f"gpurun_out/conv3d_hbm_traffic_{sys.argv[2]}.json", "w"))
print("traffic bytes/launch", traffic)
PY
