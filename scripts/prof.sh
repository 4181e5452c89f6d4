#!/bin/bash
# rocprofv3 kernel trace (+stats) of bench.py (default configuration, or EXTRA="--batch B --streams S"); CSV summaries land in gpurun_out/$OUT/
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=${OUT:-prof}
rm -rf gpurun_out/$OUT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o trace -- python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-extra ${EXTRA} > gpurun_out/${OUT}_run.log 2>&1
tail -2 gpurun_out/${OUT}_run.log | head -1 | cut -c1-400
find gpurun_out/$OUT -type f | head
f=$(find gpurun_out/$OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/tot*100:6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:110]}")
PY
find gpurun_out/$OUT -name "*kernel_trace.csv" -size +30M -delete
