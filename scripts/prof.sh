#!/bin/bash
# rocprofv3 kernel trace (+stats) of the headline bench configuration; CSV summaries land in gpurun_out/prof/
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps ${STEPS:-10} --warmup 2 --batch ${HB:-16} --no-cpu-baseline ${EXTRA} > gpurun_out/prof_run.log 2>&1
tail -2 gpurun_out/prof_run.log | head -1 | cut -c1-400
find gpurun_out/prof -type f | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/tot*100:6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:110]}")
PY
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
