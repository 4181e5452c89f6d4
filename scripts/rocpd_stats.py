"""rocprofv3 (ROCm 7.2 default output: a rocpd SQLite database) -> the kernel_stats.csv table of `--stats --output-format csv`.
    python scripts/rocpd_stats.py <results.db> [out.csv]"""
import csv, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = {}
for name, dur in db.execute("select name, duration from kernels"):
    rows.setdefault(name, []).append(dur)
tot = sum(sum(v) for v in rows.values())
table = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v), statistics.pstdev(v)) for n, v in rows.items()),
               key=lambda r: -r[2])
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for r in table:
    w.writerow([r[0], r[1], r[2], f"{r[3]:.6f}", f"{r[4]:.2f}", r[5], r[6], f"{r[7]:.6f}"])
