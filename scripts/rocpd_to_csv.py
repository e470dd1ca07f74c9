"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database; this prints its per-kernel summary (the `top_kernels` view: name, calls, total ns... in us) as CSV
for profiles/.  usage: python scripts/rocpd_to_csv.py results.db > profiles/NAME_kernel_stats.csv"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for r in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 4)])
