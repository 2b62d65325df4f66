"""Top rows of a rocprofv3 kernel_stats.csv with the kernel names cut short.  usage: stats_top.py <csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f'{int(r["Calls"]):6d} x {float(r["AverageNs"]) / 1e3:9.2f} us = {int(r["TotalDurationNs"]) / 1e6:8.2f} ms {float(r["Percentage"]):6.2f} %  {r["Name"][:90]}')
print(f"total {tot / 1e6:.2f} ms")
