"""Busy time and gaps of the LAST `n` kernel launches in a rocprofv3 --kernel-trace CSV: per kernel name count / mean duration, then
how much of the span between the first start and the last end the device was executing kernels.  usage: trace_gaps.py <csv> [n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(sum(v) for v in by.values())
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{len(v):5d} x {sum(v) / len(v) / 1e3:8.2f} us  = {sum(v) / 1e3:9.1f} us  {k}")
print(f"span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({busy / span:.2f}), launches {len(rows)}")
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gaps.sort()
print("gaps us: median %.2f  p90 %.2f  max %.2f  sum %.1f" % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3, gaps[-1] / 1e3, sum(gaps) / 1e3))
