"""Idle time between kernels of a rocprofv3 --kernel-trace csv: total busy / idle over the last `count` launches and the largest gaps.
usage: trace_gaps.py <dir> [count] [skip_from_end]"""
import csv, glob, sys
root = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = rows[max(0, len(rows) - skip - count): len(rows) - skip]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
gaps = []
for a, b in zip(sel, sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
print(f"{len(sel)} launches, span {span/1e6:.3f} ms, kernels {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms ({100*(span-busy)/span:.1f} %)")
big = sorted(gaps, reverse=True)[:12]
for g, a, b in big:
    print(f"  gap {g/1e3:8.1f} us   after {a}   before {b}")
import collections
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<1us" if g < 1000 else "1-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"] += 1
print(dict(hist))
