"""Print the kernel timeline of the LAST `count` launches of a rocprofv3 --kernel-trace csv: name, duration, gap to the
previous kernel's end.  usage: trace_timeline.py <dir> [count] [skip_from_end]"""
import csv, glob, sys
root = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = rows[len(rows) - skip - count: len(rows) - skip]
prev = None
tot = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:90]:90s} dur={(e - s) / 1e3:8.2f} us  gap={gap:7.2f} us  grid={r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')} wg={r.get('Workgroup_Size_X','?')}")
    prev = e
print(f"span {(int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3:.1f} us for {len(sel)} launches")
