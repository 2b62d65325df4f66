"""Summarise rocprofv3 --pmc CSV output (one dir per pass) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "blur" not in k and "axpby" not in k and "elementwise" not in k and "copy" not in k.lower():
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
