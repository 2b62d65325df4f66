"""HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) under <dir>/pmc_<tag>_<CTR>/.
FETCH_SIZE is doubled (MI355X_MICROARCH.md §HBM: gfx950 reports half the bytes of wide coalesced streaming reads);
WRITE_SIZE is taken as reported.  Both counters are in KiB."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*_FETCH_SIZE"))):
    tag = os.path.basename(d)[4:-len("_FETCH_SIZE")]
    per = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for ctr, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for f in glob.glob(os.path.join(root, f"pmc_{tag}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    per[r["Kernel_Name"][:90]][ctr].append(float(r["Counter_Value"]) * 1024.0 * mult)
    rows = []
    for k, v in per.items():
        fr = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        rows.append((sum(v["FETCH_SIZE"]) + sum(v["WRITE_SIZE"]), k, len(v["FETCH_SIZE"]), fr, wr))
    rows.sort(reverse=True)
    print(f"== {tag}: mean HBM-side bytes per launch (read = FETCH_SIZE x 2, write = WRITE_SIZE)")
    out[tag] = {}
    for tot, k, n, fr, wr in rows[:25]:
        print(f"{k:90s} n={n:5d} read_MB={fr/1e6:10.2f} write_MB={wr/1e6:10.2f} total_over_run_GB={tot/1e9:8.2f}")
        out[tag][k] = {"launches": n, "read_bytes": fr, "write_bytes": wr}
json.dump(out, open(os.path.join(root, "traffic_summary.json"), "w"), indent=1)
