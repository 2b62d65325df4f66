"""Condense a tools/gpu_profile.sh output directory: per-kernel stats (rocprofv3 --kernel-trace --stats) and HBM-side
traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM: on gfx950 it
reports half of the bytes of a wide coalesced streaming read)."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]


def newest(pattern):
    """gpurun merges every run's files into the same local directory: only the most recent run counts."""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in newest(os.path.join(root, "trace", "**", "*kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:8.2f} max_us={float(r['MaxNs'])/1e3:8.2f} pct={r['Percentage']}")
traffic = {}
for name, sub, mult in (("FETCH_SIZE", "pmc_fetch", 2.0), ("WRITE_SIZE", "pmc_write", 1.0)):
    acc = collections.defaultdict(list)
    for f in newest(os.path.join(root, sub, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    print(f"== {name} per launch (KiB as reported; x{mult} correction applied in the MB column)")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
        mean = sum(v) / len(v)
        print(f"{k:70s} n={len(v):4d} mean_KiB={mean:12.1f}  corrected_MB={mean*1024*mult/1e6:9.2f}")
        traffic.setdefault(k, {})[name] = mean * 1024 * mult
print("== traffic per launch (bytes) for the blur kernel")
for k, d in traffic.items():
    if "k_blur_slide<9, 9, 9, true, false>" in k:
        tot = d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)
        print(json.dumps({"k_blur_slide_fwd": tot, "read": d.get("FETCH_SIZE"), "write": d.get("WRITE_SIZE")}))
