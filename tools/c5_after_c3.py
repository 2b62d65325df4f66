"""Does what runs before C5 in the bench disturb its timed CGLS solve?  C3's extras, then C5 three times; again with the cyclic GC off."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "nogc":
    gc.disable()
if mode != "alone":
    r = bench.extra_c3_tomo(1)
    print({k: v for k, v in r.items() if "iters" in k})
for i in range(4):
    r = bench.extra_c5_dynamic(0, 1)
    print(mode, i, r["cgls_iters_per_sec"], r["gks_iters_per_sec"], "gc counts", gc.get_count())
