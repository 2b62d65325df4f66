#!/bin/bash
# CGLS over image sizes: one bench line per size
for n in 256 512 1024 2048 3072 4096 5120 8192; do
  python bench.py --size $n --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
rf=r['roofline']
print($n, r['config'].get('iteration','')[:60], round(r['value'],1), 'it/s', r['ms_per_step'], 'ms', rf.get('avg_kernel_us'), 'us', rf.get('achieved'), 'GB/s', rf.get('frac'), r['extra'].get('cgls_effective_GBps'))"
done
