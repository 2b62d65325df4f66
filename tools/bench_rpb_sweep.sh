# CGLS iterations/s and in-loop blur kernel times against the forced band height (TRK_BLUR_RPB); "def" = the library's choice
# usage: bench_rpb_sweep.sh SIZE [rpb ...]
N=${1:-4096}; shift
for r in ${@:-def 19 28 37 46 55 64 73 82 100}; do
  if [ $r = def ]; then unset TRK_BLUR_RPB; else export TRK_BLUR_RPB=$r; fi
  python bench.py --size $N --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('N=$N rpb=$r', d['value'], 'it/s  fwd', r['avg_kernel_us'], 'us  adj', r.get('adjoint_matvec_avg_kernel_us'), r['kernel'][:34])"
done
