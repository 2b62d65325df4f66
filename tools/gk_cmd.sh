set -e
python -m pytest tests -q -m gpu -x -k "solver or hybrid or golden" 2>&1 | tail -2
for n in "lsqr 1e-2 512 100" "lsqrb 1e-2 4096 50" "lsqrb 1e-2 1024 100"; do
  python tools/hybrid_profile.py $n 2>&1 | grep "it/s"
  GK_NORMALIZED=1 python tools/hybrid_profile.py $n 2>&1 | grep "it/s" | sed 's/^/  [normalised storage] /'
done
