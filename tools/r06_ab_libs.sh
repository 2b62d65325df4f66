#!/bin/bash
# builds A/B variants of libtrk.so under tools/experiments/lib/ for TRK_EXPERIMENT_LIB (round 6).  Runs where hipcc is.
# usage: tools/r06_ab_libs.sh name=[file:]"-Dswitches" ...   (file: the one source recompiled with the switches; default radon2d)
cd "$(dirname "$0")/.."
mkdir -p tools/experiments/lib
SRC="core vecops blur2d tvops radon2d spmv fanbeam2d projected cgls_loop comm cgls_tiled cgls_sharded ref64"
for v in "$@"; do
  name=${v%%=*}; defs=${v#*=}; file=radon2d
  case "$defs" in *:*) file=${defs%%:*}; defs=${defs#*:};; esac
  objs=""
  for f in $SRC; do
    if [ $f = $file ]; then
      hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -Itrips_py_amd/csrc $defs -c trips_py_amd/csrc/$f.hip -o /tmp/ab_${name}_$f.o 2>/dev/null || exit 1
      objs="$objs /tmp/ab_${name}_$f.o"
    else
      objs="$objs trips_py_amd/csrc/$f.o"
    fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/lib/libtrk_$name.so $objs -ldl && echo "built tools/experiments/lib/libtrk_$name.so ($defs)"
done
