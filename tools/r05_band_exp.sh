#!/bin/bash
# round 5: where the band-resident forward projector spends its time at 512^2 x 180 — its band load alone, its march alone.
# Builds a SEPARATE library with -DTRK_RADON_BAND_EXPERIMENT (results are wrong by construction: timing only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bexp; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
C=$R/trips_py_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DTRK_RADON_BAND_EXPERIMENT -I$R/include -I$C -c $C/radon2d.hip -o /tmp/radon_exp.o || exit 1
objs=$(ls $C/*.o | grep -v radon2d.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtrk_exp.so /tmp/radon_exp.o $objs -ldl || exit 1
export TRK_EXPERIMENT_LIB=/tmp/libtrk_exp.so
cd /tmp
for x in 0 2 4; do
  echo "== TRK_RADON_BAND_X=$x  (2: the band load alone, 4: the march alone)"
  export TRK_RADON_BAND_X=$x
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$x -- python3 $R/tools/radon_small.py 512 > $O/p$x.log 2>&1
  f=$(ls -t $O/p$x/*/*kernel_stats.csv | head -1); python3 $R/tools/stats_top.py $f | grep "fwd_band\|bands_post"
done 2>&1 | tee $O/band_exp.txt
rm -rf $O/p0 $O/p2 $O/p4
