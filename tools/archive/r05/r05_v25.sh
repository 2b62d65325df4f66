#!/bin/bash
# round 5: the band-resident forward projector (k_radon_fwd_band) against the per-wave windows: tests, then C3 / C5 rates and kernel times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v25; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -q -x -m gpu -k "radon or tomo or c3 or c5 or dyn or operators or ref64 or proj" 2>&1 | tail -6
for nb in 0 1; do
  echo "== TRK_RADON_NO_BANDRES=$nb"
  if [ $nb = 1 ]; then export TRK_RADON_NO_BANDRES=1; else unset TRK_RADON_NO_BANDRES; fi
  timeout 600 python3 tools/configs_micro.py c3 2>&1 | grep -i "it/s\|C3"
  timeout 600 python3 tools/configs_micro.py c5 2>&1 | grep -i "it/s\|C5"
done 2>&1 | tee $O/rates.txt
unset TRK_RADON_NO_BANDRES
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 $R/tools/configs_micro.py c3 > $O/prof_c3.log 2>&1
f=$(ls -t $O/prof_c3/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv
python3 $R/tools/stats_top.py $O/c3_kernel_stats.csv 2>/dev/null | head -8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/tools/configs_micro.py c5 > $O/prof_c5.log 2>&1
f=$(ls -t $O/prof_c5/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv
python3 $R/tools/stats_top.py $O/c5_kernel_stats.csv 2>/dev/null | head -8
rm -rf $O/prof_c3 $O/prof_c5
