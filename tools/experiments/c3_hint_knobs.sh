for v in "" "TRK_RADON_NO_XT_OUT=1" "TRK_RADON_NO_REC_OUT=1" "TRK_RADON_NO_XT_OUT=1 TRK_RADON_NO_REC_OUT=1"; do
  echo "== $v"
  env $v python tools/hybrid_profile.py lsqr 1e-2 512 100 2>&1 | grep it/s
  env $v python tools/hybrid_profile.py lsqr 1e-2 512 100 2>&1 | grep it/s
done
export TRK_RADON_NO_XT_OUT=1 TRK_RADON_NO_REC_OUT=1
bash tools/gpu_prof_cmd.sh tools/c3_trace.py > /dev/null; python3 tools/trace_timeline.py gpurun_out/prof_cmd 10 60
