#!/bin/bash
# Fan-beam projector: tests, time per apply at 256^2 / 512^2 / 1024^2, per-kernel durations at 512^2.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_operators.py -k "fan" -m gpu -x -q 2>&1 | tail -4
python3 tools/fan_micro.py 2>/dev/null
tools/gpu_prof_cmd.sh tools/fan_micro.py 2>&1 | grep "k_fan" | cut -c1-150
