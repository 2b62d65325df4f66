#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
tools/gpu_prof_cmd.sh tools/fan_one.py 512 2>&1 | grep "k_fan" | cut -c1-150
tools/gpu_pmc_cmd.sh k_fan tools/fan_one.py 512
