#!/bin/bash
# MMGKS at 4096^2 (BASELINE C4): wall rate + kernel trace
python tools/hybrid_profile.py mmgks 1e-2 4096 30 2>&1 | head -3
python tools/hybrid_profile.py gks 1e-2 4096 30 2>&1 | head -3
tools/gpu_prof_cmd.sh tools/hybrid_profile.py mmgks 1e-2 4096 30
