#!/bin/bash
export TRK_RADON_WIN_MIN=512
for b in 128 64 32; do
export TRK_RADON_BAND=$b
echo "=== quad kernel forced at 512^2, band $b"
$GRAFT_REPO_ROOT/tools/r04_c3trace.sh 2>&1 | grep -v "^E2026\|^W2026" | grep "k_radon\|memset\|Memset\|fill\|span\|it/s\|transpose" | cut -c1-110
done
