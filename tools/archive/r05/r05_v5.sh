#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v5; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/blur_tile.hip -o /tmp/blur_tile 2> $O/blur_tile_build.log; echo "blur_tile build rc=$?"; tail -3 $O/blur_tile_build.log
timeout 120 /tmp/blur_tile 4096 6 | tee $O/blur_tile_4096.txt
timeout 120 /tmp/blur_tile 2048 12 | tee $O/blur_tile_2048.txt
timeout 600 python -m pytest tests/test_gpu_operators.py -m gpu -q -x -k "sparse or csr or framelet" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python3 tools/spmv_micro.py 2>/dev/null | tee $O/spmv_micro.txt
for g in 8 16 32 64; do echo "TRK_CSR_GROUP=$g"; TRK_CSR_GROUP=$g timeout 600 python3 tools/spmv_micro.py 2>/dev/null | head -2; done | tee $O/spmv_micro_groups.txt
