"""C2 (512^2 blur, CGLS 100 iterations) repeated: run-to-run spread of the small-image rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for i in range(6):
    print(bench.extra_c2_blur512(1))
