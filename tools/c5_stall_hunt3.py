"""extra_c5_dynamic's own sequence, twelve times, with the host-return / device-idle times of its timed CGLS solve; variants switch
single ingredients off: 'noeng' re-uses one engine, 'nocpu' builds the phantom on the device, 'nol' skips the regulariser,
'threads8' / 'threads1' limit torch's CPU thread pool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.engine import HipEngine
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel, SpaceTimeDerivative
from trips_py_amd.solvers import CGLS, GKS
mode = set(sys.argv[1:])
if "threads8" in mode:
    torch.set_num_threads(8)
if "threads1" in mode:
    torch.set_num_threads(1)
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup:", e)
Nf, nt, na = 256, 32, 15
shared = HipEngine()
def one():
    eng = shared if "noeng" in mode else HipEngine()
    ops = [Radon2DParallel(Nf, np.deg2rad(t + 12.0 * np.arange(na)), engine=eng) for t in range(nt)]
    F = BlockDiagOp(ops, engine=eng)
    L = None if "nol" in mode else SpaceTimeDerivative(Nf, nt, engine=eng)
    if "nocpu" in mode:
        xt = torch.rand(F.shape[1], device=eng.device)
    else:
        g = torch.Generator(device="cpu").manual_seed(1234)
        frames = []
        for t in range(nt):
            img = torch.zeros((Nf, Nf))
            img[60 + 2 * t:100 + 2 * t, 40:200] = 1.0
            frames.append(img + 0.05 * torch.rand((Nf, Nf), generator=g))
        xt = torch.cat([f.reshape(-1) for f in frames]).to(eng.device)
    bl = F.apply(xt)
    x0 = torch.zeros(F.shape[1], device=eng.device)
    CGLS(F, bl, x0, 5, 0, history=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    CGLS(F, bl, x0, 100, 0, history=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r = [(t1 - t0) * 1e3, (t2 - t0) * 1e3]
    if L is not None:
        GKS(F, bl, L, 3, 3, 1e-2, history=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        GKS(F, bl, L, 3, 50, 1e-2, history=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        r += [(t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3]
    return r
for i in range(12):
    r = one()
    print(sorted(mode), i, " ".join(f"{v:7.2f}" for v in r), "  <-- stall" if (r[1] > 12 or (len(r) > 2 and r[3] > 25)) else "")
