"""Fan-beam adjoint at 512^2 x 180 x 724 (and 1024^2): the round-6 kernel (k_fan_adj_views) against rounds 3-5's (TRK_FAN_ADJ_MARCH2=1, read
per call): time per apply, and the two results against each other and against the adjoint identity."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import FanBeam2D
for N, views in ((512, 180), (1024, 180), (256, 90)):
    R = FanBeam2D(N, views=views)
    dev = R.engine.device
    g = torch.Generator(device=dev).manual_seed(N)
    x = torch.randn(N * N, device=dev, generator=g)
    y = torch.randn(R.shape[0], device=dev, generator=g)
    z = torch.empty(N * N, device=dev)
    out = {}
    for name, env in (("views", None), ("march2", "1")):
        if env: os.environ["TRK_FAN_ADJ_MARCH2"] = env
        else: os.environ.pop("TRK_FAN_ADJ_MARCH2", None)
        R.apply(y, out=z, transpose=True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): R.apply(y, out=z, transpose=True)
        e1.record(); torch.cuda.synchronize()
        out[name] = (z.clone(), e0.elapsed_time(e1) / 20 * 1e3)
    os.environ.pop("TRK_FAN_ADJ_MARCH2", None)
    Ax = R.apply(x)
    lhs = float(Ax.double() @ y.double())
    d = float((out["views"][0] - out["march2"][0]).double().norm() / out["march2"][0].double().norm())
    ident = [abs(lhs - float(x.double() @ out[k][0].double())) / float(Ax.double().norm() * y.double().norm()) for k in ("views", "march2")]
    print(f"{N}^2 x {views} x {R.n_det}: adjoint {out['views'][1]:7.1f} us (rounds 3-5: {out['march2'][1]:7.1f} us)   |views - march2| / |march2| {d:.2e}   "
          f"adjoint identity {ident[0]:.1e} (march2 {ident[1]:.1e})")
