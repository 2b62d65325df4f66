import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_fullsize import blur64, make_problem
from trips_py_amd.operators import Blur2D
from trips_py_amd.engine import Coef
dev = torch.device("cuda")
for N in (1024, 2048, 4096):
    psf, xt, b = make_problem(N, dev)
    psf_t = torch.from_numpy(psf).to(dev, torch.float64)
    A = Blur2D(psf, N, N); eng = A.engine
    x32 = xt.float()
    ref = blur64(x32.double().reshape(N, N), psf_t).reshape(-1)
    S = eng.scalars(8)
    y = A.apply(x32)
    y2 = A.apply(x32, sumsq=S[0:1])
    yt = A.apply(x32, transpose=True, sumsq=S[1:2])
    rel = lambda a, r: float(torch.linalg.norm(a.double() - r) / torch.linalg.norm(r))
    print(N, "fwd", rel(y, ref), "fwd+ss", rel(y2, ref), "adj", rel(yt, blur64(x32.double().reshape(N, N), psf_t, True).reshape(-1)),
          "ss", float(S[0]) / float((ref ** 2).sum()) - 1, float(S[1]) / float((ref**2).sum()) - 1)
    # vector kernels at this size
    n = N * N
    p = torch.randn(n, device=dev); r = torch.randn(n, device=dev); w = torch.randn(n, device=dev)
    xn = eng.empty(n); S[2] = 3.0; S[3] = 7.0
    r0 = r.clone()
    eng.cgls_update(S[2:3], S[3:4], x32, p, xn, r, w, x32, S[4:7])
    step = 3.0 / 7.0
    print("   upd x", rel(xn, (x32.double() + step * p.double())), "r", rel(r, r0.double() - step * w.double()),
          "sums", float(S[4]) / float((xn.double() ** 2).sum()) - 1, float(S[5]) / float(((step * p.double()) ** 2).sum()) - 1,
          float(S[6]) / float(((xn.double() - x32.double()) ** 2).sum()) - 1)
    p0 = p.clone()
    eng.axpby(1.0, w, Coef(1.0, num=S[2:3], den=S[3:4]), p, p)
    print("   axpby", rel(p, w.double() + step * p0.double()))
    eng.nrm2sq(w, S[7:8]); print("   nrm", float(S[7]) / float((w.double() ** 2).sum()) - 1)
