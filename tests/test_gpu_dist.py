"""The sharded (time-frames over ranks) path with the REAL HIP kernels, world_size = 2, on one GPU: both ranks use cuda:0
and talk over gloo with host staging (RCCL refuses two ranks on one device).  Exercises HipEngine + TorchComm, the
per-iteration all-reduces of the solvers and the space-time halo exchange feeding trk_spacetime_set_halo.
Each rank's slice must equal the single-process solve of the whole problem."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _solve(eng, nt=4, N=32):
    from trips_py_amd import solvers as S
    from trips_py_amd.dist import frame_range
    from trips_py_amd.operators import Blur2D, BlockDiagOp, Radon2DParallel, SpaceTimeDerivative
    from trips_py_amd.problems import gauss_psf
    rng = np.random.default_rng(5)
    frames = []
    for t in range(nt):
        img = np.zeros((N, N))
        img[4 + t:14 + t, 6:20] = 1.0
        img[20:28, 3 + 2 * t:12 + 2 * t] = 0.5
        frames.append(img + 0.05 * rng.random((N, N)))
    lo, hi = frame_range(nt, eng.world, eng.rank)
    npix = N * N
    out = {}
    # (a) dynamic blur  (b) dynamic parallel-beam tomography, both with the space-time regulariser
    for tag in ("blur", "tomo"):
        if tag == "blur":
            ops = [Blur2D(gauss_psf((5, 5), (1.0 + 0.2 * t, 1.3))[0], N, N, engine=eng) for t in range(lo, hi)]
        else:
            ops = [Radon2DParallel(N, np.deg2rad(t + 12.0 * np.arange(15)), engine=eng) for t in range(lo, hi)]
        F = BlockDiagOp(ops, engine=eng)
        L = SpaceTimeDerivative(N, nt, engine=eng)
        xl = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames[lo:hi]]).astype(np.float32)).to(eng.device)
        bl = F.apply(xl)
        x, info = S.CGLS(F, bl, torch.zeros(F.shape[1], device=eng.device), 10, 0)
        out[f"{tag}_cgls"] = (x.reshape(-1).cpu().numpy(), np.array(info["relResidual"]))
        x, info = S.GKS(F, bl, L, 3, 5, 1e-2)
        out[f"{tag}_gks"] = (x.reshape(-1).cpu().numpy(), np.array(info["Residual"]))
        x, info = S.MMGKS(F, bl, L, 2, 1, 3, 5, 1e-2)
        out[f"{tag}_mmgks"] = (x.reshape(-1).cpu().numpy(), np.array(info["Residual"]))
        # Golub-Kahan with the half steps inside the projector's output pass (tomo: trk_op_apply_axpby, norms all-reduced
        # between the two applies of a step) / by apply + axpby (blur)
        x, info = S.Hybrid_LSQR(F, bl, 8, 1e-2)
        out[f"{tag}_lsqr"] = (x.reshape(-1).cpu().numpy(), np.array(info["regParam_history"], dtype=np.float64))
    return out


def _worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from trips_py_amd.dist import TorchComm
        from trips_py_amd.engine import HipEngine
        out = _solve(HipEngine(comm=TorchComm()))
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{f"{k}_x": v[0] for k, v in out.items()},
                 **{f"{k}_s": v[1] for k, v in out.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_hip_path_matches_single_process():
    from trips_py_amd.engine import HipEngine
    ref = _solve(HipEngine())
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    for key in ref:
        x = np.concatenate([p[f"{key}_x"] for p in parts])
        err = np.linalg.norm(x - ref[key][0]) / np.linalg.norm(ref[key][0])
        assert err < 2e-5, (key, err)
        for p in parts:
            assert np.allclose(p[f"{key}_s"], ref[key][1], rtol=1e-4), key


_RCCL_ONE_RANK = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
torch.cuda.set_device(0)
from trips_py_amd.dist import RcclComm
from trips_py_amd.engine import HipEngine
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
from trips_py_amd import solvers as S
c = RcclComm(0, 1)
t = torch.tensor([1.5, -2.25, 1e300], dtype=torch.float64, device="cuda")
c.allreduce_sum_(t)                        # TRK_COMM_FORCE=1: through ncclAllReduce although world = 1
a = torch.arange(4099, dtype=torch.float32, device="cuda")
b = torch.zeros_like(a)
c.shift(a, 0, b, 0)                       # send to / receive from itself in one group
torch.cuda.synchronize()
assert t.tolist() == [1.5, -2.25, 1e300] and torch.equal(a, b)
c.shift(a, 5, b, None)                    # neighbours outside the communicator: nothing happens
# the one-all-reduce CGLS with its exchanges enqueued from C through this communicator (trk_cgls_iterate_sharded)
eng = HipEngine(comm=c)
eng.world = 2                              # pretend: makes the solver take its sharded branch; the communicator has one rank
N, nt = 64, 4
F = BlockDiagOp([Radon2DParallel(N, np.deg2rad(t + 12.0 * np.arange(15)), engine=eng) for t in range(nt)], engine=eng)
g = torch.Generator(device="cuda").manual_seed(0)
xt = torch.rand(F.shape[1], device="cuda", generator=g)
bl = F.apply(xt)
bl = bl + 0.01 * torch.randn(bl.numel(), device="cuda", generator=g) * bl.norm() / bl.numel() ** 0.5     # one percent of noise: no stagnation at the fp32 floor
x1, i1 = S.CGLS(F, bl, torch.zeros(F.shape[1], device="cuda"), 30, 0, xt)
assert i1["allreduces_per_iteration"] == 1.0
eng.world = 1
x2, i2 = S.CGLS(F, bl, torch.zeros(F.shape[1], device="cuda"), 30, 0, xt)
assert "allreduces_per_iteration" not in i2
err = float(torch.linalg.norm(x1 - x2) / torch.linalg.norm(x2))
assert err < 1e-5, err
de = float(np.max(np.abs(np.array(i1["relError"]) / np.array(i2["relError"]) - 1)))
dr = float(np.max(np.abs(np.array(i1["relResidual"]) / np.array(i2["relResidual"]) - 1)))
print("x", err, "relError", de, "relResidual", dr)
# the reported scalars are functions of iterates that agree to < 1e-5: ||x - x_true|| / ||x|| (about 0.1 here) moves by up to
# err / 0.1, and the step norm ||x_k - x_{k-1}|| / ||x_k|| is the most sensitive of the three
assert de < 2e-4 and dr < 2e-3, (de, dr)
print("rccl one-rank ok", err)
"""


def test_libtrk_rccl_entry_points_single_rank():
    """trk_comm_unique_id / trk_comm_init / trk_allreduce_f64 / trk_halo_exchange (include/trk.h) on a one-rank RCCL
    communicator, and trk_cgls_iterate_sharded enqueueing its all-reduces through it: with TRK_COMM_FORCE=1 the calls really go
    through RCCL (a one-GPU box cannot host two ranks of it).  Own process: the switch is read once per process."""
    import subprocess
    env = dict(os.environ, TRK_COMM_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK % REPO], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("case", ["tomo_dynamic", "blur96", "tomo_static"])
def test_one_reduction_cgls_equals_the_recurrence_as_written(case):
    """CGLS(one_reduction=True) on one rank = the merged-reduction arrangement without its exchange (C loop,
    trk_cgls_iterate_sharded with comm = NULL, and stepped from Python) against the two-reduction forms: iterates and reported
    scalars within 1e-5 over 40 iterations; history modes untouched."""
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, Radon2DParallel
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers.CGLS import CGLSRunSharded
    if case == "tomo_dynamic":
        A = BlockDiagOp([Radon2DParallel(64, np.deg2rad(t + 12.0 * np.arange(15))) for t in range(4)])
    elif case == "blur96":
        A = Blur2D(gauss_psf((9, 9), (3, 3))[0], 96, 96)
    else:
        A = Radon2DParallel(128, np.linspace(0, np.pi, 45, endpoint=False))
    dev = A.engine.device
    g = torch.Generator(device=dev).manual_seed(1)
    xt = torch.rand(A.shape[1], device=dev, generator=g)
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(b.numel(), device=dev, generator=g) * b.norm() / b.numel() ** 0.5
    x0 = torch.zeros(A.shape[1], device=dev)
    its = 40
    xa, ia = S.CGLS(A, b, x0, its, 0, xt, one_reduction=False, tiled=False, fused=False)
    xb, ib = S.CGLS(A, b, x0, its, 0, xt, one_reduction=True)
    assert ib["allreduces_per_iteration"] == 0.0 and ib["its"] == ia["its"] == its
    for k in range(its):
        ra, rb = ia["xHistory"][k].reshape(-1), ib["xHistory"][k].reshape(-1)
        assert float(torch.linalg.norm(ra - rb) / torch.linalg.norm(ra)) < 1e-5, k
    de = float(np.max(np.abs(np.array(ia["relError"]) / np.array(ib["relError"]) - 1)))
    dr = float(np.max(np.abs(np.array(ia["relResidual"]) / np.array(ib["relResidual"]) - 1)))
    assert de < 2e-4 and dr < 2e-3, (de, dr)
    # stepped from Python = the C loop, bit for bit
    run = CGLSRunSharded(A, b, x0, its, xt)
    for _ in range(its):
        run.step()
    assert torch.equal(run.x_cur, xb.reshape(-1))
    _g0, rows = run.rows()
    assert np.allclose(np.sqrt(rows[:, 4]) / np.sqrt(rows[:, 2]), ib["relError"], rtol=1e-12)
    xc, ic = S.CGLS(A, b, x0, its, 0, one_reduction=True, history=7)
    assert torch.equal(xc, xb) and ic["xHistory"].iterations == [6, 13, 20, 27, 34, 39]
