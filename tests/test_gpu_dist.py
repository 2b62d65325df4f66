"""The sharded (time-frames over ranks) path with the REAL HIP kernels, world_size = 2 and 4, on one GPU: all ranks use cuda:0
and talk over gloo with host staging (RCCL refuses two ranks on one device).  Exercises HipEngine + TorchComm, the
per-iteration all-reduces of the solvers, the two-sided halo exchange feeding the fused space-time stencil (trk_tv_halo:
GKS / MMGKS) and the one-frame shifts feeding trk_spacetime_set_halo (plain L / L^T applies).
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

from conftest import relerr

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
    # (a) dynamic blur  (b) dynamic parallel-beam tomography  (c) the frames cut out of ONE sparse forward matrix (io.py:223-225; every
    # rank keeps the blocks of its own frames: SparseBlockDiag.from_matrix), all with the space-time regulariser
    for tag in ("blur", "tomo", "sparse"):
        if tag == "blur":
            ops = [Blur2D(gauss_psf((5, 5), (1.0 + 0.2 * t, 1.3))[0], N, N, engine=eng) for t in range(lo, hi)]
        elif tag == "tomo":
            ops = [Radon2DParallel(N, np.deg2rad(t + 12.0 * np.arange(15)), engine=eng) for t in range(lo, hi)]
        if tag == "sparse":
            import scipy.sparse as sp
            from trips_py_amd.operators import SparseBlockDiag
            rpf = 3 * N
            big = sp.random(nt * rpf, nt * npix, density=0.02, random_state=11, format="csr")     # the same matrix on every rank
            big.data = np.round(big.data * 32) / 32 + 1 / 32
            F = SparseBlockDiag.from_matrix(big, nt, rpf, npix, engine=eng)
            assert F.shape == ((hi - lo) * rpf, (hi - lo) * npix)
        else:
            F = BlockDiagOp(ops, engine=eng)
        L = SpaceTimeDerivative(N, nt, engine=eng)
        xl = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames[lo:hi]]).astype(np.float32)).to(eng.device)
        bl = F.apply(xl)
        x, info = S.CGLS(F, bl, torch.zeros(F.shape[1], device=eng.device), 10, 0)
        out[f"{tag}_cgls"] = (x.reshape(-1).cpu().numpy(), np.array(info["relResidual"]))
        # GKS on ranks runs the kernels of the one-rank solve (fused space-time stencil with the neighbours' boundary frames,
        # trk_tv_halo): ONE halo exchange and at most 3 all-reduces per iteration — counted between a 3- and a 5-iteration solve
        assert L.fused_tv and L.streaming
        cnt = []
        for its in (3, 5):
            h0, r0 = eng.halo_exchanges, eng.reduction_points
            x, info = S.GKS(F, bl, L, 3, its, 1e-2)
            cnt.append((eng.halo_exchanges - h0, eng.reduction_points - r0))
        assert info.get("fused_tv", eng.world == 1)
        out[f"{tag}_gks"] = (x.reshape(-1).cpu().numpy(), np.array(info["Residual"]))
        out[f"{tag}_gks_counts"] = (np.array([(cnt[1][0] - cnt[0][0]) / 2.0, (cnt[1][1] - cnt[0][1]) / 2.0]), np.zeros(1))
        if tag == "tomo":        # the reference's default regparam on ranks: every rank selects the same lambda from the all-reduced Gram data
            x, info = S.GKS(F, bl, L, 3, 6, "gcv")
            out["tomo_gksgcv"] = (x.reshape(-1).cpu().numpy(), np.array(info["regParam_history"], dtype=np.float64))
        x, info = S.MMGKS(F, bl, L, 2, 1, 3, 5, 1e-2)
        out[f"{tag}_mmgks"] = (x.reshape(-1).cpu().numpy(), np.array(info["Residual"]))
        # Golub-Kahan with the half steps inside the projector's output pass (tomo: trk_op_apply_axpby, norms all-reduced
        # between the two applies of a step) / by apply + axpby (blur)
        x, info = S.Hybrid_LSQR(F, bl, 8, 1e-2)
        out[f"{tag}_lsqr"] = (x.reshape(-1).cpu().numpy(), np.array(info["regParam_history"], dtype=np.float64))
    return out


def _worker(rank, world, port, outdir, nt, N):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from trips_py_amd.dist import TorchComm
        from trips_py_amd.engine import HipEngine
        out = _solve(HipEngine(comm=TorchComm()), nt, N)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{f"{k}_x": v[0] for k, v in out.items()},
                 **{f"{k}_s": v[1] for k, v in out.items()})
    finally:
        dist.destroy_process_group()


# world 4: ranks 1 and 2 have BOTH time-neighbours (trk_tv_halo with two halo frames, exchange2's two-sided branch, the
# has_prev & has_next form of SpaceTimeDerivative) — with two frames per rank and with ONE frame per rank (no temporal row inside a rank)
@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,nt,N", [(2, 4, 32), (4, 8, 32), (4, 4, 32)])
def test_sharded_hip_path_matches_single_process(world, nt, N):
    from trips_py_amd.engine import HipEngine
    ref = _solve(HipEngine(), nt, N)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, nt, N), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    for tag in ("blur", "tomo", "sparse"):
        for p in parts:
            per_it = p[f"{tag}_gks_counts_x"]
            assert per_it[0] == 1.0 and per_it[1] <= 3.0, (tag, per_it)
    for key in ref:
        if key.endswith("_counts"):
            continue
        x = np.concatenate([p[f"{key}_x"] for p in parts])
        err = np.linalg.norm(x - ref[key][0]) / np.linalg.norm(ref[key][0])
        auto = key.endswith("gcv")               # an automatic lambda amplifies the 1e-7 differences of the summation orders
        assert err < (2e-3 if auto else 2e-5), (key, err)
        for p in parts:
            assert np.allclose(p[f"{key}_s"], ref[key][1], rtol=5e-2 if auto else 1e-4), key
        if auto:                                 # ... but every rank must have selected the SAME lambdas
            for p in parts[1:]:
                assert np.array_equal(p[f"{key}_s"], parts[0][f"{key}_s"]), key


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
# 15 iterations: past ~25 fp32 CGLS on this under-determined noisy problem leaves its float64 run by 3e-3 in EITHER arrangement
# (tools/cgls_forms_accuracy.py), and two fp32 runs then only agree to that
x1, i1 = S.CGLS(F, bl, torch.zeros(F.shape[1], device="cuda"), 15, 0, xt)
assert i1["allreduces_per_iteration"] == 1.0
eng.world = 1
x2, i2 = S.CGLS(F, bl, torch.zeros(F.shape[1], device="cuda"), 15, 0, xt)
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
    trk_cgls_iterate_sharded with comm = NULL, and stepped from Python) against the two-reduction form AND the float64 oracle.
    The yardstick is the oracle: on noisy ill-posed problems fp32 CGLS in EITHER arrangement leaves the float64 run of the same
    recurrence once it is past semi-convergence (tools/cgls_forms_accuracy.py on the MI355X, 4 x 64^2 dynamic problem: both
    3e-6 / 2e-5 at iterate 20, both 3.5e-3 at iterate 30; C5 size: both 1e-6 at 20, both 1.8e-3 at 30), so the merged form is held
    (a) to 1e-5 of the two-reduction form wherever THAT is within 1e-6 of float64, (b) everywhere to ten times the distance from
    float64 the two-reduction form has reached within the next two iterations (+ 1e-5): the merged form meets the onset of the fp32
    breakdown an iteration or two earlier and is the same from there on (profiles/r03/cgls_forms_table.txt)."""
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, BlockDiagOp, Radon2DParallel
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers.CGLS import CGLSRunSharded
    if case == "tomo_dynamic":
        angs = [np.deg2rad(t + 12.0 * np.arange(15)) for t in range(4)]
        A, Ao = BlockDiagOp([Radon2DParallel(64, a) for a in angs]), O.BlockDiag([O.Radon2D(64, a) for a in angs])
    elif case == "blur96":
        psf = gauss_psf((9, 9), (3, 3))[0]
        A, Ao = Blur2D(psf, 96, 96), O.Blur2D(psf, 96, 96)
    else:
        ang = np.linspace(0, np.pi, 45, endpoint=False)
        A, Ao = Radon2DParallel(128, ang), O.Radon2D(128, ang)
    rng = np.random.default_rng(1)
    xt = rng.random(A.shape[1])
    b = Ao @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x0 = np.zeros(A.shape[1])
    its = 40
    xo, io = O.cgls(Ao, b.reshape(-1, 1), x0.reshape(-1, 1), its, 0, xt.reshape(-1, 1))
    xa, ia = S.CGLS(A, b, x0, its, 0, xt, one_reduction=False, tiled=False, fused=False)
    xb, ib = S.CGLS(A, b, x0, its, 0, xt, one_reduction=True)
    assert ib["allreduces_per_iteration"] == 0.0 and ib["its"] == ia["its"] == its
    d2 = [relerr(ia["xHistory"][k], io["xHistory"][k]) for k in range(its)]           # two reductions vs float64
    d1 = [relerr(ib["xHistory"][k], io["xHistory"][k]) for k in range(its)]           # one reduction vs float64
    d12 = [relerr(ib["xHistory"][k], ia["xHistory"][k]) for k in range(its)]
    resolved = [k for k in range(its) if d2[k] < 1e-6]
    assert len(resolved) >= 3 and all(d12[k] < 1e-5 for k in resolved), (d2, d12)
    # the merged form may meet the breakdown up to two iterations earlier (dynamic problem, iterate 21: 3.5e-4 against 2.7e-5,
    # iterate 22: 4.7e-3 against 1.5e-3, iterate 23: 4.5e-3 both), never more than that and never further out
    for k in range(its):
        assert d1[k] < 10 * max(d2[k:k + 3]) + 1e-5, (k, d2[k:k + 3], d1[k])
    # stepped from Python = the C loop, bit for bit
    dev = A.engine.device
    bt, xtt = torch.from_numpy(b.astype(np.float32)).to(dev), torch.from_numpy(xt.astype(np.float32)).to(dev)
    xc, ic = S.CGLS(A, bt, torch.zeros(A.shape[1], device=dev), 15, 0, xtt, one_reduction=True)
    run = CGLSRunSharded(A, bt, torch.zeros(A.shape[1], device=dev), 15, xtt)
    for _ in range(15):
        run.step()
    assert torch.equal(run.x_cur, xc.reshape(-1))
    _g0, rows = run.rows()
    assert np.allclose(np.sqrt(rows[:, 4]) / np.sqrt(rows[:, 2]), ic["relError"], rtol=1e-12)
    xd, idd = S.CGLS(A, bt, torch.zeros(A.shape[1], device=dev), 15, 0, one_reduction=True, history=7)
    assert torch.equal(xd, xc) and idd["xHistory"].iterations == [6, 13, 14]
