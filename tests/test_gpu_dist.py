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


def test_libtrk_rccl_entry_points_single_rank():
    """trk_comm_unique_id / trk_comm_init / trk_allreduce_f64 / trk_halo_exchange (include/trk.h) on a one-rank RCCL
    communicator: the calls really go through RCCL (a one-GPU box cannot host two ranks of it)."""
    import torch
    from trips_py_amd.dist import RcclComm
    torch.cuda.set_device(0)
    c = RcclComm(0, 1)
    t = torch.tensor([1.5, -2.25, 1e300], dtype=torch.float64, device="cuda")
    c.allreduce_sum_(t)
    a = torch.arange(4099, dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    c.shift(a, 0, b, 0)                       # send to / receive from itself in one group
    torch.cuda.synchronize()
    assert t.tolist() == [1.5, -2.25, 1e300] and torch.equal(a, b)
    c.shift(a, 5, b, None)                    # neighbours outside the communicator: nothing happens
    del c
