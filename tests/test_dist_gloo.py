"""world_size = 2 / 4 / 8 runs over gloo (CPU): the sharded dynamic problem of BASELINE config C5 in miniature.

Frames are sharded over ranks (frame-major x, b; block-diagonal forward operator), every inner product is all-reduced
through trips_py_amd.dist.TorchComm (the same class that drives RCCL on GPUs), and the temporal rows of the space-time
regulariser exchange one frame with the time-neighbour (trips_py_amd.operators.spacetime_halo_exchange).  The vector
arithmetic runs on the test-only CPU engine; what is under test is the product's distributed host logic: the solvers'
reduction points, TorchComm, the halo exchange and frame partitioning.  Each rank's slice of the result must equal the
single-process solve."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(nt=4, N=12):
    from oracle import cpu_ref as O
    psfs = [O.gauss_psf((5, 5), (1.0 + 0.25 * t, 1.2))[0] for t in range(nt)]
    rng = np.random.default_rng(3)
    frames = []
    for t in range(nt):
        img = np.zeros((N, N))
        img[2 + t // 2:7 + t // 2, 3:8] = 1.0
        img[8:11, 1 + t:4 + t] = 0.5
        frames.append(img + 0.05 * rng.random((N, N)))
    x_true = np.concatenate([f.reshape(-1) for f in frames])
    F = O.BlockDiag([O.Blur2D(p, N, N) for p in psfs])
    b = F @ x_true
    e = rng.standard_normal(b.shape)
    b = b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e
    return psfs, x_true, b, nt, N


class ShardedSpaceTime:
    """Test adapter: the oracle's space-time operator restricted to this rank's frames, with the product's halo exchange."""

    def __init__(self, N, nt, eng):
        from oracle import cpu_ref as O
        from trips_py_amd.dist import frame_range
        self.N, self.eng = N, eng
        lo, hi = frame_range(nt, eng.world, eng.rank)
        self.ntl = hi - lo
        self.has_next, self.has_prev = eng.rank < eng.world - 1, eng.rank > 0
        self.D2 = O.FirstDerivative2D(N)
        npix, ps = N * N, self.D2.shape[0]
        self.ps, self.npix = ps, npix
        self.ntemp = self.ntl - 1 + (1 if self.has_next else 0)
        self.shape = (self.ntl * ps + self.ntemp * npix, self.ntl * npix)
        self.halo_next = torch.zeros(npix, dtype=torch.float32) if self.has_next else None
        self.halo_prev = torch.zeros(npix, dtype=torch.float32) if self.has_prev else None

    def fwd(self, x):
        from trips_py_amd.operators import spacetime_halo_exchange
        xt = torch.from_numpy(x.astype(np.float32))
        if self.eng.world > 1:
            spacetime_halo_exchange(self.eng, xt, False, self.N, self.ntl, self.halo_next, self.halo_prev)
        X = x.reshape(self.ntl, self.npix)
        spat = np.concatenate([self.D2._fwd(X[t]) for t in range(self.ntl)])
        Xe = np.vstack([X, self.halo_next.numpy().astype(np.float64)[None]]) if self.has_next else X
        return np.concatenate((spat, (Xe[:-1] - Xe[1:]).reshape(-1)))

    def adj(self, y):
        from trips_py_amd.operators import spacetime_halo_exchange
        yt = torch.from_numpy(y.astype(np.float32))
        if self.eng.world > 1:
            spacetime_halo_exchange(self.eng, yt, True, self.N, self.ntl, self.halo_next, self.halo_prev)
        out = np.stack([self.D2._adj(y[t * self.ps:(t + 1) * self.ps]) for t in range(self.ntl)])
        Tm = y[self.ntl * self.ps:].reshape(self.ntemp, self.npix)
        out[:self.ntemp] += Tm
        out[1:] -= Tm[:self.ntl - 1]
        if self.has_prev:
            out[0] -= self.halo_prev.numpy().astype(np.float64)
        return out.reshape(-1)


def _as_fused_operator(fused, plain_op):
    """An OracleOp (so that as_operator() takes it as it is) carrying the fused surface of `fused`."""
    for name in ("streaming", "fused_tv", "sharded", "npix", "tv_weights_len", "halo_frames", "tv_weights", "tv_grad", "N"):
        setattr(plain_op, name, getattr(fused, name))
    return plain_op


class FusedShardedSpaceTime:
    """Test adapter with the product's FUSED surface for a time-sharded space-time regulariser (operators.SpaceTimeDerivative:
    streaming / fused_tv / sharded, halo_frames, tv_weights, tv_grad with `halo=`): the stencil itself is the oracle's operator over
    this rank's frames extended by the neighbour ranks' boundary frames; the exchange is the product's TorchComm.exchange2.  What is
    under test is the solvers' halo bookkeeping (solvers/GKS._HaloTrack: x = V y and v_new from the basis vectors' halos, ONE
    exchange per iteration) — the HIP kernel with halos meets the same solves in tests/test_gpu_dist.py."""
    streaming = True
    fused_tv = True

    def __init__(self, N, nt, eng, plain):
        from oracle import cpu_ref as O
        self.N, self.engine, self.plain = N, eng, plain
        self.npix = N * N
        self.ntl, self.has_next, self.has_prev = plain.ntl, plain.has_next, plain.has_prev
        self.sharded = eng.world > 1
        self.shape = plain.shape
        self.tv_weights_len = self.shape[0] + (self.npix if self.has_prev else 0)
        self.nte = self.ntl + int(self.has_prev) + int(self.has_next)
        self.Le = O.SpaceTimeDerivative(N, self.nte)
        self.ps = plain.ps
        self._xh = torch.zeros(2 * self.npix, dtype=torch.float32)

    # what solvers use of an operator besides the fused forms
    def apply(self, x, out=None, transpose=False, sumsq=None):
        return self._op.apply(x, out=out, transpose=transpose, sumsq=sumsq)

    def halo_frames(self, x, out=None):
        eng, npix = self.engine, self.npix
        out = self._xh if out is None else out
        eng.halo_exchanges += 1
        if eng.world > 1:
            eng.comm.exchange2(x[:npix], out[:npix], x[(self.ntl - 1) * npix:self.ntl * npix], out[npix:2 * npix])
        return out

    def _extended(self, x, halo):
        if self.sharded and halo is None:
            halo = self.halo_frames(x)
        parts = ([halo[:self.npix].numpy().astype(np.float64)] if self.has_prev else []) + [x.numpy().astype(np.float64)] \
            + ([halo[self.npix:].numpy().astype(np.float64)] if self.has_next else [])
        return np.concatenate(parts)

    def _rows(self, w):
        """This rank's weight layout [spatial | own temporal rows | previous rank's boundary row] -> the extended operator's rows
        (the spatial rows of the neighbour frames never reach a local pixel: any weight)."""
        w = w.numpy().astype(np.float64)
        npix, ps, ntl = self.npix, self.ps, self.ntl
        spat, temp = w[:ntl * ps], w[ntl * ps:]
        own = temp[:(ntl - 1 + int(self.has_next)) * npix]
        prev_row = temp[len(own):len(own) + npix] if self.has_prev else np.zeros(0)
        pad = np.zeros(ps)
        return np.concatenate(([pad] if self.has_prev else []) + [spat] + ([pad] if self.has_next else [])
                              + ([prev_row] if self.has_prev else []) + [own])

    def _local(self, full):
        lo = self.npix if self.has_prev else 0
        return full[lo:lo + self.ntl * self.npix]

    def tv_weights(self, x, eps, q, out, halo=None):
        y = self.Le._fwd(self._extended(x, halo))
        w = (y ** 2 + eps ** 2) ** (q / 2 - 1)
        ps, npix, ntl = self.ps, self.npix, self.ntl
        s0 = ps if self.has_prev else 0
        spat = w[s0:s0 + ntl * ps]
        temp = w[self.nte * ps:].reshape(self.nte - 1, npix)
        t0 = 1 if self.has_prev else 0
        own = temp[t0:t0 + ntl - 1 + int(self.has_next)].reshape(-1)
        prev_row = temp[0] if self.has_prev else np.zeros(0)
        out[:self.tv_weights_len].copy_(torch.from_numpy(np.concatenate((spat, own, prev_row)).astype(np.float32)))

    def tv_grad(self, x, w, r_in, lam, out, dot_with=None, dot_out=None, halo=None):
        from cpu_engine import _put
        y = self.Le._fwd(self._extended(x, halo))
        if w is not None:
            y = y * self._rows(w)
        g = self._local(self.Le._adj(y))
        res = (0.0 if r_in is None else r_in.numpy().astype(np.float64)) + lam * g
        out.copy_(torch.from_numpy(res.astype(np.float32)))
        if dot_with is not None:
            _put(dot_out, float(out.double() @ dot_with.double()))


def _solve_all(eng, nt=4, N=12):
    """CGLS, GKS and MMGKS on this rank's shard; returns the local slices of the three solutions + info scalars."""
    from cpu_engine import OracleOp
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.dist import frame_range
    psfs, x_true, b, nt, N = _problem(nt, N)
    lo, hi = frame_range(nt, eng.world, eng.rank)
    npix = N * N
    F = OracleOp(O.BlockDiag([O.Blur2D(psfs[t], N, N) for t in range(lo, hi)]), eng)
    st = ShardedSpaceTime(N, nt, eng)
    out_sides = np.array([float(st.has_prev), float(st.has_next)])

    class _L:                                           # minimal oracle-like wrapper for OracleOp
        shape = st.shape
        _fwd = staticmethod(st.fwd)
        _adj = staticmethod(st.adj)
    L = OracleOp(_L, eng)
    bl, xl = b[lo * npix:hi * npix], x_true[lo * npix:hi * npix]
    out = {}
    calls = getattr(eng.comm, "calls", None)
    c0 = 0 if calls is None else calls["allreduce"]
    x, info = S.CGLS(F, bl, np.zeros(F.shape[1]), 12, 0, x_true=xl)                 # world > 1: ONE all-reduce per iteration
    out["cgls"] = (x.reshape(-1), np.array(info["relResidual"]))
    out["cgls_relerr"] = (np.array(info["relError"]), np.array([0 if calls is None else calls["allreduce"] - c0,
                                                                 info.get("allreduces_per_iteration", -1.0)]))
    c0 = 0 if calls is None else calls["allreduce"]
    x, info = S.CGLS(F, bl, np.zeros(F.shape[1]), 12, 0, x_true=xl, one_reduction=False)   # the two reductions as written
    out["cgls_two"] = (x.reshape(-1), np.array(info["relResidual"]))
    out["cgls_two_relerr"] = (np.array(info["relError"]), np.array([0 if calls is None else calls["allreduce"] - c0, -1.0]))
    x, info = S.CGLS(F, bl, np.zeros(F.shape[1]), 12, 0, x_true=xl, one_reduction=True, history=False)
    out["cgls_one_nohist"] = (x.reshape(-1), np.array(info["relResidual"]))
    x, info = S.GKS(F, bl, L, 3, 6, 1e-2, xl)
    out["gks"] = (x.reshape(-1), np.array(info["Residual"]))
    # the FUSED space-time forms, time-sharded: the kernels' surface of the one-rank solve on every rank, one neighbour exchange per
    # iteration (counted between a 4- and a 6-iteration solve: the start basis costs exchanges of its own)
    Lf = FusedShardedSpaceTime(N, nt, eng, st)
    Lf = _as_fused_operator(Lf, OracleOp(_L, eng))
    cnt = []
    for its in (4, 6):
        h0, c0 = eng.halo_exchanges, 0 if calls is None else calls["allreduce"]
        x, info = S.GKS(F, bl, Lf, 3, its, 1e-2, xl)
        cnt.append((eng.halo_exchanges - h0, (0 if calls is None else calls["allreduce"]) - c0))
    out["gks_fused"] = (x.reshape(-1), np.array(info["Residual"]))
    out["gks_fused_counts"] = (np.array([(cnt[1][0] - cnt[0][0]) / 2.0, (cnt[1][1] - cnt[0][1]) / 2.0, float(info.get("fused_tv", True))]),
                               np.zeros(1))
    x, info = S.MMGKS(F, bl, Lf, 2, 1, 3, 6, 1e-2, xl)
    out["mmgks_fused"] = (x.reshape(-1), np.array(info["Residual"]))
    x, info = S.MMGKS(F, bl, L, 2, 1, 3, 6, 1e-2, xl)
    out["mmgks"] = (x.reshape(-1), np.array(info["Residual"]))
    x, info = S.Hybrid_LSQR(F, bl, 8, 1e-2, xl)
    out["lsqr"] = (x.reshape(-1), np.array(info["regParam_history"], dtype=float))
    out["sides"] = (out_sides, np.zeros(1))
    return out, (lo, hi, npix)


def _worker(rank, world, port, outdir, nt, N):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_engine import CpuEngine
        from trips_py_amd.dist import TorchComm

        class CountingComm(TorchComm):
            def __init__(self):
                super().__init__()
                self.calls = {"allreduce": 0}

            def allreduce_sum_(self, t):
                self.calls["allreduce"] += 1
                return super().allreduce_sum_(t)

        eng = CpuEngine(comm=CountingComm())
        out, (lo, hi, npix) = _solve_all(eng, nt, N)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), lo=lo, hi=hi, npix=npix,
                 **{f"{k}_x": v[0] for k, v in out.items()}, **{f"{k}_s": v[1] for k, v in out.items()})
    finally:
        dist.destroy_process_group()


# (world, frames, N): two ranks (each has ONE neighbour); four ranks x two frames and four ranks x ONE frame (ranks 1, 2 have both
# neighbours; with one frame per rank a rank has no temporal row of its own); C5's real partition, eight ranks x four frames
# (ranks 1..6 interior) at a reduced frame size
@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,nt,N", [(2, 4, 12), (4, 8, 12), (4, 4, 12), (8, 32, 64)])
def test_sharded_solvers_match_single_process(world, nt, N):
    sys.path.insert(0, HERE)
    from cpu_engine import CpuEngine
    ref, _ = _solve_all(CpuEngine(), nt, N)               # single process, all frames
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, nt, N), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    # the two-neighbour branch really ran: every interior rank reports both sides
    sides = [tuple(p["sides_x"]) for p in parts]
    assert sides[0] == (0.0, 1.0) and sides[-1] == (1.0, 0.0) and all(s == (1.0, 1.0) for s in sides[1:-1]), sides
    # CGLS over ranks: one all-reduce per iteration (12 iterations + the one after the solve for the reported norms) against
    # 1 + 2 per iteration for the recurrence as written; iterates of the two forms within 1e-5, identical reported scalars
    for p in parts:
        assert p["cgls_relerr_s"][0] == 12 + 1 and p["cgls_relerr_s"][1] == 1.0, p["cgls_relerr_s"]
        assert p["cgls_two_relerr_s"][0] == 1 + 2 * 12, p["cgls_two_relerr_s"]
        assert np.linalg.norm(p["cgls_x"] - p["cgls_two_x"]) / np.linalg.norm(p["cgls_two_x"]) < 1e-5
        assert np.allclose(p["cgls_s"], p["cgls_two_s"], rtol=1e-5) and np.allclose(p["cgls_relerr_x"], p["cgls_two_relerr_x"], rtol=1e-5)
        assert np.array_equal(p["cgls_one_nohist_x"], p["cgls_x"])
    # fused space-time forms on ranks: ONE neighbour exchange per GKS iteration, no more all-reduces than the plain form, and the
    # result of the single-process fused solve
    for p in parts:
        assert p["gks_fused_counts_x"][0] == 1.0 and p["gks_fused_counts_x"][2] == 1.0, p["gks_fused_counts_x"]
        assert p["gks_fused_counts_x"][1] <= 4.0, p["gks_fused_counts_x"]
    for key in ("cgls", "cgls_two", "gks", "mmgks", "lsqr", "gks_fused", "mmgks_fused"):
        assert [int(p["lo"]) for p in parts] == [r * nt // world for r in range(world)]
        x = np.concatenate([p[f"{key}_x"] for p in parts])
        rx = ref[key][0]
        err = np.linalg.norm(x - rx) / np.linalg.norm(rx)
        assert err < 2e-5, (key, err)
        for p in parts:                                  # global scalars are identical on every rank
            assert np.allclose(p[f"{key}_s"], ref[key][1], rtol=1e-4), key


def test_frame_range_and_comm_validation():
    from trips_py_amd.dist import frame_range
    assert frame_range(32, 8, 3) == (12, 16)
    with pytest.raises(ValueError):
        frame_range(30, 8, 0)
