"""GPU parity of the regulariser / block-diagonal / Radon operators with the oracle (and the reference's dense matrices
for the derivative operators, tests/golden/deriv_ops.npz)."""
import numpy as np
import pytest
import torch

from conftest import bar, load_golden, relerr

pytestmark = pytest.mark.gpu


def test_derivative_operators_equal_reference_matrices():
    from trips_py_amd.operators import FirstDerivative2D, SpaceTimeDerivative
    g = load_golden("deriv_ops")
    for n in (4, 5):
        L = FirstDerivative2D(n)
        assert np.array_equal(L.todense(), g[f"D2_{n}"])                    # entries are 0, +1, -1: exact in fp32
        assert np.array_equal(L.T @ np.eye(L.shape[0]), g[f"D2_{n}"].T)
    for (N, nt) in ((4, 3), (3, 2)):
        L = SpaceTimeDerivative(N, nt)
        assert np.array_equal(L.todense(), g[f"Dst_{N}_{nt}"])
        assert np.array_equal(L.T @ np.eye(L.shape[0]), g[f"Dst_{N}_{nt}"].T)


@pytest.mark.parametrize("N", [7, 64, 257, 1024])
def test_first_derivative_2d_vs_oracle(N):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FirstDerivative2D
    rng = np.random.default_rng(N)
    L, Lo = FirstDerivative2D(N), O.FirstDerivative2D(N)
    x, y = rng.standard_normal(N * N), rng.standard_normal(Lo.shape[0])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    assert relerr(L @ x, Lo @ f(x)) < 1e-6 and relerr(L.T @ y, Lo.T @ f(y)) < 1e-6
    eng = L.engine
    S = eng.scalars(2)
    xd = torch.from_numpy(x.astype(np.float32)).to(eng.device)
    out = L.apply(xd, sumsq=S.ref(0))
    back = L.apply(out, transpose=True, sumsq=S.ref(1))
    s = S.host()
    assert np.isclose(s[0], float((out.double() ** 2).sum()), rtol=1e-12)
    assert np.isclose(s[1], float((back.double() ** 2).sum()), rtol=1e-12)
    X3 = rng.standard_normal((N * N, 3))                                     # (n,k) operand, GKS.py:38 `L@V`
    assert relerr(L @ X3, np.stack([Lo @ f(X3[:, j]) for j in range(3)], 1)) < 1e-6


def test_first_derivative_2d_large_image_grid_stride_path():
    """N = 4500: more row batches than the capped grid of the reducing kernels holds (the batches are then grid-strided):
    L x, L^T y and their fused norms against the same differences formed with torch."""
    from trips_py_amd.operators import FirstDerivative2D
    N = 4500
    L = FirstDerivative2D(N)
    eng = L.engine
    g = torch.Generator(device=eng.device).manual_seed(1)
    X = torch.randn(N, N, device=eng.device, generator=g)
    S = eng.scalars(2)
    out = L.apply(X.reshape(-1), sumsq=S.ref(0))
    H, V = X[:, :-1] - X[:, 1:], X[:-1, :] - X[1:, :]
    want = torch.cat([H.reshape(-1), V.reshape(-1)])
    assert torch.equal(out, want)
    assert torch.equal(L.apply(X.reshape(-1)), want)                          # the uncapped grid of the plain apply
    back = L.apply(out, transpose=True, sumsq=S.ref(1))
    Z = torch.zeros(N, N, device=eng.device)
    Z[:, :-1] += H
    Z[:, 1:] -= H
    Z[:-1, :] += V
    Z[1:, :] -= V
    assert float(torch.linalg.norm(back - Z.reshape(-1)) / torch.linalg.norm(Z)) < 1e-6
    s = S.host()
    assert np.isclose(s[0], float((out.double() ** 2).sum()), rtol=1e-12)
    assert np.isclose(s[1], float((back.double() ** 2).sum()), rtol=1e-12)


@pytest.mark.parametrize("N,nt", [(16, 3), (33, 5), (256, 4)])
def test_spacetime_vs_oracle(N, nt):
    from oracle import cpu_ref as O
    from trips_py_amd.operators import SpaceTimeDerivative
    rng = np.random.default_rng(N + nt)
    L, Lo = SpaceTimeDerivative(N, nt), O.SpaceTimeDerivative(N, nt)
    assert L.shape == Lo.shape
    x, y = rng.standard_normal(Lo.shape[1]), rng.standard_normal(Lo.shape[0])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    assert relerr(L @ x, Lo @ f(x)) < 1e-6 and relerr(L.T @ y, Lo.T @ f(y)) < 1e-6
    S = L.engine.scalars(1)
    out = L.apply(torch.from_numpy(x.astype(np.float32)).to(L.engine.device), sumsq=S.ref(0))
    assert np.isclose(S.host()[0], float((out.double() ** 2).sum()), rtol=1e-12)
    back = L.apply(out, transpose=True, sumsq=S.ref(0))
    assert np.isclose(S.host()[0], float((back.double() ** 2).sum()), rtol=1e-12)


def test_blockdiag_frames():
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Blur2D, BlockDiagOp
    rng = np.random.default_rng(2)
    N, nt = 48, 4
    psfs = [O.gauss_psf((5, 5), (1.0 + 0.3 * t, 1.5))[0] for t in range(nt)]
    F = BlockDiagOp([Blur2D(p, N, N) for p in psfs])
    Fo = O.BlockDiag([O.Blur2D(p, N, N) for p in psfs])
    x = rng.standard_normal(nt * N * N)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    assert F.shape == Fo.shape
    assert relerr(F @ x, Fo @ f(x)) < 1e-5 and relerr(F.T @ x, Fo.T @ f(x)) < 1e-5
    S = F.engine.scalars(1)
    out = F.apply(torch.from_numpy(x.astype(np.float32)).to(F.engine.device), sumsq=S.ref(0))
    assert np.isclose(S.host()[0], float((out.double() ** 2).sum()), rtol=1e-8)


# ----------------------------------------------------------------------- Radon (convention pinned via the fan-beam images; weights: Joseph)
@pytest.mark.parametrize("N,na,nd", [(32, 12, 32), (64, 45, 64), (64, 45, 40), (96, 30, 140), (256, 180, 256), (257, 33, 301)])
def test_radon_vs_oracle_convention(N, na, nd):
    """The HIP projector against the oracle's sparse-matrix Joseph projector on the same inputs.  The oracle itself is
    NOT pinned to ASTRA (absent, un-pinned in the reference) — this checks the two implementations of the recorded
    convention against each other.  Bar: 1e-5 (north_star); the fixed-point ray coordinate keeps every interpolation
    weight within 2^-24 of its float64 value."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    rng = np.random.default_rng(N)
    ang = np.linspace(0, np.pi, na, endpoint=False)
    R, Ro = Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)
    assert R.shape == Ro.shape
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    img = np.exp(-((ii - N / 2.5) ** 2 + (jj - N / 1.7) ** 2) / (0.02 * N * N)) + 0.3 * (np.abs(ii - N / 2) < N / 5) * (np.abs(jj - N / 3) < N / 6)
    x = (img + 0.05 * rng.random((N, N))).reshape(-1)
    y = rng.standard_normal(Ro.shape[0])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    assert relerr(R @ x, Ro @ f(x)) < 1e-5, relerr(R @ x, Ro @ f(x))
    assert relerr(R.T @ y, Ro.T @ f(y)) < 1e-5, relerr(R.T @ y, Ro.T @ f(y))


@pytest.mark.parametrize("case", ["fan7", "scattered", "wide_detector", "narrow_detector", "near45", "quads", "signs", "twice", "deg180"])
def test_radon_shared_window_forward_vs_oracle(case):
    """N >= 1024 runs the window-sharing forward kernel (4 neighbouring angles stage one LDS window; rays are owned by
    their column at the band's top row).  Against the oracle's sparse Joseph matrix, including what the kernel special-
    cases: a last group of fewer than 4 angles, groups mixing row- and column-driven angles, angles in scattered order
    (the union window does not fit -> direct gathers), detectors wider / narrower than the image."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    N = 1024
    ang = {"fan7": np.linspace(0, np.pi, 7, endpoint=False),
           "scattered": np.array([0.1, 2.0, 0.8, 2.9, 1.3, 0.05]),
           "wide_detector": np.deg2rad([10.0, 11.0, 12.0, 13.0, 100.0, 101.0]),
           "narrow_detector": np.deg2rad([30.0, 31.0, 32.0, 33.0, 34.0]),
           "near45": np.deg2rad([43.0, 44.0, 45.0, 46.0, 133.0, 134.0, 135.0, 136.0]),
           # round 4 (k_radon_fwd_quad: the four images of a base angle under the grid's symmetries share a wave): complete quads in
           # caller order, every sign case of (cos, sin) incl. angles outside [0, pi), an angle given twice, the full 1-degree set
           "quads": np.deg2rad([10.0, 80.0, 100.0, 170.0, 11.0, 79.0, 101.0, 169.0, 12.0, 78.0, 13.0]),
           "signs": np.deg2rad([20.0, 160.0, -20.0, 200.0, 70.0, 110.0, -70.0, 250.0, 290.0, 340.0, -160.0, 21.0]),
           "twice": np.deg2rad([30.0, 30.0, 150.0, 60.0, 30.0, 120.0]),
           "deg180": np.linspace(0, np.pi, 180, endpoint=False)}[case]
    nd = {"wide_detector": 1500, "narrow_detector": 700}.get(case, N)
    R, Ro = Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)
    rng = np.random.default_rng(11)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    x = (np.exp(-((ii - N / 2.5) ** 2 + (jj - N / 1.7) ** 2) / (0.02 * N * N)) + 0.05 * rng.random((N, N))).reshape(-1)
    f = lambda v: v.astype(np.float32).astype(np.float64)
    assert relerr(R @ x, Ro @ f(x)) < 1e-5, relerr(R @ x, Ro @ f(x))
    yn = rng.standard_normal(Ro.shape[0])
    assert relerr(R.T @ yn, Ro.T @ f(yn)) < 1e-5, relerr(R.T @ yn, Ro.T @ f(yn))
    # adjoint identity with the gather adjoint
    eng = R.engine
    xd = torch.from_numpy(rng.standard_normal(N * N).astype(np.float32)).to(eng.device)
    yd = torch.from_numpy(rng.standard_normal(R.shape[0]).astype(np.float32)).to(eng.device)
    Rx, RTy = R.apply(xd), R.apply(yd, transpose=True)
    S = eng.scalars(2)
    eng.dot(Rx, yd, S.ref(0))
    eng.dot(xd, RTy, S.ref(1))
    dd = S.host()
    assert abs(dd[0] - dd[1]) <= 1e-6 * float(torch.linalg.norm(Rx.double()) * torch.linalg.norm(yd.double()))


@pytest.mark.parametrize("case", ["deg180", "quads", "signs", "scattered", "ragged1028", "wide_detector", "dynamic"])
def test_radon_lean_quad_kernel_equals_the_round4_kernel_bit_for_bit(case, monkeypatch):
    """Round 6: k_radon_fwd_quadf (the march alone, its bookkeeping from a per-operator plan: k_radon_quad_plan) serves the workgroups
    the plan allows, k_radon_fwd_quad the listed rest.  Both own the same rays and add the same products in the same order, so the
    sinogram must not change by a bit against the all-k_radon_fwd_quad run (TRK_RADON_NO_QUADF=1) — on full 1-degree sets (every
    workgroup lean but ragged ones), incomplete quads, scattered angles (windows that do not fit: listed), an image whose bands are
    ragged, a detector wider than the image and a dynamic operator (frames of few angles)."""
    from trips_py_amd.operators import Radon2DParallel
    N = 1028 if case == "ragged1028" else 1024
    ang = {"deg180": np.linspace(0, np.pi, 180, endpoint=False),
           "quads": np.deg2rad([10.0, 80.0, 100.0, 170.0, 11.0, 79.0, 101.0, 169.0, 12.0, 78.0, 13.0]),
           "signs": np.deg2rad([20.0, 160.0, -20.0, 200.0, 70.0, 110.0, -70.0, 250.0, 290.0, 340.0, -160.0, 21.0]),
           "scattered": np.array([0.1, 2.0, 0.8, 2.9, 1.3, 0.05]),
           "ragged1028": np.linspace(0, np.pi, 60, endpoint=False),
           "wide_detector": np.linspace(0, np.pi, 36, endpoint=False),
           "dynamic": np.deg2rad(np.arange(2)[:, None] * 7.0 + 12.0 * np.arange(15)[None, :])}[case]
    nd = 1500 if case == "wide_detector" else N
    if case == "dynamic":                            # two frames of 15 angles each: ONE handle, one launch (BlockDiagOp)
        from trips_py_amd.operators import BlockDiagOp
        R = BlockDiagOp([Radon2DParallel(N, a, n_det=nd) for a in ang])
    else:
        R = Radon2DParallel(N, ang, n_det=nd)
    eng = R.engine
    g = torch.Generator(device=eng.device).manual_seed(5)
    x = torch.rand(R.shape[1], device=eng.device, generator=g)
    y_lean = R.apply(x).clone()
    monkeypatch.setenv("TRK_RADON_NO_QUADF", "1")
    y_old = R.apply(x).clone()
    monkeypatch.delenv("TRK_RADON_NO_QUADF")
    y_again = R.apply(x)
    assert torch.equal(y_lean, y_old), float((y_lean - y_old).abs().max())
    assert torch.equal(y_lean, y_again)
    assert float(y_lean.abs().max()) > 0.0


@pytest.mark.parametrize("case", ["deg180", "quads", "signs", "wide_detector", "two_frames"])
def test_radon_adjoint_by_mirrored_tile_pairs(case, monkeypatch):
    """Round 6: k_radon_adj_quad — a workgroup owns the orbit of a 32 x 32 tile under the two mirrors of the grid and evaluates the BASE
    geometry of a quad once for two members (the column-mirrored pair shares slots 0 / 1, the row-mirrored pair slots 2 / 3).  Same taps,
    same fixed-point weights as k_radon_adj_tile, another summation order: against that kernel (TRK_RADON_NO_ADJQ=1) at fp32 summation
    noise, against the float64 oracle at 1e-5, the adjoint identity with the forward, and run to run to the bit."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
    N = 1024
    monkeypatch.setenv("TRK_RADON_ADJQ_MIN", "1024")       # (the product takes this kernel from 2048^2 on: read when the handle is made)
    ang = {"deg180": np.linspace(0, np.pi, 180, endpoint=False),
           "quads": np.deg2rad([10.0, 80.0, 100.0, 170.0, 11.0, 79.0, 101.0, 169.0, 12.0, 78.0, 102.0, 13.0]),
           "signs": np.deg2rad([20.0, 160.0, -20.0, 200.0, 70.0, 110.0, -70.0, 250.0, 290.0, 340.0, -160.0, 21.0, 69.0, 111.0, 159.0]),
           "wide_detector": np.linspace(0, np.pi, 36, endpoint=False),
           "two_frames": np.deg2rad(np.arange(2)[:, None] * 0.5 + 5.0 * np.arange(36)[None, :])}[case]
    nd = 1500 if case == "wide_detector" else N
    if case == "two_frames":
        R = BlockDiagOp([Radon2DParallel(N, a, n_det=nd) for a in ang])
    else:
        R = Radon2DParallel(N, ang, n_det=nd)
    eng = R.engine
    g = torch.Generator(device=eng.device).manual_seed(7)
    y = torch.randn(R.shape[0], device=eng.device, generator=g)
    z_new = R.apply(y, transpose=True).clone()
    monkeypatch.setenv("TRK_RADON_NO_ADJQ", "1")
    z_old = R.apply(y, transpose=True).clone()
    monkeypatch.delenv("TRK_RADON_NO_ADJQ")
    z_again = R.apply(y, transpose=True)
    assert torch.equal(z_new, z_again)
    scale = float(z_old.abs().max())
    dev = float((z_new - z_old).abs().max()) / scale
    assert dev < 2e-6, dev                                          # another summation order: the same sums
    assert float(torch.linalg.norm((z_new - z_old).double()) / torch.linalg.norm(z_old.double())) < 3e-7
    if case in ("quads", "wide_detector"):                          # the oracle's sparse matrix at 1024^2: the small angle sets only
        Ro = O.Radon2D(N, ang, n_det=nd)
        yh = y.cpu().numpy().astype(np.float64)
        assert relerr(z_new.cpu().numpy().astype(np.float64), Ro.T @ yh) < 1e-5
    x = torch.randn(R.shape[1], device=eng.device, generator=g)
    S = eng.scalars(2)
    eng.dot(R.apply(x), y, S.ref(0))
    eng.dot(x, z_new, S.ref(1))
    dd = S.host()
    assert abs(dd[0] - dd[1]) <= 1e-6 * float(torch.linalg.norm(R.apply(x).double()) * torch.linalg.norm(y.double()))


@pytest.mark.parametrize("N,na,nd", [(200, 90, 200), (256, 180, 256), (288, 60, 410), (512, 180, 512), (800, 40, 800)])
def test_radon_adjoint_with_the_angles_of_a_tile_split_over_workgroups(N, na, nd):
    """Small images, many angles: the adjoint runs 32 x 32 tiles whose angles are split over 4 or 8 workgroups; the partial tiles
    meet in the workgroup that takes the last ticket (k_radon_adj_tile, nsplit > 1).  (a) Against the float64 oracle at 1e-5, at
    sizes with ragged tiles (200, 288, 800 are not multiples of 32) and angle counts that do not divide evenly; (b) the hand-off
    between workgroups (write-through stores, ticket, loads past the L1) under load: 40 applies interleaved with forward applies
    and a different sinogram in between, every result bit-identical to the first — a stale partial tile would differ."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel
    rng = np.random.default_rng(N + na)
    ang = np.linspace(0, np.pi, na, endpoint=False)
    R, Ro = Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)
    y = rng.standard_normal(Ro.shape[0]).astype(np.float32)
    ref = Ro.T @ y.astype(np.float64)
    eng = R.engine
    yd = torch.from_numpy(y).to(eng.device)
    y2 = torch.from_numpy(rng.standard_normal(Ro.shape[0]).astype(np.float32)).to(eng.device)
    first = R.apply(yd, transpose=True).clone()
    assert relerr(first.cpu().numpy().astype(np.float64), ref) < 1e-5
    out, tmp, sino = torch.empty_like(first), torch.empty_like(first), torch.empty_like(yd)
    for k in range(40):
        R.apply(y2, out=tmp, transpose=True)          # other partial tiles in the same buffers
        if k % 3 == 0:
            R.apply(tmp, out=sino)
        R.apply(yd, out=out, transpose=True)
        assert torch.equal(out, first), k
    # the fused half step on the same path: a * A^T y + b * z with its norm
    z = torch.from_numpy(rng.standard_normal(N * N).astype(np.float32)).to(eng.device)
    S = eng.scalars(1)
    o2 = torch.empty_like(first)
    R.apply_axpby(yd, 1.0, -0.5, z, o2, transpose=True, sumsq=S.ref(0))
    want = first.double() - 0.5 * z.double()
    assert relerr(o2.double().cpu().numpy(), want.cpu().numpy()) < 1e-6
    assert abs(S.host()[0] - float((o2.double() ** 2).sum())) <= 1e-9 * float((o2.double() ** 2).sum())


@pytest.mark.parametrize("N,na,nd,nt", [(128, 7, 100, 1), (192, 33, 300, 1), (320, 50, 453, 1), (512, 180, 724, 1), (256, 15, 364, 5),
                                        (448, 3, 700, 2)])
def test_radon_forward_with_the_band_resident_in_lds(N, na, nd, nt, monkeypatch):
    """Small images (N a multiple of 64 up to 512): the forward projector keeps a 64-row band of the image — of the transposed image
    for the column-driven angles — resident in LDS (k_radon_fwd_band).  (a) Against the float64 oracle; (b) against the per-wave-window
    kernel it replaces (TRK_RADON_NO_BANDRES=1 at create: same tables, same weights, other partial sums: 1e-6); (c) both ways the
    column bands are filled — transposed while loading (a plain apply) and from the transposed copy the adjoint leaves in a chain
    (trk_op_apply_axpby with the hints of a Golub-Kahan step): the same bits; (d) angle sets of ONE marching mode, detectors that are not
    a multiple of 64, several frames in one launch."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import Radon2DParallel, BlockDiagOp
    rng = np.random.default_rng(N + na)
    frames = [np.linspace(0.05 * f, np.pi + 0.05 * f, na, endpoint=False) for f in range(nt)]
    if na == 3:
        frames = [np.array([0.1, 0.2, 2.9]) + 0.01 * f for f in range(nt)]                   # row-driven angles only
    def make():
        ops = [Radon2DParallel(N, a, n_det=nd) for a in frames]
        return ops[0] if nt == 1 else BlockDiagOp(ops)
    R = make()
    monkeypatch.setenv("TRK_RADON_NO_BANDRES", "1")
    Rw = make()
    monkeypatch.delenv("TRK_RADON_NO_BANDRES")
    eng = R.engine
    x = rng.standard_normal(nt * N * N).astype(np.float32)
    xd = torch.from_numpy(x).to(eng.device)
    got, old = R.apply(xd).clone(), Rw.apply(xd).clone()
    ref = np.concatenate([O.Radon2D(N, a, n_det=nd) @ x[f * N * N:(f + 1) * N * N].astype(np.float64) for f, a in enumerate(frames)])
    assert relerr(got.double().cpu().numpy(), ref) < 2e-6
    assert relerr(got.double().cpu().numpy(), old.double().cpu().numpy()) < 1e-6
    # (where the per-wave-window kernel runs 64-row bands too — few workgroups — the two kernels agree to the bit: same chunks, same sums)
    # in a chain: x = A^T y leaves its transpose behind, the forward apply that follows reads it
    y = torch.from_numpy(rng.standard_normal(R.shape[0]).astype(np.float32)).to(eng.device)
    xa = torch.empty_like(xd)
    z = torch.zeros_like(xd)
    R.apply_axpby(y, 1.0, 0.0, z, xa, transpose=True, hints=1)                               # TRK_HINT_OUT_FEEDS_OPPOSITE
    chained = torch.empty_like(got)
    R.apply_axpby(xa, 1.0, 0.0, torch.zeros_like(got), chained, hints=2)                     # TRK_HINT_INPUT_FROM_OPPOSITE
    plain = R.apply(xa.clone()).clone()
    assert torch.equal(plain, chained)


@pytest.mark.parametrize("N,na", [(64, 20), (512, 180), (1500, 24)])
def test_radon_invariants(N, na):
    """What pins the Radon operator: exact-adjoint identity, axis-aligned views = column / row sums, mass conservation,
    central chord of a centred disc."""
    from trips_py_amd.operators import Radon2DParallel
    rng = np.random.default_rng(1)
    ang = np.linspace(0, np.pi, na, endpoint=False)
    R = Radon2DParallel(N, ang)
    eng = R.engine
    x = torch.from_numpy(rng.standard_normal(N * N).astype(np.float32)).to(eng.device)
    y = torch.from_numpy(rng.standard_normal(R.shape[0]).astype(np.float32)).to(eng.device)
    Rx, RTy = R.apply(x), R.apply(y, transpose=True)
    S = eng.scalars(2)
    eng.dot(Rx, y, S.ref(0))
    eng.dot(x, RTy, S.ref(1))
    d = S.host()
    assert abs(d[0] - d[1]) <= 1e-6 * float(torch.linalg.norm(Rx.double()) * torch.linalg.norm(y.double()))
    img = rng.random((N, N))
    sino = (R @ img.reshape(-1)).reshape(na, N) * N
    assert np.allclose(sino[0], img.sum(axis=0), rtol=1e-5)
    if na % 2 == 0:
        assert np.allclose(sino[na // 2], img.sum(axis=1)[::-1], rtol=1e-5) or np.allclose(sino[na // 2], img.sum(axis=1), rtol=1e-5)
    ii, jj = np.meshgrid(np.arange(N) - (N - 1) / 2, np.arange(N) - (N - 1) / 2, indexing="ij")
    rad = N / 3.2
    disc = (ii ** 2 + jj ** 2 <= rad ** 2).astype(np.float64)
    sd = (R @ disc.reshape(-1)).reshape(na, N) * N
    assert np.allclose(sd.sum(axis=1), disc.sum(), rtol=2e-2)
    assert np.all(np.abs(sd[:, N // 2 - 1:N // 2 + 1].mean(axis=1) - 2 * rad) < 1.5)


@pytest.mark.parametrize("N,nt,na", [(48, 5, 9), (96, 3, 40)])
def test_dynamic_radon_equals_blockdiag_of_frames(N, nt, na):
    """BlockDiagOp of same-geometry Radon frames runs as ONE handle (trk_radon2d_dynamic_create): identical to applying the
    per-frame operators one by one, and to the oracle's block-diagonal operator.  (96, 3, 40): many angles per small frame — the
    adjoint with a tile's angles split over workgroups, frames in the grid's second dimension."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
    angs = [np.deg2rad(t + (180.0 / na) * np.arange(na)) for t in range(nt)]
    frames = [Radon2DParallel(N, a) for a in angs]
    F = BlockDiagOp(frames)
    rng = np.random.default_rng(3)
    x = rng.random(nt * N * N)
    y = rng.standard_normal(F.shape[0])
    per_f = np.concatenate([frames[t] @ x[t * N * N:(t + 1) * N * N] for t in range(nt)])
    per_a = np.concatenate([frames[t].T @ y[t * na * N:(t + 1) * na * N] for t in range(nt)])
    assert np.array_equal(F @ x, per_f) and np.array_equal(F.T @ y, per_a)
    Fo = O.BlockDiag([O.Radon2D(N, a) for a in angs])
    f = lambda v: v.astype(np.float32).astype(np.float64)
    assert relerr(F @ x, Fo @ f(x)) < 1e-5 and relerr(F.T @ y, Fo.T @ f(y)) < 1e-5


def test_dynamic_radon_large_frames_use_the_shared_window_kernel():
    """Frames of 1024^2 (window-sharing forward kernel, frame-major angle groups): one dynamic handle == per-frame handles."""
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
    N, nt, na = 1024, 2, 5
    angs = [np.deg2rad(3.0 * t + 36.0 * np.arange(na)) for t in range(nt)]
    frames = [Radon2DParallel(N, a) for a in angs]
    F = BlockDiagOp(frames)
    x = np.random.default_rng(4).random(nt * N * N)
    per_f = np.concatenate([frames[t] @ x[t * N * N:(t + 1) * N * N] for t in range(nt)])
    assert np.array_equal(F @ x, per_f)


def test_sparse_operator_and_reference_built_regularisers():
    """SparseOp (device CSR SpMV) vs scipy on the same matrices; a scipy.sparse L handed straight to an engine solver."""
    import scipy.sparse as sp
    from oracle import cpu_ref as O
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, SparseOp
    rng = np.random.default_rng(0)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    for M in (O.first_derivative_2d(17, 17), O.spacetime_derivative(9, 9, 4), sp.random(300, 4000, density=0.05, random_state=1, format="csr"),
              sp.random(50, 70, density=0.5, random_state=2, format="csr")):
        Op = SparseOp(M)
        M32 = sp.csr_matrix(M).astype(np.float32).astype(np.float64)
        x, y = rng.standard_normal(M.shape[1]), rng.standard_normal(M.shape[0])
        assert relerr(Op @ x, M32 @ f(x)) < 1e-6 and relerr(Op.T @ y, M32.T @ f(y)) < 1e-6
        S_ = Op.engine.scalars(1)
        out = Op.apply(torch.from_numpy(x.astype(np.float32)).to(Op.engine.device), sumsq=S_.ref(0))
        assert np.isclose(S_.host()[0], float((out.double() ** 2).sum()), rtol=1e-10)
    g = load_golden("gks_blur32_lam1e-2")
    N = int(g["N"])
    A = Blur2D(g["psf"], N, N)
    x, info = S.GKS(A, g["b"], O.first_derivative_2d(N, N), 3, int(g["n_iter"]), 1e-2, g["x_true"])   # scipy.sparse L
    assert relerr(x, g["x"]) < 1e-5


@pytest.mark.parametrize("shape,density", [((1000, 3000), 0.001), ((512, 512), 0.012), ((300, 4000), 0.05), ((64, 9000), 0.3), ((7, 5), 0.5),
                                           ((2000, 50), 0.02)])
def test_csr_group_kernel_every_group_size(shape, density):
    """k_csr_group<G> for every lanes-per-row choice (mean row lengths 3 .. 2700: G = 4 .. 64; and empty rows, rows longer than
    2 G, a matrix with fewer rows than a wave has groups), both directions, against scipy on the fp32-rounded operands."""
    import scipy.sparse as sp
    from trips_py_amd.operators import SparseOp
    M = sp.random(shape[0], shape[1], density=density, random_state=shape[0] + shape[1], format="csr")
    M.data = np.round(M.data * 64) / 64 + 1 / 64
    Op = SparseOp(M)
    rng = np.random.default_rng(shape[1])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    x, y = rng.standard_normal(shape[1]), rng.standard_normal(shape[0])
    assert relerr(Op @ x, M @ f(x)) < 5e-7 and relerr(Op.T @ y, M.T @ f(y)) < 5e-7
    # the k-column form (round 6: `A @ V` in passes of 8 / 4 / 2 / 1 columns, each pass reading the matrix once): every column
    # carries the single-vector kernel's chains and summation order, so it equals y = A x of that column to the bit — both directions
    X, Y = rng.standard_normal((shape[1], 15)), rng.standard_normal((shape[0], 15))
    AX, ATY = Op @ X, Op.T @ Y
    for j in range(15):
        assert np.array_equal(AX[:, j], Op @ X[:, j]), j
        assert np.array_equal(ATY[:, j], Op.T @ Y[:, j]), j


def test_sparse_dynamic_path_vs_the_reference_loader_and_solvers():
    """the real-data dynamic problems ship one sparse forward matrix that the reference's loader slices into
    per-frame blocks (io.py:197-229).  Fixture: generate_crossPhantom ITSELF run on a synthetic stand-in file, and the reference's
    CGLS / Hybrid_LSQR / MMGKS / GKS on what it returned (tools/make_goldens.py g9).  Here: the whole matrix as ONE CSR handle, the
    block-diagonal of its frames as ONE CSR handle (SparseBlockDiag.from_matrix, BlockDiagOp of SparseOps), and the solvers."""
    from test_oracle_golden import sparse_dynamic_golden, sparse_dynamic_vectors
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import BlockDiagOp, FirstDerivative2D, SparseBlockDiag, SparseOp, SpaceTimeDerivative, slice_dynamic_frames
    g, F = sparse_dynamic_golden()
    T, N, rpf = int(g["T"]), int(g["N"]), int(g["rows_per_frame"])
    npix = N * N
    xr, yr = sparse_dynamic_vectors(T * npix, T * rpf)
    Fop = SparseOp(F)
    assert relerr(Fop @ xr, g["F_fwd"]) < 1e-6 and relerr((Fop.T @ yr)[::37], g["F_adj_s"]) < 1e-6
    # the block-diagonal of the loader's frames: what is outside the blocks is gone, one handle for all frames
    D = SparseBlockDiag.from_matrix(F, T, rpf, npix)
    assert D.shape == F.shape and D.matrix.nnz == int(g["block_nnz"].sum()) < F.nnz
    assert relerr(D @ xr, g["blk_fwd"]) < 1e-6 and relerr((D.T @ yr)[::37], g["blk_adj_s"]) < 1e-6
    AA, B = slice_dynamic_frames(F, g["b"], T, rpf, npix)
    D2 = BlockDiagOp([SparseOp(a) for a in AA])                     # merged into one CSR handle, not T launches
    assert hasattr(D2, "matrix") and np.array_equal(D2 @ xr, D @ xr)
    # the solvers of the demo (2_demo_dynamic_CrossPhantom.ipynb cells 5, 15, 23) on the loader's outputs
    x, info = S.Hybrid_LSQR(F, g["b"], 10, 1e-2)                                    # a scipy.sparse matrix straight in
    assert info["its"] == int(g["lsqr_its"])
    bar("sparse_dynamic.lsqr.x", relerr(x.reshape(-1)[::16], g["lsqr_x_s"]), 1e-5)
    assert abs(np.linalg.norm(x) / float(g["lsqr_x_norm"]) - 1) < 1e-5
    x, info = S.CGLS(Fop, g["b"], np.zeros(T * npix), 12, 0)
    bar("sparse_dynamic.cgls.x", relerr(x.reshape(-1)[::16], g["cgls_x_s"]), 1e-5)
    assert np.allclose(info["relResidual"], g["cgls_relResidual"], rtol=1e-4)
    tf = int(g["frame"])
    x, info = S.MMGKS(AA[tf], B[tf], FirstDerivative2D(N), 2, 1, 1, 6, 1e-2, None, epsilon=0.1)
    bar("sparse_dynamic.mmgks_frame.x", relerr(x, g["mmgks_frame_x"]), 5e-5)
    assert np.allclose(info["Residual"], g["mmgks_frame_Residual"], rtol=2e-3)
    x, info = S.GKS(Fop, g["b"], SpaceTimeDerivative(N, T), 2, 4, 1e-2, None)
    bar("sparse_dynamic.gks.x", relerr(x.reshape(-1)[::16], g["gks_x_s"]), 1e-5)
    assert np.allclose(info["Residual"], g["gks_Residual"], rtol=1e-3)


def test_framelet_operator_matches_reference():
    from trips_py_amd.operators import create_framelet_operator
    g = load_golden("framelet_ops")
    W = create_framelet_operator(8, 6, 2)
    assert W.shape == g["dense_8_6_2"].shape
    assert np.allclose(W.todense(), g["dense_8_6_2"], atol=1e-7)
    for (n, m, l) in ((8, 6, 2), (12, 12, 1), (16, 10, 3)):
        W = create_framelet_operator(n, m, l)
        assert relerr(W @ g[f"x_{n}_{m}_{l}"], g[f"Wx_{n}_{m}_{l}"]) < 1e-6
        assert relerr(W.T @ g[f"y_{n}_{m}_{l}"], g[f"WTy_{n}_{m}_{l}"]) < 1e-6


@pytest.mark.parametrize("N,na", [(16, 7), (34, 12), (50, 20), (132, 24)])
def test_fanbeam_vs_bruteforce_oracle(N, na):
    """Fan-beam line projector (pinned to the reference's ASTRA images in the test below): the HIP traversal / gather kernels against the
    oracle's brute-force ray-pixel clipping, plus the geometry defaults of Tomography.define_proj_id."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FanBeam2D
    ang = np.linspace(0, np.pi, na, endpoint=False)
    A, Ao = FanBeam2D(N, views=na), O.FanBeam2D(N, ang)
    assert A.shape == Ao.shape == (na * int(np.sqrt(2) * N), N * N) and abs(A.pitch - 4.0 / 3.0) < 1e-15
    rng = np.random.default_rng(N)
    x, y = rng.random(N * N), rng.standard_normal(Ao.shape[0])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    # N chosen so that p = int(sqrt(2) N) is even: no detector ray runs exactly along a pixel boundary (a measure-zero
    # tie both implementations break arbitrarily).  Row-march form: lengths from pixel-sized quantities, 1e-5 (the Siddon pair
    # below differences ray parameters over a ~4N-long segment: 1e-4).
    assert int(np.sqrt(2) * N) % 2 == 0
    assert relerr(A @ x, Ao @ f(x)) < 1e-5, relerr(A @ x, Ao @ f(x))
    assert relerr(A.T @ y, Ao.T @ f(y)) < 1e-5, relerr(A.T @ y, Ao.T @ f(y))
    # several columns at once = the columns one by one
    X = rng.random((N * N, 3))
    assert np.array_equal((A @ X)[:, 1], A @ X[:, 1])
    # a detector inside the image's circumscribed circle: the general (Siddon traversal / slab clipping) pair
    B, Bo = FanBeam2D(N, views=na, origin_detector=0.5 * N), O.FanBeam2D(N, ang, odd=0.5 * N)
    assert relerr(B @ x, Bo @ f(x)) < 1e-4 and relerr(B.T @ y[:Bo.shape[0]], Bo.T @ f(y[:Bo.shape[0]])) < 1e-4


def test_fanbeam_against_the_astra_outputs_the_reference_holds():
    """The demo problem itself (tectonic 32^2, 30 views, p = 45: demos/demo_Tomo_small_scale.ipynb) through the HIP projector:
    against the decoded ASTRA sinogram / matrix images (rotation sense, detector order, layouts, boundary rays — see
    tests/astra_demo_image.py) and against the oracle INCLUDING the rays that run along a pixel boundary (p odd, N even: the
    central ray of views 0 and 15), which the image assigns to the larger column / row index."""
    from astra_demo_image import check_against_demo_images
    from conftest import load_golden
    from oracle import cpu_ref as O
    from trips_py_amd.problems import Tomography
    g = load_golden("fanbeam_demo_image")
    N, views = int(g["nx"]), int(g["views"])
    A, _, A_mis = Tomography(CommitCrime=False).forward_Op(N, N, views)
    x = g["phantom"].reshape(-1)
    dense = A @ np.eye(N * N)
    # (through A, not A_mis: ASTRA computes in float32, where the reference's 1e-8 angle shift cannot move a ray off a pixel
    #  boundary — 1.6e-7 pixels at most, under half a float32 spacing at 15.5 — so on the two boundary rays the reference's A_mis IS
    #  its A, and the image shows the tie rule's value (11.5 and 6.0) on both; the engine resolves 1e-9 pixels and its A_mis splits
    #  those two rays between the neighbouring columns / rows, 10.75 and 8.0 on this phantom — everywhere else the two agree to 1e-7)
    out = check_against_demo_images(g, A @ x, dense)
    s_mis = (A_mis @ x).reshape(views, -1)
    mask = np.ones_like(s_mis, dtype=bool)
    mask[0, 22] = mask[15, 22] = False
    assert relerr(s_mis[mask], (A @ x).reshape(views, -1)[mask]) < 1e-6
    assert out["sino_corr"] > 0.9998 and out["dense_corr"] > 0.98, out
    Ao = O.FanBeam2D(N, np.linspace(0, np.pi, views, endpoint=False))
    Mo = np.asarray(Ao.matrix().todense())
    assert relerr(dense, Mo) < 1e-5, relerr(dense, Mo)
    s = (A @ x).reshape(views, -1)
    assert abs(s[0, 22] - g["phantom"][:, 16].sum()) < 1e-5 and abs(s[15, 22] - g["phantom"][16].sum()) < 1e-5
    rng = np.random.default_rng(5)
    xr, yr = rng.random(N * N), rng.standard_normal(Ao.shape[0])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    assert relerr(A @ xr, Ao @ f(xr)) < 1e-5 and relerr(A.T @ yr, Ao.T @ f(yr)) < 1e-5


def test_parallel_beam_projector_is_the_far_source_limit_of_the_pinned_fan_beam():
    """The HIP parallel-beam projector against the HIP fan-beam projector with its source far away (the fan-beam
    convention is pinned to the reference's ASTRA images, test above; ten thousand widths here): same rotation sense, detector order and layout; the two
    interpolation models differ by 1e-3 on a smooth image."""
    from astra_demo_image import corr
    from trips_py_amd.operators import FanBeam2D, Radon2DParallel
    N, views = 128, 45
    ang = np.linspace(0, np.pi, views, endpoint=False)
    sod = 1e4 * N                     # rays parallel to 1e-4 rad: 0.006 pixels across the image
    F = FanBeam2D(N, angles=ang, n_det=N, source_origin=sod, origin_detector=float(N), det_pitch=(sod + N) / sod)
    R = Radon2DParallel(N, ang, n_det=N, scale=1.0)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    img = np.exp(-((ii - 40) ** 2 + (jj - 80) ** 2) / 300.0) + 0.5 * np.exp(-((ii - 90) ** 2 + (jj - 35) ** 2) / 500.0)
    sf, sr = (F @ img.reshape(-1)).reshape(views, N), (R @ img.reshape(-1)).reshape(views, N)
    assert corr(sf, sr) > 0.99999 and relerr(sf, sr) < 3e-3, (corr(sf, sr), relerr(sf, sr))
    assert max(corr(sf, sr[:, ::-1]), corr(sf, sr[::-1])) < 0.9


def test_fanbeam_invariants_and_problem_class():
    from trips_py_amd.problems import Tomography
    N, views = 128, 45
    A, A2, A_mis = Tomography(CommitCrime=False).forward_Op(N, N, views)
    assert A is A2 and A_mis.shape == A.shape
    eng = A.engine
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal(N * N).astype(np.float32)).to(eng.device)
    y = torch.from_numpy(rng.standard_normal(A.shape[0]).astype(np.float32)).to(eng.device)
    Ax, ATy = A.apply(x), A.apply(y, transpose=True)
    S = eng.scalars(2)
    eng.dot(Ax, y, S.ref(0))
    eng.dot(x, ATy, S.ref(1))
    d = S.host()
    assert abs(d[0] - d[1]) <= 1e-5 * float(torch.linalg.norm(Ax.double()) * torch.linalg.norm(y.double()))
    # a centred disc: the central ray of every view crosses a chord of length ~ 2 R
    ii, jj = np.meshgrid(np.arange(N) - (N - 1) / 2, np.arange(N) - (N - 1) / 2, indexing="ij")
    R = N / 4
    disc = (ii ** 2 + jj ** 2 <= R ** 2).astype(np.float64)
    sd = (A @ disc.reshape(-1)).reshape(views, A.n_det)
    mid = sd[:, A.n_det // 2 - 1:A.n_det // 2 + 1].mean(axis=1)
    assert np.all(np.abs(mid - 2 * R) < 1.5)
    # the 1e-8-shifted operator of the non-inverse-crime setup is the same operator to rounding
    assert relerr(A_mis @ disc.reshape(-1), sd.reshape(-1)) < 1e-4
    Ac, Ac2 = Tomography(CommitCrime=True).forward_Op(N, N, views)
    assert Ac.shape == A.shape and Ac2 is Ac
    # the demos' data helpers (Tomography.py:153-168, 203-227): tools/demo_helpers.py, outside the package
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from demo_helpers import demo_classes
    T = demo_classes()[2](CommitCrime=False)
    xr = rng.random(N * N)
    Aop, b, p, q, Amat = T.gen_data(xr, N, N, views)
    assert b.shape == (A.shape[0], 1) and (p, q) == (views, A.shape[0] // views) and Amat is Aop
    assert relerr(b.reshape(-1), A_mis @ xr) < 1e-6
    np.random.seed(2)
    bm, delta = T.add_noise(b, "Gaussian", 0.01)
    assert bm.shape == (p, q) and np.isclose(delta / np.linalg.norm(b), 0.01)


def _radon_case(name):
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
    if name == "static64":          # 16 x 16 tiles, records from the pre-pass (180 angles), 4 row bands? (N = 64: one band)
        return Radon2DParallel(64, np.linspace(0, np.pi, 40, endpoint=False))
    if name == "static512":         # BASELINE C3 shape: banded forward, 16 x 16 tiles, record pre-pass
        return Radon2DParallel(512, np.linspace(0, np.pi, 180, endpoint=False))
    if name == "static1030":        # shared-window forward kernel (N >= 1024), 32 x 32 tiles
        return Radon2DParallel(1030, np.linspace(0, np.pi, 36, endpoint=False), n_det=1100)
    if name == "static1024":        # round 6: the lean quad forward and the adjoint by mirrored tile pairs (whole 64 x 64 super-tiles, full quads)
        import os
        was = os.environ.get("TRK_RADON_ADJQ_MIN")
        os.environ["TRK_RADON_ADJQ_MIN"] = "1024"       # (the product takes the mirrored-pair adjoint from 2048^2 on; read at create)
        try:
            return Radon2DParallel(1024, np.linspace(0, np.pi, 60, endpoint=False))
        finally:
            if was is None:
                del os.environ["TRK_RADON_ADJQ_MIN"]
            else:
                os.environ["TRK_RADON_ADJQ_MIN"] = was
    if name == "dynamic":           # BASELINE C5 structure: frames in one handle, 15 angles per frame (records made in the tile kernel)
        return BlockDiagOp([Radon2DParallel(128, np.deg2rad(3.0 * t + 12.0 * np.arange(15))) for t in range(6)])
    if name == "tiny":              # N < 16: no tiled adjoint -> apply + axpby inside the C entry point
        return Radon2DParallel(12, np.linspace(0, np.pi, 7, endpoint=False))
    raise KeyError(name)


@pytest.mark.parametrize("case", ["static64", "static512", "static1030", "static1024", "dynamic", "tiny"])
def test_radon_fused_half_step_equals_apply_then_axpby(case):
    """trk_op_apply_axpby (out = a Op(x) + b z and ||out||^2 in the projector's own output pass: band reduction of the
    forward, tile gather of the adjoint; norm finished by the last workgroup) against trk_op_apply + trk_axpby, both
    directions, device-scalar coefficients as Golub-Kahan uses them, with and without z; then the hinted chain
    forward -> adjoint -> forward (records / transposed image left behind by the producing apply) against the plain one."""
    from trips_py_amd.engine import Coef
    A = _radon_case(case)
    eng = A.engine
    assert A.native_axpby
    m, n = A.shape
    g = torch.Generator(device=eng.device).manual_seed(11)
    x, zx = torch.rand(n, device=eng.device, generator=g), torch.randn(n, device=eng.device, generator=g)
    y, zy = torch.randn(m, device=eng.device, generator=g), torch.randn(m, device=eng.device, generator=g)
    S = eng.scalars(8)
    S.set(0, np.array([3.7, 0.61, 0, 0, 0, 0, 0, 0]))
    ca, cb = Coef(1.0, den=S.ref(0), sqrt_den=True), Coef(-1.0, num=S.ref(0), den=S.ref(1), sqrt_num=True, sqrt_den=True)
    ca_abs, cb_abs = 1.0 / np.sqrt(3.7), np.sqrt(3.7) / np.sqrt(0.61)
    for tr, xin, z in ((False, x, zy), (True, y, zx)):
        nout = n if tr else m
        t, want, got = eng.empty(nout), eng.empty(nout), eng.empty(nout)
        A.apply(xin, out=t, transpose=tr)
        for zz, bb in ((z, cb), (None, 0.0)):
            eng.axpby(ca, t, bb, zz, want, sumsq=S.ref(2))
            A.apply_axpby(xin, ca, bb, zz, got, transpose=tr, sumsq=S.ref(3))
            # the projector combines in float64 (one rounding of a Op(x) + b z), trk_axpby in fp32 with rounded coefficients:
            # equal to a unit in the last place of the larger operand (TRK_RADON_EPI_F32=1: to the bit)
            tol = 2.0 ** -22 * (float(ca_abs) * t.abs() + (0 if zz is None else float(cb_abs) * zz.abs())) + 1e-30
            assert bool(((got - want).abs() <= tol).all()), (tr, zz is None, float((got - want).abs().max()))
            s = S.host(2, 4)
            assert abs(s[0] - s[1]) <= 1e-6 * s[0]
            A.apply_axpby(xin, ca, bb, zz, got, transpose=tr, sumsq=S.ref(4))      # run to run: the same bits
            assert S.host(4, 5)[0] == s[1]
    # chain with hints: u1 = A x ; v1 = A^T u1 (records from the forward) ; u2 = A v1 (transposed copy from the adjoint)
    F, T = A.OUT_FEEDS_OPPOSITE, A.INPUT_FROM_OPPOSITE
    u1, v1, u2 = eng.empty(m), eng.empty(n), eng.empty(m)
    A.apply_axpby(x, ca, cb, zy, u1, sumsq=S.ref(5), hints=F)
    A.apply_axpby(u1, ca, cb, zx, v1, transpose=True, sumsq=S.ref(6), hints=F | T)
    A.apply_axpby(v1, ca, cb, zy, u2, sumsq=S.ref(7), hints=F | T)
    hinted = S.host(5, 8).copy()
    u1p, v1p, u2p = eng.empty(m), eng.empty(n), eng.empty(m)
    A.apply_axpby(x, ca, cb, zy, u1p, sumsq=S.ref(5))
    A.apply_axpby(u1p, ca, cb, zx, v1p, transpose=True, sumsq=S.ref(6))
    A.apply_axpby(v1p, ca, cb, zy, u2p, sumsq=S.ref(7))
    assert torch.equal(u1, u1p) and torch.equal(v1, v1p) and torch.equal(u2, u2p)
    assert np.array_equal(hinted, S.host(5, 8))
    # an apply in between takes the side buffers: the promise of the hint is void and the consumer re-derives them
    A.apply_axpby(x, ca, cb, zy, u1, hints=F)
    other = A.apply(zy, transpose=True)
    A.apply_axpby(u1, ca, cb, zx, v1, transpose=True, hints=T)
    assert torch.equal(v1, v1p) and other.shape[0] == n
    A.apply_axpby(u1, ca, cb, zx, v1, transpose=True, hints=F)
    A.apply(x)
    A.apply_axpby(v1, ca, cb, zy, u2, hints=T)
    assert torch.equal(u2, u2p)


@pytest.mark.parametrize("case", ["static64", "static512", "dynamic"])
def test_radon_deferred_norm_is_finished_by_the_next_chained_apply_or_flush(case):
    """TRK_HINT_SUMSQ_DEFERRED: the norm of a fused apply stays block partials inside the operator; the next chained apply
    adds them for its own coefficients and stores the finished scalar; trk_op_flush or any other call on the operator
    finishes it too.  Same vectors as the undeferred chain (the scalar to fp64 rounding: another summation order)."""
    from trips_py_amd.engine import Coef
    A = _radon_case(case)
    eng = A.engine
    m, n = A.shape
    g = torch.Generator(device=eng.device).manual_seed(5)
    u0, v0 = torch.randn(m, device=eng.device, generator=g), torch.randn(n, device=eng.device, generator=g)
    F, T, D = A.OUT_FEEDS_OPPOSITE, A.INPUT_FROM_OPPOSITE, A.SUMSQ_DEFERRED

    def chain(defer):
        S = eng.scalars(4)
        S.set(0, np.array([2.5, -1.0, -1.0, -1.0]))
        v1, u1, v2 = eng.empty(n), eng.empty(m), eng.empty(n)
        d = D if defer else 0
        # v1 = A^T u0 / sqrt(S0) - v0 ; S1 = |v1|^2     (deferred)
        A.apply_axpby(u0, Coef(1.0, den=S.ref(0), sqrt_den=True), -1.0, v0, v1, transpose=True, sumsq=S.ref(1), hints=F | d)
        # u1 = A v1 / sqrt(S1) - sqrt(S1)/sqrt(S0) u0 ; S2 = |u1|^2   (reads S1: from the partials when deferred)
        A.apply_axpby(v1, Coef(1.0, den=S.ref(1), sqrt_den=True), Coef(-1.0, num=S.ref(1), den=S.ref(0), sqrt_num=True, sqrt_den=True),
                      u0, u1, sumsq=S.ref(2), hints=F | T | d)
        # v2 = A^T u1 / sqrt(S2) - sqrt(S2)/sqrt(S1) v1 ; S3
        A.apply_axpby(u1, Coef(1.0, den=S.ref(2), sqrt_den=True), Coef(-1.0, num=S.ref(2), den=S.ref(1), sqrt_num=True, sqrt_den=True),
                      v1, v2, transpose=True, sumsq=S.ref(3), hints=F | T | d)
        if defer:
            A.flush_deferred()
        return S.host(0, 4), v1, u1, v2

    s_ref, v1r, u1r, v2r = chain(False)
    s_def, v1d, u1d, v2d = chain(True)
    assert np.all(s_def > 0) and np.allclose(s_def, s_ref, rtol=1e-12)
    assert torch.equal(v1d, v1r)
    for a, b in ((u1d, u1r), (v2d, v2r)):
        assert float(torch.linalg.norm(a - b) / torch.linalg.norm(b)) < 1e-6
    # a plain apply on the operator finishes a pending norm as well
    S = eng.scalars(2)
    S.set(0, np.array([-1.0, -1.0]))
    t = eng.empty(n)
    A.apply_axpby(u0, 1.0, 0.0, None, t, transpose=True, sumsq=S.ref(0), hints=D)
    A.apply(v0)
    assert abs(S.host(0, 1)[0] - float(torch.sum(t.double() ** 2))) <= 1e-6 * S.host(0, 1)[0]


def test_half_step_entry_point_serves_any_operator():
    """Operators without a native form: trk_op_apply_axpby = apply into `out` + trk_axpby in place."""
    from trips_py_amd.engine import Coef
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    A = Blur2D(gauss_psf((7, 7), (2, 2))[0], 50, 70)
    eng = A.engine
    assert not A.native_axpby
    x, z = torch.rand(3500, device=eng.device), torch.rand(3500, device=eng.device)
    S = eng.scalars(3)
    S.set(0, np.array([2.0, 0, 0]))
    want, got = eng.empty(3500), eng.empty(3500)
    eng.axpby(Coef(1.0, den=S.ref(0)), A.apply(x, transpose=True), -0.25, z, want, sumsq=S.ref(1))
    A.apply_axpby(x, Coef(1.0, den=S.ref(0)), -0.25, z, got, transpose=True, sumsq=S.ref(2))
    assert torch.equal(got, want) and S.host(1, 2)[0] == S.host(2, 3)[0]
    with pytest.raises(ValueError):
        A.apply_axpby(x, 1.0, 1.0, z, x)


@pytest.mark.parametrize("N,na", [(128, 24), (192, 31), (512, 45), (640, 20)])
def test_fanbeam_forward_with_the_band_resident_in_lds(N, na, monkeypatch):
    """Images up to 1024 wide (N a multiple of 64): the fan-beam forward marches its rays through 64-row (32-row beyond 512 columns)
    bands held in LDS (k_fan_fwd_band) instead of gathering two taps per step through the texture path from padded copies.  The same
    integers and weights, other partial sums: against the gather march (TRK_FAN_NO_BANDRES=1) at 1e-6, against the brute-force float64
    oracle at the small size, and the pair stays matched."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FanBeam2D
    ang = np.linspace(0.0, 2 * np.pi, na, endpoint=False) + 0.013
    A = FanBeam2D(N, angles=ang)
    eng = A.engine
    g = torch.Generator(device=eng.device).manual_seed(N + na)
    x = torch.randn(N * N, device=eng.device, generator=g)
    got = A.apply(x).clone()
    monkeypatch.setenv("TRK_FAN_NO_BANDRES", "1")
    old = A.apply(x).clone()
    monkeypatch.delenv("TRK_FAN_NO_BANDRES")
    assert relerr(got.double().cpu().numpy(), old.double().cpu().numpy()) < 1e-6
    assert torch.equal(A.apply(x), got)                                   # (and back on the band-resident path: the same bits every time)
    if N <= 128:
        Ao = O.FanBeam2D(N, ang)
        assert relerr(got.double().cpu().numpy(), Ao @ x.double().cpu().numpy()) < 1e-5
    y = torch.randn(A.shape[0], device=eng.device, generator=g)
    lhs, rhs = float(got.double() @ y.double()), float(x.double() @ A.apply(y, transpose=True).double())
    assert abs(lhs - rhs) <= 2e-6 * float(torch.linalg.norm(got.double()) * torch.linalg.norm(y.double()))


@pytest.mark.parametrize("N,pitch,nd", [(64, 4.0, 24), (64, 3.0, 40), (96, 2.5, 60), (64, 1.2, 90)])
def test_fanbeam_coarse_detector_adjoint_is_matched(N, pitch, nd):
    """A detector pitch coarser than a pixel's footprint (2 w < 1): most pixels see NO ray of a view, and the adjoint's candidate
    interval is empty there — the ray next to it may lie several columns away and must weigh nothing (the guard on
    the first candidate).  The adjoint against the oracle's brute-force matrix and the adjoint identity."""
    from oracle import cpu_ref as O
    from trips_py_amd.operators import FanBeam2D
    na = 36
    ang = np.linspace(0, np.pi, na, endpoint=False) + 0.01
    A = FanBeam2D(N, angles=ang, n_det=nd, det_pitch=pitch)
    Ao = O.FanBeam2D(N, ang, n_det=nd, pitch=pitch)
    rng = np.random.default_rng(N + nd)
    x, y = rng.random(N * N), rng.standard_normal(A.shape[0])
    f = lambda a: a.astype(np.float32).astype(np.float64)
    assert relerr(A @ x, Ao @ f(x)) < 1e-5 and relerr(A.T @ y, Ao.T @ f(y)) < 1e-5, (relerr(A @ x, Ao @ f(x)), relerr(A.T @ y, Ao.T @ f(y)))
    Ax, ATy = A @ x, A.T @ y
    assert abs(Ax @ y - x @ ATy) <= 2e-6 * np.linalg.norm(Ax) * np.linalg.norm(y)
    # ... view by view (a single view of a coarse detector leaves most pixels untouched: they must get exactly nothing from it)
    A1, A1o = FanBeam2D(N, angles=ang[:1], n_det=nd, det_pitch=pitch), O.FanBeam2D(N, ang[:1], n_det=nd, pitch=pitch)
    untouched = np.asarray(abs(A1o.matrix()).sum(axis=0)).reshape(-1) == 0
    back = A1.T @ np.ones(nd)
    if pitch >= 2.5:
        assert untouched.any()
    assert np.all(back[untouched] <= 1e-6 * back.max())


@pytest.mark.parametrize("N,views", [(256, 90), (512, 180), (1000, 50)])
def test_fanbeam_matched_pair_at_demo_sizes(N, views):
    """Row-march fan-beam pair at the sizes the tomography demos use: <A x, y> = <x, A^T y> to fp32 summation error (the adjoint
    weighs by the forward's own integers), every ray's weights sum to its chord through the image square (a constant image), and
    the sinogram of a centred disc is the chord through the disc along every central ray."""
    from trips_py_amd.operators import FanBeam2D
    A = FanBeam2D(N, views=views)
    eng = A.engine
    g = torch.Generator(device=eng.device).manual_seed(N)
    x = torch.randn(N * N, device=eng.device, generator=g)
    y = torch.randn(A.shape[0], device=eng.device, generator=g)
    Ax, ATy = A.apply(x), A.apply(y, transpose=True)
    lhs, rhs = float(Ax.double() @ y.double()), float(x.double() @ ATy.double())
    assert abs(lhs - rhs) <= 2e-6 * float(torch.linalg.norm(Ax.double()) * torch.linalg.norm(y.double())), (lhs, rhs)
    ones = A.apply(torch.ones(N * N, device=eng.device)).double().cpu().numpy().reshape(views, A.n_det)
    # chord of the ray source -> detector pixel centre through the square [-N/2, N/2]^2, in float64 on the host
    th = A.angles[:, None]
    off = (np.arange(A.n_det) - 0.5 * (A.n_det - 1))[None, :] * A.pitch
    sx, sy = A.sod * np.sin(th), -A.sod * np.cos(th)
    dx, dy = -A.odd * np.sin(th) + off * np.cos(th) - sx, A.odd * np.cos(th) + off * np.sin(th) - sy
    h = 0.5 * N
    with np.errstate(divide="ignore", invalid="ignore"):
        tx0, tx1 = (-h - sx) / dx, (h - sx) / dx
        ty0, ty1 = (-h - sy) / dy, (h - sy) / dy
    t0 = np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1))
    t1 = np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1))
    chord = np.clip(t1 - t0, 0.0, None) * np.hypot(dx, dy)
    assert np.abs(ones - chord).max() <= 2e-4 * N, np.abs(ones - chord).max()
    assert relerr(ones, chord) < 1e-5
