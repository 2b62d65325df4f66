"""history= modes of the solvers on the HIP engine (trips_py_amd._io.History): a stride, 'host' (iterates streamed to host
memory through a ring of device slots while the solver runs ahead) and '<file>.npy' (memory-mapped file) must return exactly
the iterates that the default on-device history returns — also through the C iteration loops of CGLS (chunked over the ring)
and with a ring much shorter than the iteration count."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _blur(N=96):
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * b.norm() / N
    return A, xt, b


@pytest.mark.parametrize("spec", [4, "host", "file"])
@pytest.mark.parametrize("fused", [False, True])
def test_cgls_history_modes(spec, fused, tmp_path, monkeypatch):
    from trips_py_amd import _io
    from trips_py_amd.solvers import CGLS
    A, xt, b = _blur()
    n = A.shape[1]
    if spec == "file":
        spec = str(tmp_path / "cgls.npy")
    x0 = torch.zeros(n, device=b.device)
    xa, ia = CGLS(A, b, x0, 37, 0, xt, fused=fused)
    # a ring of 6 slots for 37 iterates: the chunks wrap around it several times
    orig = _io.History.__init__
    monkeypatch.setattr(_io.History, "__init__", lambda self, eng, sp, count, nn, what, ring=None: orig(self, eng, sp, count, nn, what, ring=6))
    xb, ib = CGLS(A, b, x0, 37, 0, xt, fused=fused, history=spec)
    assert torch.equal(xa, xb) and np.array_equal(ia["relError"], ib["relError"])
    H = ib["xHistory"]
    assert H.iterations == (list(range(37)) if isinstance(spec, str) else [3, 7, 11, 15, 19, 23, 27, 31, 35, 36])
    for j, k in enumerate(H.iterations):
        ref = ia["xHistory"][k].reshape(-1)
        got = H[j].reshape(-1)
        assert isinstance(got, torch.Tensor) and got.dtype == torch.float32      # torch callers read tensors, whatever the sink
        assert torch.equal(ref.float(), got.to(ref.device)), (j, k)


@pytest.mark.parametrize("spec", [2, "host"])
def test_projection_solvers_history_modes(spec):
    from trips_py_amd import solvers as S
    from trips_py_amd.operators import Blur2D, FirstDerivative2D
    g = load_golden("mmgks_blur32_p2q1_lam1e-2")
    N = int(g["N"])
    A, L = Blur2D(g["psf"], N, N), FirstDerivative2D(N)
    for name, call in (("MMGKS", lambda **kw: S.MMGKS(A, g["b"], L, 2, 1, 3, 10, 1e-2, g["x_true"], **kw)),
                       ("GKS", lambda **kw: S.GKS(A, g["b"], L, 3, 10, 1e-2, g["x_true"], **kw)),
                       ("Hybrid_LSQR", lambda **kw: S.Hybrid_LSQR(A, g["b"], 10, 1e-2, g["x_true"], **kw)),
                       ("Hybrid_GMRES", lambda **kw: S.Hybrid_GMRES(A, g["b"], 10, 1e-2, g["x_true"], **kw))):
        xa, ia = call()
        xb, ib = call(history=spec)
        assert np.array_equal(xa, xb), name
        H = ib["xHistory"]
        for j, k in enumerate(H.iterations):
            assert np.array_equal(H[j], ia["xHistory"][k]), (name, j, k)
        assert len(H) == (len(ia["xHistory"]) if isinstance(spec, str) else len(H.iterations))


@pytest.mark.parametrize("spec", [4, "file", 0, np.int64(1)])
def test_history_of_a_solve_that_stops_early(spec, tmp_path):
    """CGLS with tol > 0 stops before max_iter: a strided history still ends with the LAST iterate formed (it is not on the stride),
    a .npy sink holds exactly the iterates that exist, and 0 / NumPy integers are accepted as strides (0 = False, 1 = True)."""
    from trips_py_amd.solvers import CGLS
    A, xt, b = _blur()
    n = A.shape[1]
    x0 = torch.zeros(n, device=b.device)
    xa, ia = CGLS(A, b, x0, 200, 5e-4, xt)
    its = int(ia["its"])
    assert 6 < its < 200 and len(ia["xHistory"]) == its
    path = str(tmp_path / "early.npy")
    xb, ib = CGLS(A, b, x0, 200, 5e-4, xt, history=path if spec == "file" else spec)
    assert torch.equal(xa, xb) and int(ib["its"]) == its
    H = ib["xHistory"]
    if isinstance(spec, (int, np.integer)) and spec == 0:
        assert H == []
    elif isinstance(spec, (int, np.integer)) and spec == 1:
        assert len(H) == its and torch.equal(H[its - 1].reshape(-1), xa.reshape(-1))
    elif spec == 4:
        want = list(range(3, its, 4))
        want += [] if want and want[-1] == its - 1 else [its - 1]
        assert H.iterations == want
        assert torch.equal(H[len(want) - 1].reshape(-1), xa.reshape(-1))
        for j, k in enumerate(want):
            assert torch.equal(H[j].reshape(-1), ia["xHistory"][k].reshape(-1))
    else:
        assert len(H) == its
        on_disk = np.load(path)
        assert on_disk.shape == (its, n) and on_disk.dtype == np.float32
        assert np.array_equal(on_disk[its - 1], xa.reshape(-1).cpu().numpy())
