"""What agreement with the float64 oracle can an fp32-storage engine reach on C3 / C5 at all?  CPU only (the oracle twice).

The oracle's solver is run (a) as it is and (b) with every operator product rounded to float32 on the way in and out — the least
an engine that STORES its vectors in fp32 does to the iteration, with exact (float64) arithmetic everywhere else.  The distance
between the two runs is a floor for the parity bar of tests/test_gpu_configs_fullsize.py: un-reorthogonalised Golub-Kahan
(Hybrid_LSQR.py:73-110 via decompositions.py:230-255) amplifies the 6e-8 roundings, GKS's thrice re-orthogonalised basis does not.
`c5cgls`: CGLS (no regularisation) on the C5 data: past semi-convergence (iterate ~25) the recurrence amplifies the roundings by four
orders of magnitude — what tools/cgls_forms_accuracy.py measures for the engine's two arrangements on the GPU.
`c3emul` (round 4): the ENGINE's own arrangement of Hybrid-LSQR at a fixed lambda restated in NumPy — Golub-Kahan on unnormalised
vectors (trk_gk_step: include/trk.h), damped LSQR's short recurrence for the iterate (k_lsqr_damped_update: csrc/vecops.hip) — with
float64 arithmetic everywhere and a rounding to float32 exactly where the engine STORES a vector (u, v, w, x; the operator's output
once more, standing for the projector's own fp32 accumulation): what ANY fp32-storage run of that arrangement does, against the
float64 oracle's iterates.  VERDICT r03 asked whether the engine's 1.6e-3 at iterate 7 is the arrangement's floor or a defect.
usage: python3 tools/fp32_floor.py c3|c3emul|c5|c5cgls [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O  # noqa: E402


class Rounded(O._Op):
    """op with fp32-rounded operands and results."""

    def __init__(self, op):
        self.op, self.shape = op, op.shape

    def _fwd(self, x):
        return self.op._fwd(x.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)

    def _adj(self, y):
        return self.op._adj(y.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def r32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def engine_hybrid_lsqr_emulated(A, b, its, lam, store=r32, op_round=r32):
    """Iterates of the engine's fixed-lambda Hybrid-LSQR (solvers/Hybrid_LSQR.py, x_by_recurrence) with `store` applied wherever
    the engine writes a vector to memory and `op_round` to every operator product; identity functions give the exact arithmetic."""
    damp = float(np.sqrt(lam))      # Hybrid_LSQR.py:83-84: [B; sqrt(lam) I] -> LSQR's damping
    u = store(b)                                   # beta_1 u_1
    AB = [float(u @ u)]
    vprev = None
    w = x = None
    st = None
    out = []
    for k in range(its):
        beta = np.sqrt(AB[2 * k])
        t = op_round(A._adj(u))
        v = store(t / beta - ((beta / np.sqrt(AB[2 * k - 1])) * vprev if k else 0.0))      # alpha_k v_k
        AB.append(float(v @ v))
        alpha = np.sqrt(AB[2 * k + 1])
        t = op_round(A._fwd(v))
        un = store(t / alpha - (alpha / beta) * u)                                            # beta_{k+1} u_{k+1}
        AB.append(float(un @ un))
        # k_lsqr_damped_update
        bnext = np.sqrt(AB[2 * k + 2])
        if k == 0:
            rhobar, phibar, tw = alpha, np.sqrt(AB[0]), 0.0
        else:
            rhobar, tw, phibar = -st[0] * alpha, st[1] * alpha / st[2], st[3]
        rhobar1 = np.sqrt(rhobar * rhobar + damp * damp)
        phibar *= rhobar / rhobar1
        rho = np.sqrt(rhobar1 * rhobar1 + bnext * bnext)
        cs, sn = rhobar1 / rho, bnext / rho
        w = store(v / alpha - (tw * w if k else 0.0))
        x = store((x if k else 0.0) + (cs * phibar / rho) * w)
        st = (cs, sn, rho, sn * phibar)
        out.append(x.copy())
        u, vprev = un, v
    return out


def c3_problem():
    N, na = 512, 180
    ang = np.linspace(0, np.pi, na, endpoint=False)
    Ro = O.Radon2D(N, ang)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    xt = (((ii - 256) / 180.0) ** 2 + ((jj - 256) / 230.0) ** 2 < 1).astype(np.float64) + 0.5 * ((((ii - 300) / 60.0) ** 2 + ((jj - 200) / 40.0) ** 2) < 1)
    rng = np.random.default_rng(5)
    b = Ro @ xt.reshape(-1)
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    return Ro, b, xt


which = sys.argv[1]
if which == "c3emul":
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    Ro, b, xt = c3_problem()
    xa, ia = O.hybrid_lsqr(Ro, b.reshape(-1, 1), its, 1e-2, xt.reshape(-1, 1))
    ident = lambda a: np.asarray(a, dtype=np.float64)
    n_it = len(ia["xHistory"])
    for name, st, opr in (("float64 storage (the arrangement itself, exact arithmetic)", ident, ident),
                          ("float32 storage of u, v, w, x", r32, ident),
                          ("float32 storage and float32-rounded operator products", r32, r32)):
        # the reference forms no iterate at its first step (Hybrid_LSQR.py:77-78): its i-th iterate spans i + 2 basis vectors
        xs = engine_hybrid_lsqr_emulated(Ro, b, n_it + 1, 1e-2, st, opr)[1:]
        d = [rel(xe, np.ravel(xo)) for xe, xo in zip(xs, ia["xHistory"])]
        print(f"C3 engine arrangement, {name}: vs the float64 oracle's iterates")
        print("   per iterate:", " ".join(f"{v:.1e}" for v in d))
        print(f"   max over iterates 1-20: {max(d[:20]):.2e} (at {1 + int(np.argmax(d[:20]))}); max from 21 on: {max(d[20:]) if len(d) > 20 else float('nan'):.2e}; last: {d[-1]:.2e}")
    sys.exit(0)
if which == "c3":
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    N, na = 512, 180
    ang = np.linspace(0, np.pi, na, endpoint=False)
    Ro = O.Radon2D(N, ang)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    xt = (((ii - 256) / 180.0) ** 2 + ((jj - 256) / 230.0) ** 2 < 1).astype(np.float64) + 0.5 * ((((ii - 300) / 60.0) ** 2 + ((jj - 200) / 40.0) ** 2) < 1)
    rng = np.random.default_rng(5)
    b = Ro @ xt.reshape(-1)
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    xa, ia = O.hybrid_lsqr(Ro, b.reshape(-1, 1), its, 1e-2, xt.reshape(-1, 1))
    xb, ib = O.hybrid_lsqr(Rounded(Ro), b.reshape(-1, 1), its, 1e-2, xt.reshape(-1, 1))
    print(f"C3 Hybrid_LSQR {its} iterations: final x fp32-rounded products vs float64: {rel(xb, xa):.3e}")
    print("per iterate:", " ".join(f"{rel(hb, ha):.1e}" for ha, hb in zip(ia["xHistory"], ib["xHistory"])))
    print("relError max rel. difference:", float(np.max(np.abs(np.array(ib["relError"]) / np.array(ia["relError"]) - 1))))
else:
    cgls = which == "c5cgls"
    its = int(sys.argv[2]) if len(sys.argv) > 2 else (60 if cgls else 8)
    N, nt, na = 256, 32, 15
    angs = [np.deg2rad(t + 12.0 * np.arange(na)) for t in range(nt)]
    Fo = O.BlockDiag([O.Radon2D(N, a) for a in angs])
    Lo = O.SpaceTimeDerivative(N, nt)
    frames = []
    for t in range(nt):
        img = np.zeros((N, N))
        img[60 + 2 * t:100 + 2 * t, 40:200] = 1.0
        img[150:190, 30 + 3 * t:90 + 3 * t] = 0.6
        frames.append(img.reshape(-1))
    xt = np.concatenate(frames)
    rng = np.random.default_rng(9)
    b = Fo @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    if cgls:
        x0 = np.zeros((Fo.shape[1], 1))
        xa, ia = O.cgls(Fo, b.reshape(-1, 1), x0, its, 0, xt.reshape(-1, 1))
        xb, ib = O.cgls(Rounded(Fo), b.reshape(-1, 1), x0, its, 0, xt.reshape(-1, 1))
        print(f"C5 CGLS {its} iterations: final x fp32-rounded products vs float64: {rel(xb, xa):.3e}")
        print("per iterate:", " ".join(f"{rel(hb, ha):.1e}" for ha, hb in zip(ia["xHistory"], ib["xHistory"])))
        sys.exit(0)
    xa, ia = O.gks(Fo, b.reshape(-1, 1), Lo, 3, its, 1e-2, xt.reshape(-1, 1))
    xb, ib = O.gks(Rounded(Fo), b.reshape(-1, 1), Rounded(Lo), 3, its, 1e-2, xt.reshape(-1, 1))
    print(f"C5 GKS {its} iterations: final x fp32-rounded products vs float64: {rel(xb, xa):.3e}")
    print("per iterate:", " ".join(f"{rel(hb, ha):.1e}" for ha, hb in zip(ia["xHistory"], ib["xHistory"])))
    print("Residual max rel. difference:", float(np.max(np.abs(np.array(ib["Residual"]) / np.array(ia["Residual"]) - 1))))
