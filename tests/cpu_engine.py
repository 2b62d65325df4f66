"""TEST-ONLY engine: the HipEngine interface re-implemented with NumPy / torch-CPU so that the solvers' HOST LOGIC
(iteration structure, projected problems, lambda selection, info dictionaries, the distributed reduction points) can be
exercised without a GPU (`-m "not gpu"`), including world_size-2 runs over gloo.

It is never importable from the product (it lives under tests/), and it is not a fallback: trips_py_amd's default engine
raises without a GPU.  Arithmetic mimics the device: fp32 vector storage, float64 reductions.  Operators are adapters
over the oracle's operators (oracle/cpu_ref.py)."""
import numpy as np
import torch

from trips_py_amd.engine import SQRT_DEN, SQRT_NUM, Coef
from trips_py_amd.operators import LinearOperator


class CpuScalars:
    def __init__(self, n):
        self.a = np.zeros(int(n), dtype=np.float64)

    def ref(self, i):
        return (self, int(i))

    def view(self, i=0, j=None):
        return torch.from_numpy(self.a)[i:j]      # shares memory: collectives act in place

    def host(self, i=0, j=None):
        return self.a[i:j].copy()

    def host_later(self, i, j):
        vals = self.a[i:j].copy()

        class _Now:
            def get(self):
                return vals
        return _Now()

    def set(self, i, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        self.a[i:i + v.size] = v

    def __len__(self):
        return self.a.size


def _put(ref, vals):
    s, i = ref
    vals = np.atleast_1d(np.asarray(vals, dtype=np.float64))
    s.a[i:i + vals.size] = vals


def _get(ref, n=1):
    s, i = ref
    return s.a[i:i + n] if n > 1 else s.a[i]


def _coef(c):
    if not isinstance(c, Coef):
        return float(c)
    v = c.c
    if c.num is not None:
        t = _get(c.num)
        v *= np.sqrt(t) if c.flags & SQRT_NUM else t
    if c.den is not None:
        t = _get(c.den)
        v /= np.sqrt(t) if c.flags & SQRT_DEN else t
    return float(v)


def _d(t):
    return t.detach().numpy().astype(np.float64)


class CpuEngine:
    is_native = False

    def __init__(self, comm=None):
        self.device = torch.device("cpu")
        self.comm = comm
        self.world = 1 if comm is None else comm.world
        self.rank = 0 if comm is None else comm.rank
        self.reduction_points = 0
        self.halo_exchanges = 0

    def empty(self, n):
        return torch.zeros(int(n), dtype=torch.float32)

    zeros = empty

    def empty_basis(self, k, n):
        return torch.zeros((int(k), int(n)), dtype=torch.float32)

    def scalars(self, n):
        return CpuScalars(n)

    def to_vec(self, a, n=None):
        t = a.detach().reshape(-1).to(torch.float32).clone() if isinstance(a, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1)))
        if n is not None and t.numel() != n:
            raise ValueError(f"vector has {t.numel()} entries, expected {n}")
        return t

    def to_host(self, s):
        return s.host() if isinstance(s, CpuScalars) else s.detach().numpy().astype(np.float64)

    def synchronize(self):
        pass

    def allreduce(self, scal, i=0, j=None):
        self.reduction_points += 1
        if self.comm is not None and self.world > 1:
            self.comm.allreduce_sum_(scal.view(i, j))
        return scal

    def copy_scalars(self, src, i, dst, j, n=1):
        dst.a[j:j + n] = src.a[i:i + n]

    # reductions
    def dot(self, x, y, out):
        _put(out, np.dot(_d(x), _d(y)))

    def nrm2sq(self, x, out):
        _put(out, np.dot(_d(x), _d(x)))

    def diff_nrm2sq(self, x, y, out):
        d = _d(x) - _d(y)
        _put(out, np.dot(d, d))

    # axpy family
    def axpby(self, a, x, b, y, out, sumsq=None):
        r = np.float32(_coef(a)) * x.numpy()
        if y is not None:
            r = r + np.float32(_coef(b)) * y.numpy()
        out.copy_(torch.from_numpy(np.asarray(r, dtype=np.float32)))
        if sumsq is not None:
            _put(sumsq, np.dot(_d(out), _d(out)))

    def scale(self, a, x, out, sumsq=None):
        self.axpby(a, x, 0.0, None, out, sumsq)

    def mul(self, x, y, out):
        out.copy_(x * y)

    def mul_diff(self, w, x, y, out):
        out.copy_(w * (x - y))

    def mm_weights(self, x, y, eps, p, out):
        v = _d(x) - (0 if y is None else _d(y))
        out.copy_(torch.from_numpy(((v ** 2 + eps ** 2) ** (p / 2 - 1)).astype(np.float32)))

    def group_weights(self, d, groups, group_len, add, expo, copies, out):
        D = _d(d)[:groups * group_len].reshape(groups, group_len)
        w = ((D ** 2).sum(axis=1) + add) ** expo
        out.copy_(torch.from_numpy(np.tile(w, copies).astype(np.float32)))

    def isotv_weights(self, x, N, nt, u_tail, eps, q, out):
        X = _d(x)[:N * N * nt].reshape(N, N, nt)
        g1, g2 = np.zeros_like(X), np.zeros_like(X)
        g1[:, 1:-1, :] = 0.5 * X[:, 2:, :] - 0.5 * X[:, :-2, :]
        g2[1:-1, :, :] = 0.5 * X[2:, :, :] - 0.5 * X[:-2, :, :]
        e = (q - 2.0) / 4.0
        w = ((g1 ** 2 + g2 ** 2 + eps ** 2) ** e).reshape(-1)
        parts = [w, w]
        if u_tail is not None and u_tail.numel():
            parts.append((_d(u_tail) ** 2 + eps ** 2) ** e)
        out.copy_(torch.from_numpy(np.concatenate(parts).astype(np.float32)))

    def sparse_operator(self, M):
        from oracle import cpu_ref as O
        return OracleOp(O.MatrixOp(M), self)

    def cgls_update(self, gamma, delta, x, p, x_new, r, w, x_true, sums):
        step = np.float32(_get(gamma) / _get(delta))
        d = step * p.numpy()
        xn = x.numpy() + d
        x_new.copy_(torch.from_numpy(xn))
        r.copy_(torch.from_numpy(r.numpy() - step * w.numpy()))
        xn64 = xn.astype(np.float64)
        e = 0.0 if x_true is None else np.sum((xn64 - _d(x_true)) ** 2)
        _put(sums, [np.dot(xn64, xn64), np.dot(d.astype(np.float64), d.astype(np.float64)), e])

    # CGLS with one all-reduce per iteration (csrc/cgls_sharded.hip restated)
    def dot_pair(self, q, w, out3):
        qq = _d(q)
        _put(out3, [np.dot(qq, qq), 0.0 if w is None else np.dot(qq, _d(w)), 0.0 if w is None else np.dot(_d(w), _d(w))])

    def cgls_sharded_update(self, G4, gamma_prev, first, x, p, t, x_new, r, q, w, x_true, pub_delta, pub_gamma, partials, capacity):
        g, qq, qw, ww = _get(G4, 4)
        beta = 0.0 if first else g / _get(gamma_prev)
        delta = qq if first else qq + 2.0 * beta * qw + beta * beta * ww
        b32, a32 = np.float32(beta), np.float32(g / delta)
        pn = t.numpy().copy() if first else (b32 * p.numpy() + t.numpy()).astype(np.float32)
        wn = q.numpy().copy() if first else (b32 * w.numpy() + q.numpy()).astype(np.float32)
        p.copy_(torch.from_numpy(pn))
        w.copy_(torch.from_numpy(wn))
        d = a32 * pn
        xn = (x.numpy() + d).astype(np.float32)
        x_new.copy_(torch.from_numpy(xn))
        r.copy_(torch.from_numpy((r.numpy() - a32 * wn).astype(np.float32)))
        _put(pub_delta, delta)
        _put(pub_gamma, g)
        xn64, d64 = xn.astype(np.float64), d.astype(np.float64)
        e = 0.0 if x_true is None else np.sum((xn64 - _d(x_true)) ** 2)
        _put(partials, [np.dot(xn64, xn64), np.dot(d64, d64), e])
        return 1

    def finalize_batched(self, partials, nblocks, nvals, batches, out, out_stride):
        s, i = partials
        P = s.a[i:i + batches * nblocks * nvals].reshape(batches, nblocks, nvals).sum(axis=1)
        so, io = out
        for b in range(batches):
            so.a[io + b * out_stride: io + b * out_stride + nvals] = P[b]

    # tall-skinny
    def gemv_t(self, V, k, r, out_h, w2=None):
        rr = _d(r) if w2 is None else (r.numpy() * w2.numpy()).astype(np.float64)
        _put(out_h, _d(V[:k]) @ rr)

    def gemv_t2(self, V, k, r, r2, out_h2k):
        Vk = _d(V[:k])
        _put(out_h2k, np.concatenate((Vk @ _d(r), Vk @ _d(r2))))

    GRAM_TIKHONOV_MAX_K = 139

    def gram_tikhonov(self, GA, lda, GL, ldl, c, k, lam, y, Minv=None, ldm=0, k_from=0):
        (sa, ia), (sl, il) = GA, GL           # (the bordering form of the HIP engine solves the same system)
        A = np.array([sa.a[ia + i * lda: ia + i * lda + k] for i in range(k)])
        Lm = np.array([sl.a[il + i * ldl: il + i * ldl + k] for i in range(k)])
        _put(y, np.linalg.solve(A + lam * Lm, np.asarray(_get(c, k)).reshape(-1)))

    def cgs_coeffs(self, G, ldg, h, g_new, k, passes, c):
        Gs, g0 = G
        M = Gs.a[g0:g0 + ldg * ldg].reshape(ldg, ldg)
        if g_new is not None:
            gn = np.asarray(_get(g_new, k)).reshape(-1)
            M[k - 1, :k] = gn
            M[:k, k - 1] = gn
        if passes > 0:
            hh = np.asarray(_get(h, k)).reshape(-1)
            cc = np.zeros(k)
            for _ in range(passes):
                cc = cc + (hh - M[:k, :k] @ cc)
            _put(c, cc)

    def gemv_n(self, V, k, y, out, a=0.0, base=None, s=1.0, sumsq=None):
        o = s * (np.asarray(_get(y, k)).reshape(-1) @ _d(V[:k]))
        if base is not None:
            o = o + a * _d(base)
        out.copy_(torch.from_numpy(o.astype(np.float32)))
        if sumsq is not None:
            _put(sumsq, np.dot(_d(out), _d(out)))

    def bidiag_tikhonov(self, alpha_sq, alpha_stride, beta_sq, beta_stride, k, mu, beta0_sq, y, work=None,
                        y_over_alpha=False):
        (sa, ia), (sb, ib) = alpha_sq, beta_sq
        al = np.sqrt(sa.a[ia:ia + alpha_stride * k:alpha_stride])
        be = np.sqrt(sb.a[ib:ib + beta_stride * k:beta_stride])
        B = np.zeros((k + 1, k))
        B[np.arange(k), np.arange(k)] = al
        B[np.arange(1, k + 1), np.arange(k)] = be
        rhs = np.zeros(2 * k + 1)
        rhs[0] = np.sqrt(_get(beta0_sq))
        sol = np.linalg.lstsq(np.vstack((B, mu * np.eye(k))), rhs, rcond=None)[0]
        _put(y, sol / al if y_over_alpha else sol)

    def wgram(self, W, k, w, b1, G, c1=None, c2=None):
        Wk = _d(W[:k])
        ww = np.ones(Wk.shape[1]) if w is None else _d(w)
        _put(G, ((Wk * ww ** 2) @ Wk.T).reshape(-1))
        if b1 is not None:
            _put(c1, Wk @ (ww * _d(b1)))
            _put(c2, Wk @ (ww ** 2 * _d(b1)))


class OracleOp(LinearOperator):
    """Adapter: an oracle operator behind the engine-native `apply()` protocol (torch CPU fp32 in / out)."""

    def __init__(self, oracle_op, engine):
        self.o = oracle_op
        super().__init__(oracle_op.shape, engine)

    def _apply(self, x2, y2, transpose, sumsq):
        fn = self.o._adj if transpose else self.o._fwd
        for b in range(x2.shape[0]):
            y2[b].copy_(torch.from_numpy(fn(x2[b].numpy().astype(np.float64)).astype(np.float32)))
        if sumsq is not None:
            _put(sumsq, float((y2.double() ** 2).sum()))
