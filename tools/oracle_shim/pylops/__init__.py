"""Container-only stand-in for the `pylops` package (absent from this image, no network).

TEST TOOLING, NOT PRODUCT.  Used only by tools/make_goldens.py, in the build
container, so that the reference package under /root/reference can be imported
and *run* to generate golden vectors.  It is our own code: it implements the few
pieces of the public PyLops >= 2 operator protocol the reference touches
(LinearOperator algebra, FunctionOperator, Identity, BlockDiag, and — restated
from PyLops' published semantics, NOT its code — FirstDerivative, Kronecker,
VStack) and nothing of the reference.  It never travels to the GPU box as part of the product path.

Behaviour that matters for faithfulness to PyLops >= 2:
  * `Op @ x` / `Op * x` with a 1-D operand -> matvec; with a 2-D operand ->
    matmat, which hands 1-D columns to matvec and stacks the results as columns.
  * `Op.T` / `Op.H` -> adjoint operator (real operators: same thing).
  * `scalar * Op`, `Op + Op`, `Op @ Op` -> lazily composed operators.
  * `.todense()` -> apply to identity columns.
"""
import numpy as np

__version__ = "0+oracle.shim"


class LinearOperator:
    def __init__(self, shape=None, dtype="float64"):
        self.shape = tuple(shape) if shape is not None else None
        self.dtype = np.dtype(dtype)
        self.explicit = False

    # ---- to be provided by subclasses
    def _matvec(self, x):
        raise NotImplementedError

    def _rmatvec(self, x):
        raise NotImplementedError

    # ---- protocol
    def matvec(self, x):
        x = np.asarray(x)
        if x.ndim == 2 and x.shape[1] == 1:
            x = x[:, 0]
        y = np.asarray(self._matvec(x))
        return y.reshape(-1)

    def rmatvec(self, x):
        x = np.asarray(x)
        if x.ndim == 2 and x.shape[1] == 1:
            x = x[:, 0]
        y = np.asarray(self._rmatvec(x))
        return y.reshape(-1)

    def matmat(self, X):
        X = np.asarray(X)
        return np.stack([self.matvec(X[:, j]) for j in range(X.shape[1])], axis=1)

    def rmatmat(self, X):
        X = np.asarray(X)
        return np.stack([self.rmatvec(X[:, j]) for j in range(X.shape[1])], axis=1)

    def dot(self, x):
        if isinstance(x, LinearOperator):
            return _Product(self, x)
        if np.isscalar(x):
            return _Scaled(self, x)
        x = np.asarray(x)
        if x.ndim == 1:
            return self.matvec(x)
        if x.ndim == 2:
            return self.matmat(x)
        raise ValueError("operand must be 1-D or 2-D")

    def __matmul__(self, x):
        return self.dot(x)

    def __mul__(self, x):
        return self.dot(x)

    def __call__(self, x):
        return self.dot(x)

    def __rmul__(self, a):
        if np.isscalar(a):
            return _Scaled(self, a)
        return NotImplemented

    def __rmatmul__(self, X):
        # ndarray @ Op  ==  (Op.T @ ndarray.T).T
        X = np.asarray(X)
        if X.ndim == 1:
            return self.rmatvec(X)
        return self.rmatmat(X.T).T

    __array_priority__ = 1000.0

    def __add__(self, other):
        return _Sum(self, other)

    def __neg__(self):
        return _Scaled(self, -1.0)

    def __sub__(self, other):
        return _Sum(self, -other)

    def adjoint(self):
        return _Adjoint(self)

    def transpose(self):
        return _Adjoint(self)

    H = property(adjoint)
    T = property(transpose)

    def todense(self):
        n = self.shape[1]
        return self.matmat(np.eye(n, dtype=self.dtype))


class _Adjoint(LinearOperator):
    def __init__(self, op):
        super().__init__((op.shape[1], op.shape[0]), op.dtype)
        self.op = op

    def _matvec(self, x):
        return self.op.rmatvec(x)

    def _rmatvec(self, x):
        return self.op.matvec(x)

    def adjoint(self):
        return self.op

    def transpose(self):
        return self.op

    H = property(adjoint)
    T = property(transpose)


class _Scaled(LinearOperator):
    def __init__(self, op, a):
        super().__init__(op.shape, op.dtype)
        self.op, self.a = op, a

    def _matvec(self, x):
        return self.a * self.op.matvec(x)

    def _rmatvec(self, x):
        return self.a * self.op.rmatvec(x)


class _Sum(LinearOperator):
    def __init__(self, a, b):
        assert a.shape == b.shape
        super().__init__(a.shape, a.dtype)
        self.a, self.b = a, b

    def _matvec(self, x):
        return self.a.matvec(x) + self.b.matvec(x)

    def _rmatvec(self, x):
        return self.a.rmatvec(x) + self.b.rmatvec(x)


class _Product(LinearOperator):
    def __init__(self, a, b):
        assert a.shape[1] == b.shape[0]
        super().__init__((a.shape[0], b.shape[1]), a.dtype)
        self.a, self.b = a, b

    def _matvec(self, x):
        return self.a.matvec(self.b.matvec(x))

    def _rmatvec(self, x):
        return self.b.rmatvec(self.a.rmatvec(x))


class FunctionOperator(LinearOperator):
    """pylops.FunctionOperator(f, fc, nr[, nc]) — nc defaults to nr."""

    def __init__(self, f, fc, nr, nc=None, dtype="float64", name="F"):
        nc = nr if nc is None else nc
        super().__init__((int(nr), int(nc)), dtype)
        self.f, self.fc = f, fc

    def _matvec(self, x):
        return np.squeeze(self.f(x))

    def _rmatvec(self, x):
        return np.squeeze(self.fc(x))


class Identity(LinearOperator):
    """pylops.Identity(N[, M]) — rectangular identities pad / truncate."""

    def __init__(self, N, M=None, inplace=True, dtype="float64", name="I"):
        M = N if M is None else M
        super().__init__((int(N), int(M)), dtype)

    def _matvec(self, x):
        N, M = self.shape
        y = np.zeros(N, dtype=np.result_type(x, self.dtype))
        k = min(N, M)
        y[:k] = x[:k]
        return y

    def _rmatvec(self, x):
        N, M = self.shape
        y = np.zeros(M, dtype=np.result_type(x, self.dtype))
        k = min(N, M)
        y[:k] = x[:k]
        return y


class BlockDiag(LinearOperator):
    def __init__(self, ops, dtype="float64"):
        self.ops = list(ops)
        nr = sum(o.shape[0] for o in self.ops)
        nc = sum(o.shape[1] for o in self.ops)
        super().__init__((nr, nc), dtype)
        self._ro = np.cumsum([0] + [o.shape[0] for o in self.ops])
        self._co = np.cumsum([0] + [o.shape[1] for o in self.ops])

    def _matvec(self, x):
        return np.concatenate([np.asarray(o @ x[self._co[i]:self._co[i + 1]]).reshape(-1)
                               for i, o in enumerate(self.ops)])

    def _rmatvec(self, x):
        return np.concatenate([np.asarray(o.T @ x[self._ro[i]:self._ro[i + 1]]).reshape(-1)
                               for i, o in enumerate(self.ops)])


class FirstDerivative(LinearOperator):
    """pylops.FirstDerivative(dims, axis=-1, sampling=1.0, kind="centered", edge=False, order=3, dtype=...), 1-D `dims`
    only — our restatement of PyLops' published 3-point centered stencil (its arithmetic is not available here):
        y[1:-1] = (0.5 x[2:] - 0.5 x[:-2]) / sampling,   y[0] = y[-1] = 0 (edge=False);
    adjoint: y[:-2] -= 0.5 x[1:-1] / sampling ; y[2:] += 0.5 x[1:-1] / sampling.
    The output array is created with the OPERATOR's dtype (float32 for the operators of operators_old.py:31)."""

    def __init__(self, dims, axis=-1, sampling=1.0, kind="centered", edge=False, order=3, dtype="float64", name="F"):
        if not np.isscalar(dims):
            raise NotImplementedError("oracle shim: FirstDerivative over a 1-D axis only")
        if kind != "centered" or order != 3 or edge:
            raise NotImplementedError("oracle shim: FirstDerivative(kind='centered', order=3, edge=False) only")
        super().__init__((int(dims), int(dims)), dtype)
        self.sampling = float(sampling)

    def _matvec(self, x):
        y = np.zeros(x.shape, self.dtype)
        y[1:-1] = (0.5 * x[2:] - 0.5 * x[:-2]) / self.sampling
        return y

    def _rmatvec(self, x):
        y = np.zeros(x.shape, self.dtype)
        y[:-2] -= (0.5 * x[1:-1]) / self.sampling
        y[2:] += (0.5 * x[1:-1]) / self.sampling
        return y


class Kronecker(LinearOperator):
    """pylops.Kronecker(Op1, Op2): y = kron(Op1, Op2) x, evaluated as Op1 (Op2 X^T)^T on X = x.reshape(m1, m2)."""

    def __init__(self, Op1, Op2, dtype="float64", name="K"):
        super().__init__((Op1.shape[0] * Op2.shape[0], Op1.shape[1] * Op2.shape[1]), dtype)
        self.Op1, self.Op2 = Op1, Op2

    def _matvec(self, x):
        X = x.reshape(self.Op1.shape[1], self.Op2.shape[1])
        Y = self.Op2.matmat(X.T).T
        return self.Op1.matmat(Y).ravel()

    def _rmatvec(self, x):
        X = x.reshape(self.Op1.shape[0], self.Op2.shape[0])
        Y = self.Op2.rmatmat(X.T).T
        return self.Op1.rmatmat(Y).ravel()


class VStack(LinearOperator):
    """pylops.VStack(ops): operators stacked by rows; the result arrays carry the common dtype of the operators."""

    def __init__(self, ops, dtype=None):
        self.ops = list(ops)
        dt = np.result_type(*[o.dtype for o in self.ops]) if dtype is None else dtype
        super().__init__((sum(o.shape[0] for o in self.ops), self.ops[0].shape[1]), dt)
        self._ro = np.cumsum([0] + [o.shape[0] for o in self.ops])

    def _matvec(self, x):
        y = np.zeros(self.shape[0], dtype=self.dtype)
        for i, o in enumerate(self.ops):
            y[self._ro[i]:self._ro[i + 1]] = o.matvec(x)
        return y

    def _rmatvec(self, x):
        y = np.zeros(self.shape[1], dtype=self.dtype)
        for i, o in enumerate(self.ops):
            y = y + o.rmatvec(x[self._ro[i]:self._ro[i + 1]])
        return y
