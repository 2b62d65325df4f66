"""NumPy emulation of the fixed-point ray-coordinate arithmetic of csrc/radon2d.hip (round 2), checked against the float64
oracle BEFORE the kernels were written: forward weights from Q32 = A32[a][d] + B32[a][tt] (24 fractional bits, columns mod
256), adjoint from the same tables for the nearest ray and +-inv for its two neighbours.  CPU only.

    python tools/experiments/radon_fixedpoint_emul.py [N]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cpu_ref as O  # noqa: E402

f32 = np.float32
F = 24
ONE = 1 << F


def tables(N, nd, th):
    half = 0.5 * (N - 1)
    ct, st = np.cos(th), np.sin(th)
    if abs(ct) >= abs(st):
        mode, inv, dq, k0, wgt, rinv = 0, 1.0 / ct, st / ct, half - half * st / ct, 1.0 / abs(ct), ct
    else:
        mode, inv, dq, k0, wgt, rinv = 1, -1.0 / st, ct / st, half - half * ct / st, 1.0 / abs(st), -st
    sdh = 0.5 * (nd - 1)
    d = np.arange(-2, nd + 2)
    A32 = (np.rint(((d - sdh) * inv + k0) * ONE).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    tt = np.arange(N)
    B32 = (np.rint(tt * dq * ONE).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    C = (sdh - (k0 + tt * dq) * rinv).astype(f32)          # d*_est = col * rinv + C[tt]
    return dict(mode=mode, inv=inv, dq=dq, k0=k0, wgt=wgt, rinv=rinv, sdh=sdh, A32=A32, B32=B32, C=C)


def forward(img, N, nd, T):
    """sino row of one angle; absolute columns from a float estimate +- the mod-256 column (generic kernel form)."""
    I = img if T["mode"] == 0 else img.T
    out = np.zeros(nd)
    d = np.arange(nd)
    base = (f32(d) - f32(T["sdh"])) * f32(T["inv"]) + f32(T["k0"])
    for tt in range(N):
        Q = (T["A32"][d + 2].astype(np.uint64) + np.uint64(T["B32"][tt])) & np.uint64(0xFFFFFFFF)
        cm = (Q >> np.uint64(F)).astype(np.int64)
        fr = (Q & np.uint64(ONE - 1)).astype(np.int64)
        qest = f32(tt) * f32(T["dq"]) + base
        ce = np.floor(qest).astype(np.int64)
        c = ce + (((cm - ce + 128) & 255) - 128)
        w1 = fr.astype(np.float64)
        w0 = ONE - w1
        row = I[tt]
        ok0 = (c >= 0) & (c < N)
        ok1 = (c + 1 >= 0) & (c + 1 < N)
        out += np.where(ok0, w0 * row[np.clip(c, 0, N - 1)], 0) + np.where(ok1, w1 * row[np.clip(c + 1, 0, N - 1)], 0)
    return out * T["wgt"] / ONE


def adjoint(srow, N, nd, T):
    """image contribution of one angle (gather form of the kernel, fp32 where the kernel is fp32)."""
    S = np.zeros(nd + 4)
    S[2:-2] = srow * T["wgt"]
    tt, col = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")     # mode 0: tt = i, col = j
    dstar = f32(col) * f32(T["rinv"]) + T["C"][tt]
    d0 = np.rint(dstar).astype(np.int64)
    d0c = np.clip(d0, -1, nd)                                            # out of range: weightless records
    tint = (T["A32"][d0c + 2].astype(np.int64) + T["B32"][tt].astype(np.int64) - (col.astype(np.int64) << F))
    tint = ((tint + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)                 # int32 wrap-around
    tf = tint.astype(f32)
    inv24 = f32(T["inv"] * ONE)
    w0 = f32(ONE) - np.abs(tf)
    wp = np.maximum(f32(ONE) - np.abs(tf + inv24), f32(0))
    wm = np.maximum(f32(ONE) - np.abs(tf - inv24), f32(0))
    inr = (d0 >= -1) & (d0 <= nd)
    val = np.where(inr, w0 * S[d0c + 2] + wp * S[d0c + 3] + wm * S[d0c + 1], 0.0) / ONE
    assert np.all(np.abs(tint[inr]) < ONE), "nearest ray farther than one pixel"
    return val if T["mode"] == 0 else val.T


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nd = N
    ang = np.array([0.3, 1.1, 2.0, 0.0, np.pi / 4, 3.0])
    Ro = O.Radon2D(N, ang, scale=1.0)
    M = Ro.matrix()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, N)).astype(f32).astype(np.float64)
    y = rng.standard_normal((len(ang), nd)).astype(f32).astype(np.float64)
    yo = (M @ x.reshape(-1)).reshape(len(ang), nd)
    fwd = np.stack([forward(x, N, nd, tables(N, nd, th)) for th in ang])
    print("forward  vs float64 oracle:", np.linalg.norm(fwd - yo) / np.linalg.norm(yo))
    xo = (M.T @ y.reshape(-1)).reshape(N, N)
    adj = sum(adjoint(y[a], N, nd, tables(N, nd, th)) for a, th in enumerate(ang))
    print("adjoint  vs float64 oracle:", np.linalg.norm(adj - xo) / np.linalg.norm(xo))
    print("<A x, y> - <x, A^T y> rel :", abs(np.sum(fwd * y) - np.sum(x * adj)) / abs(np.sum(fwd * y)))


if __name__ == "__main__":
    main()
