#!/usr/bin/env python
"""C3 (parallel-beam 512^2 x 180, Hybrid-LSQR, lambda = 1e-2) per-iterate distance from the float64 oracle, with the float64
instrument of csrc/ref64.hip separating the causes:

  chain64/w64    trk_gk_lsqr_chain, vectors in float64, weights from the float64 geometry   -> arrangement + kernels exact?
  chain32/w64    the same chain, vectors in float32                                         -> what fp32 STORAGE alone costs
  chain32/tab    vectors in float32, the product's fixed-point weights, float64 sums        -> + the 2^-24 weight grid
  solver/w64     the product's Hybrid_LSQR on an operator switched to float64 arithmetic    -> the product's host path, exact projector
  solver/tab     ... switched to table weights with float64 sums
  product        the product as it ships                                                    -> + the kernels' fp32 partial sums

GPU box:  python tools/r05_c3_instrument.py [n_iter=100] > profiles/r05/c3_instrument.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402  (checker)
from trips_py_amd import solvers as S  # noqa: E402
from trips_py_amd.operators import Radon2DParallel  # noqa: E402


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def problem(N=512, na=180):
    ang = np.linspace(0, np.pi, na, endpoint=False)
    Ro = O.Radon2D(N, ang)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    c = N / 2.0
    s = N / 512.0                                    # (the phantom of tests/test_gpu_configs_fullsize.py c3_numbers at N = 512)
    xt = (((ii - c) / (180.0 * s)) ** 2 + ((jj - c) / (230.0 * s)) ** 2 < 1).astype(np.float64) \
        + 0.5 * ((((ii - 300.0 * s) / (60.0 * s)) ** 2 + ((jj - 200.0 * s) / (40.0 * s)) ** 2) < 1)
    rng = np.random.default_rng(5)
    b = Ro @ xt.reshape(-1)
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    return ang, Ro, xt.reshape(-1), b


def main():
    its = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    na = int(sys.argv[3]) if len(sys.argv) > 3 else 180
    ang, Ro, xt, b = problem(N, na)
    t0 = time.time()
    xo, io = O.hybrid_lsqr(Ro, b.reshape(-1, 1), its, 1e-2, xt.reshape(-1, 1))
    print(f"# oracle: {its} iterations of Hybrid-LSQR at {N}^2 x {na} in {time.time() - t0:.0f} s on the host")
    R = Radon2DParallel(N, ang)
    dev = R.engine.device
    # the operators themselves, on the data and on white noise
    rng = np.random.default_rng(7)
    for tag, v, tr in (("A x_true", xt, False), ("A noise", rng.standard_normal(N * N), False),
                       ("A^T b", b, True), ("A^T noise", rng.standard_normal(na * N), True)):
        ref = (Ro.T @ v) if tr else (Ro @ v)
        v32 = torch.from_numpy(v.astype(np.float32)).to(dev)
        v64 = torch.from_numpy(v).to(dev)
        refr = (Ro.T @ v32.cpu().numpy().astype(np.float64)) if tr else (Ro @ v32.cpu().numpy().astype(np.float64))
        row = [relerr(R.apply_ref(v64, tr, "float64").cpu().numpy(), ref),
               relerr(R.apply_ref(v64, tr, "tables64").cpu().numpy(), ref),
               relerr(R.apply_ref(v32, tr, "float64").cpu().numpy(), refr),
               relerr(R.apply(v32, transpose=tr).cpu().numpy(), refr)]
        print(f"# operator {tag:10s}: f64/w64 {row[0]:.2e}   f64/tables {row[1]:.2e}   f32/w64 {row[2]:.2e}   product {row[3]:.2e}")
    runs = {}
    for tag, kw in (("chain64/w64", dict(storage="float64", weights="float64")), ("chain32/w64", dict(storage="float32", weights="float64")),
                    ("chain32/tab", dict(storage="float32", weights="tables64")), ("chain64/tab", dict(storage="float64", weights="tables64"))):
        x, info = S.Hybrid_LSQR(R, b, its, 1e-2, xt, dtype="float64", **kw)
        runs[tag] = info["xHistory"]
    # (chunk_fwd, chunk_adj): the instrument's kernels with EMULATED fp32 partial sums — which of the product's sums matter?
    sums = {"solver/w64": ("float64", 0, 0), "solver/tab": ("tables64", 0, 0), "tab/f16,a180": ("tables64", 16, 180),
            "tab/f16,a0": ("tables64", 16, 0), "tab/f0,a180": ("tables64", 0, 180), "tab/f4,a32": ("tables64", 4, 32),
            "tab/f8,a16": ("tables64", 8, 16),
            # round 6: float64 sums, table weights, but the adjoint's two NEIGHBOUR rays weighed by the product's rule (from the nearest
            # ray's t0 and the detector spacing) with 1 - |inv| in float64 / in fp32: what the weight rule alone costs
            "tab/nbr64": ("tables64", 0, -1), "tab/nbr32": ("tables64", 0, -2),
            "product": ("product", 0, 0)}
    for tag, (mode, cf, ca) in sums.items():
        R.set_arithmetic(mode)
        R.set_ref_sums(cf, ca)
        if mode != "product":
            v32 = torch.from_numpy(rng.standard_normal(N * N).astype(np.float32)).to(dev)
            u32 = torch.from_numpy(rng.standard_normal(na * N).astype(np.float32)).to(dev)
            ef = relerr(R.apply(v32).cpu().numpy(), Ro @ v32.cpu().numpy().astype(np.float64))
            ea = relerr(R.apply(u32, transpose=True).cpu().numpy(), Ro.T @ u32.cpu().numpy().astype(np.float64))
            print(f"# {tag:14s}: A noise {ef:.2e}   A^T noise {ea:.2e}")
        x, info = S.Hybrid_LSQR(R, b, its, 1e-2, xt)
        runs[tag] = info["xHistory"]
    R.set_arithmetic("product")
    R.set_ref_sums(0, 0)
    tags = list(runs)
    print("# iterate " + " ".join(f"{t:>13s}" for t in tags))
    d = {t: [relerr(h, ho) for h, ho in zip(runs[t], io["xHistory"])] for t in tags}
    for k in range(len(io["xHistory"])):
        print(f"{k + 1:9d} " + " ".join(f"{d[t][k]:13.3e}" for t in tags))
    print("# max over iterates 1..20 " + " ".join(f"{t}={max(d[t][:20]):.3e}" for t in tags))
    print("# max over iterates 21..  " + " ".join(f"{t}={max(d[t][20:]):.3e}" for t in tags))
    print("# final iterate           " + " ".join(f"{t}={d[t][-1]:.3e}" for t in tags))


if __name__ == "__main__":
    main()
