#!/usr/bin/env python
"""bench.py — Krylov iterations/second on the 4096 x 4096 blur, with the blur-matvec HBM roofline and a same-box
CPU baseline, as one JSON line (driver contract).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one Krylov iteration of the workload's solver on one batch of synthetic input (generated on the GPU,
resident in HBM before the timed region).  Static single-image problems do not shard (replicas only):
with N ranks every rank solves its own 4096^2 problem and `value` is the aggregate iterations/s ("weak").
The sharded dynamic-tomography configuration (frames over ranks, RCCL all-reduce of the inner products) is measured
in the same run and reported under "extra" (it is not `value`).

Only the `cpu_baseline` leg touches oracle/ (the float64 NumPy/SciPy restatement of the reference path, calling
scipy.ndimage.convolve exactly as Deblurring2D.py:70-71 does) — as the reported baseline, never as the thing measured.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=4096, help="image side of the blur workload")
    ap.add_argument("--workload", default="blur_cgls", choices=["blur_cgls"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3, help="CPU-baseline sample: CGLS iterations timed on the host")
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    return rank, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(v, world):
    if world == 1:
        return v
    import torch.distributed as dist
    t = torch.tensor([v], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class KernelTimer:
    """hipEvent pairs recorded by libtrk around the operator's main kernel (trk_timer_*, include/trk.h)."""

    def __init__(self, op, capacity, which):
        self.lib, self.op, self.cap = op.engine.lib, op, capacity
        self.h = ctypes.c_void_p()
        rc = self.lib.trk_timer_create(capacity, ctypes.byref(self.h))
        assert rc == 0, self.lib.trk_last_error()
        self.which = which

    def attach(self):
        self.lib.trk_timer_reset(self.h)
        assert self.lib.trk_op_set_timer(self.op._h, self.h, self.which) == 0

    def detach(self):
        self.lib.trk_op_set_timer(self.op._h, None, 0)

    def read(self):
        buf = (ctypes.c_float * self.cap)()
        n = ctypes.c_int()
        rc = self.lib.trk_timer_read(self.h, buf, self.cap, ctypes.byref(n))
        assert rc == 0, self.lib.trk_last_error()
        return np.array(buf[:n.value], dtype=np.float64)


def load_traffic(kernel_key):
    """HBM bytes per launch from the rocprofv3 PMC passes, if a summary was committed (profiles/traffic.json)."""
    p = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(kernel_key)
        except Exception:
            return None
    return None


def run_blur_cgls(args, rank, world):
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import CGLSRun

    N, K, W = args.size, args.steps, args.warmup
    n = N * N
    psf, _ = gauss_psf((9, 9), (3, 3))
    A = Blur2D(psf, N, N)
    eng = A.engine
    # synthetic, seeded, generated on the device: rectangles + texture; b = A x + 1% noise
    g = torch.Generator(device="cpu").manual_seed(rank)
    img = torch.zeros((N, N), dtype=torch.float32)
    rr = torch.randint(0, N - N // 8, (8, 2), generator=g)
    hw = torch.randint(N // 16, N // 3, (8, 2), generator=g)
    amp = torch.rand(8, generator=g) * 0.8 + 0.2
    for q in range(8):
        img[rr[q, 0]:rr[q, 0] + hw[q, 0], rr[q, 1]:rr[q, 1] + hw[q, 1]] += amp[q]
    x_true = img.reshape(-1).to(eng.device)
    x_true += 0.1 * torch.rand(n, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(100 + rank))
    b = A.apply(x_true)
    e = torch.randn(n, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(200 + rank))
    b = b + e * (0.01 * torch.linalg.norm(b) / torch.linalg.norm(e))
    x0 = torch.zeros(n, dtype=torch.float32, device=eng.device)

    run = CGLSRun(A, b, x0, W + K, x_true=None, history=False)    # reference call without x_true (CGLS.py:16)
    for _ in range(W):
        run.step()
    tfwd = KernelTimer(A, K + 4, 0)
    tfwd.attach()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(K):
        run.step()
    barrier(world)
    t1 = time.perf_counter()
    tfwd.detach()
    elapsed = max_over_ranks(t1 - t0, world)
    ms_fwd = tfwd.read()
    _g0, rows = run.rows()
    assert np.all(np.isfinite(rows)) and rows.shape[0] == W + K

    # transpose kernel, timed the same way in a short extra loop (outside the timed region)
    tadj = KernelTimer(A, 32, 1)
    tadj.attach()
    y = eng.empty(n)
    for _ in range(20):
        A.apply(b, out=y, transpose=True)
    torch.cuda.synchronize()
    tadj.detach()
    ms_adj = tadj.read()

    alg_bytes = 8.0 * n                                   # read x once + write y once
    t_kernel = float(np.mean(ms_fwd)) * 1e-3
    achieved = alg_bytes / t_kernel / 1e9
    roofline = {"bound": "hbm", "kernel": "k_blur_slide<9,9,D=6,sumsq> (forward blur matvec, fused ||Ap||^2)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": load_traffic("k_blur_slide_fwd"),
                "alg_bytes_per_launch": alg_bytes, "avg_kernel_us": round(t_kernel * 1e6, 2),
                "min_kernel_us": round(float(np.min(ms_fwd)) * 1e3, 2), "launches_timed": int(len(ms_fwd)),
                "adjoint_avg_kernel_us": round(float(np.mean(ms_adj[2:])) * 1e3, 2),
                "adjoint_GBps": round(alg_bytes / (float(np.mean(ms_adj[2:])) * 1e-3) / 1e9, 1)}

    res = {"metric": "krylov_iters_per_sec", "value": round(world * K / elapsed, 3), "unit": "iters/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"blur{N}_cgls", "image": f"{N}x{N} fp32", "psf": "Gaussian 9x9 sigma=(3,3), reflect",
                      "solver": "CGLS (trips.solvers.CGLS semantics, tol=0)", "noise": "1% Gaussian",
                      "parallelism": "replicas" if world > 1 else "single"},
           "roofline": roofline,
           "extra": {"relError_after_timed_iters": float(torch.linalg.norm(run.x_cur - x_true) / torch.linalg.norm(x_true)),
                     "cgls_alg_bytes_per_iter": 44.0 * n,
                     "cgls_effective_GBps": round(44.0 * n * K / elapsed / 1e9, 1)}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_cgls(psf, N, b, args.cpu_iters)
    return res


def cpu_baseline_cgls(psf, N, b_dev, iters):
    """The oracle CGLS (float64, scipy.ndimage.convolve like the reference) on the same b, on this box's host cores."""
    from oracle import cpu_ref as O
    Ao = O.Blur2D(psf, N, N)
    bh = b_dev.detach().to("cpu").numpy().astype(np.float64).reshape(-1, 1)
    # setup (r = b - A x0, t = A^T r) is outside the timed iterations, as on the GPU side
    stamps = []
    orig_fwd = Ao._fwd

    def fwd_stamped(x):
        stamps.append(time.perf_counter())
        return orig_fwd(x)

    Ao._fwd = fwd_stamped                      # one forward apply opens each CGLS iteration
    t_end = None
    x, info = O.cgls(Ao, bh, np.zeros((N * N, 1)), iters, 0)
    t_end = time.perf_counter()
    # stamps[0] is the setup apply (A x0); stamps[1..iters] open iterations 1..iters
    t_iter = (t_end - stamps[1]) / iters
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = None
    return {"value": round(1.0 / t_iter, 4), "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": f"{iters} CGLS iterations of the same {N}x{N} problem (float64 oracle; scipy.ndimage.convolve is "
                      f"single-threaded; NumPy BLAS threads={blas_threads}; host has {os.cpu_count()} logical cores)",
            "seconds_per_iter": round(t_iter, 3)}


def main():
    args = parse()
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the engine has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    rank, world = dist_setup(args)
    res = run_blur_cgls(args, rank, world)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
