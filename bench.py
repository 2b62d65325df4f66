#!/usr/bin/env python
"""bench.py — Krylov iterations/second on the 4096 x 4096 blur, with the blur-matvec HBM roofline and a same-box
CPU baseline, as one JSON line (driver contract).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Either a launcher started the ranks (`python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or — plain `python bench.py --gpus N`
— this process starts them itself as child processes of torch.distributed.run before anything has touched the GPU.

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

EXTRAS_BUDGET_S = 420
# The loop needs ~100 iterations (14 ms) from a cold start before its kernels run at their steady duration (tools/bench_ramp.py on
# the MI355X box: forward blur 25.1-25.7 us over iterations 0-50, 24.8 over 50-100, 24.1 from 100 on, whatever idle time precedes;
# profiles/r03/ramp_cold.txt).  `--steps 20 --warmup 5` alone therefore times the ramp, not the loop.  The run-in is untimed, stated
# in the JSON line ("run_in_iters"), and part of neither `steps` nor `warmup`.
RUN_IN_ITERS = 400
HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=4096, help="image side of the blur workload")
    ap.add_argument("--workload", default="blur_cgls", choices=["blur_cgls"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="force the generic (unfused-operand) CGLS iteration")
    ap.add_argument("--fused", action="store_true", help="force the fused three-launch CGLS iteration")
    ap.add_argument("--no-extras", action="store_true", help="skip the C4 (MMGKS) and C5 (sharded dynamic tomo) legs")
    ap.add_argument("--cpu-iters", type=int, default=8, help="CPU-baseline sample: CGLS iterations timed on the host")
    ap.add_argument("--run-in", type=int, default=RUN_IN_ITERS,
                    help="untimed iterations of the same loop BEFORE the --warmup ones (clock / cache settling; stated in the line)")
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching HIP / HSA (the parent must stay clean so that it can start its ranks as
    children): KFD topology nodes with SIMDs (/sys/class/kfd/kfd/topology/nodes/*/properties, `simd_count` > 0 = a GPU node;
    CPU nodes have 0), narrowed by a HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES list if one is set."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split(None, 1) for line in open(f).read().splitlines() if " " in line)
            n += int(props.get("simd_count", "0").strip()) > 0
        except OSError:
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([t for t in v.split(",") if t.strip() != ""])
            n = min(n, listed) if n else listed
    return n


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed launcher: start the N ranks as CHILD processes of
    torch.distributed.run and leave with their exit code.  Nothing in this process has touched the GPU yet (no
    torch.cuda call that initialises HIP, libtrk not loaded), and the launcher is a child, never an exec."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    single = bool(os.environ.get("TRK_SINGLE_DEVICE"))
    have = visible_gpu_count()                        # sysfs / environment only: no HIP or HSA call in this process
    if have < args.gpus and not single:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible "
              "(TRK_SINGLE_DEVICE=1 TRK_DIST_BACKEND=gloo runs all ranks on one GPU for debugging)", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("TRK_DIST_BACKEND", "nccl")          # "gloo": single-GPU debugging of the N > 1 path
        if os.environ.get("TRK_SINGLE_DEVICE"):
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    return rank, world


def comm_check(rank, world):
    """What the communicator really is: the rank count torch.distributed reports and one all-reduced double
    (sum of rank + 1 over ranks = world (world + 1) / 2) over the backend in use (nccl = RCCL over xGMI)."""
    if world == 1:
        return {"ranks_seen": 1, "backend": None, "allreduce_check": 1.0, "allreduce_expected": 1.0}
    import torch.distributed as dist
    nccl = dist.get_backend() == "nccl"
    t = torch.tensor([rank + 1.0], dtype=torch.float64, device="cuda" if nccl else "cpu")
    dist.all_reduce(t)
    out = {"ranks_seen": dist.get_world_size(), "backend": "rccl (torch nccl)" if nccl else dist.get_backend(),
           "allreduce_check": float(t.item()), "allreduce_expected": world * (world + 1) / 2.0}
    if nccl:                                           # latency of the one-double all-reduce the sharded solvers issue
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        out["allreduce_1double_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 2)
    return out


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(v, world):
    if world == 1:
        return v
    import torch.distributed as dist
    t = torch.tensor([v], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
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


def emit_json(fd, res):
    sys.stdout.flush()
    os.write(fd, (json.dumps(res) + "\n").encode())


def run_blur_cgls(args, rank, world, json_fd=1):
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import CGLSRun, CGLSRunFused

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

    # the class trips_py_amd.solvers.CGLS itself picks for this operator (tol = 0, single rank per problem)
    fused = CGLSRunFused.usable(A, eng) and not args.unfused and (args.fused or CGLSRunFused.auto(n))
    tiled = not args.unfused and not args.fused and CGLSRunFused.tiled_usable(A, eng)      # small images (--size <= 1024)
    tiled = CGLSRunFused.TILED_DEFAULT if tiled else 0      # the tiled form CGLS() picks (2: two blurs per iteration)
    Run = CGLSRunFused if (fused or tiled) else CGLSRun
    # reference call without x_true (CGLS.py:16); norms deferred exactly as CGLS() does for tol = 0 on one rank
    # run-in: a separate, untimed solve of the same problem with the same iteration form (see RUN_IN_ITERS)
    RI = max(0, args.run_in)
    if RI:
        rin = Run(A, b, x0, RI, x_true=None, history=False, tiled=tiled) if (fused or tiled) else \
            Run(A, b, x0, RI, x_true=None, history=False, defer_norms=True)
        rin.run(RI)
        torch.cuda.synchronize()
        del rin
    run = Run(A, b, x0, W + K, x_true=None, history=False, tiled=tiled) if (fused or tiled) else \
        Run(A, b, x0, W + K, x_true=None, history=False, defer_norms=True)
    run.run(W)
    tfwd = KernelTimer(A, K + 4, 0)
    tfwd.attach()
    barrier(world)
    t0 = time.perf_counter()
    run.run(K)                                  # K iterations, enqueued by one library call (trk_cgls_iterate*)
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

    # algorithmic bytes of the forward-blur launch: plain matvec reads x and writes y (8 n); the fused form also reads
    # p_old and writes p_new (16 n) — SURVEY §8d's own accounting of a fused update + matvec
    alg_bytes = (16.0 if fused else 8.0) * n
    if tiled:       # two tile kernels per iteration, no separate blur launch: the iteration's 44n bytes over its time
        alg_bytes, ms_fwd = 44.0 * n, np.array([elapsed / K * 1e3])
    t_kernel = float(np.mean(ms_fwd)) * 1e-3
    achieved = alg_bytes / t_kernel / 1e9
    kname = ("k_cgls_tile_a2 + k_cgls_tile_b2 (whole iteration: two launches, 44n algorithmic bytes)" if tiled == 2 else
             "k_cgls_tile_a + k_cgls_tile_b (whole iteration: two launches, 44n algorithmic bytes)" if tiled else
             "k_blur_slide<9,9,D=9,sumsq,fuse> (p = t + ratio*p fused into w = A p, + ||w||^2)" if fused
             else "k_blur_slide<9,9,D=9,sumsq> (forward blur matvec w = A p, fused ||w||^2)")
    roofline = {"bound": "hbm", "kernel": kname,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None if fused else load_traffic("k_blur_slide_fwd"),
                "alg_bytes_per_launch": alg_bytes, "avg_kernel_us": round(t_kernel * 1e6, 2),
                "min_kernel_us": round(float(np.min(ms_fwd)) * 1e3, 2),
                "median_kernel_us": round(float(np.median(ms_fwd)) * 1e3, 2),
                "p90_kernel_us": round(float(np.percentile(ms_fwd, 90)) * 1e3, 2),
                "max_kernel_us": round(float(np.max(ms_fwd)) * 1e3, 2), "launches_timed": int(len(ms_fwd)),
                "adjoint_matvec_avg_kernel_us": round(float(np.mean(ms_adj[2:])) * 1e3, 2),
                "adjoint_matvec_GBps": round(8.0 * n / (float(np.mean(ms_adj[2:])) * 1e-3) / 1e9, 1)}

    res = {"metric": "krylov_iters_per_sec", "value": round(world * K / elapsed, 3), "unit": "iters/s",
           "n_gpus": world, "steps": K, "warmup": W, "run_in_iters": RI, "ms_per_step": round(elapsed / K * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"blur{N}_cgls", "image": f"{N}x{N} fp32", "psf": "Gaussian 9x9 sigma=(3,3), reflect",
                      "solver": "CGLS (trips.solvers.CGLS semantics, tol=0)", "noise": "1% Gaussian",
                      "iteration": "tiled: 2 launches, 2 blurs (a workgroup per 32x32 tile; w = A p advanced as A t + beta w)" if tiled == 2 else
                                   "tiled: 2 launches (a workgroup per 32x32 tile recomputes its halo of p and A p in LDS)" if tiled else
                                   "fused: 3 launches (blur+p-update, x-update, blur^T+r-update)" if Run is CGLSRunFused else (("4 launches (blur, r update, blur^T, x/p update in one pass over p; consumers add the block partials)"
                                     if getattr(run, "grouping", 0) == 1 else
                                     "4 launches (blur, x/r update, blur^T, p update; consumers add the block partials)")
                                    if getattr(run, "raw", False) else "generic: 6 launches"),
                      "parallelism": (f"replicas: `value` is {world} independent {N}x{N} solves, one per GPU, no collective on the data path "
                                      "(a static single-image problem does not shard: DESIGN.md 5) — the SHARDED workload of north_star "
                                      "(C5, frames over ranks, RCCL all-reduce per iteration) is the top-level `sharded` object") if world > 1 else "single"},
           "roofline": roofline,
           "extra": {"comm": comm_check(rank, world),
                     "relError_after_timed_iters": float(torch.linalg.norm(run.x_cur - x_true) / torch.linalg.norm(x_true)),
                     "cgls_alg_bytes_per_iter": 44.0 * n,
                     "cgls_effective_GBps": round(44.0 * n * K / elapsed / 1e9, 1)}}

    def solves_leg():
        """solver rates "history off and on", and what a reference-style call sees from idle.  Whole CGLS() calls of
        100 iterations (CGLS.py:16 semantics, tol = 0, device tensors in and out), wall clock around the call, synchronised both ends:
        history off / on (the reference keeps every iterate, CGLS.py:66: here 100 rows of 67 MB on the device), after a warm call;
        cold: ONE call after two seconds of an idle GPU, no run-in — the ~100-iteration clock / cache ramp of RUN_IN_ITERS is inside.
        Runs LAST of the GPU legs: the idle period it needs costs whatever is measured next its clocks (C2 at 512^2 read 53 k
        instead of 70 k iterations/s right behind it)."""
        try:
            from trips_py_amd.solvers import CGLS as _CGLS
            solves = {}
            for tag, hist in (("history_off", False), ("history_on", True)):
                _CGLS(A, b, x0, 100, 0, history=hist)
                torch.cuda.synchronize()
                barrier(world)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    _CGLS(A, b, x0, 100, 0, history=hist)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                solves[f"{tag}_iters_per_sec"] = round(world * 100 / max_over_ranks(float(np.median(ts)), world), 1)
                torch.cuda.empty_cache()
            time.sleep(2.0)
            barrier(world)
            t0 = time.perf_counter()
            _CGLS(A, b, x0, 100, 0, history=False)
            torch.cuda.synchronize()
            solves["cold_100_iter_solve_iters_per_sec"] = round(world * 100 / max_over_ranks(time.perf_counter() - t0, world), 1)
            solves["note"] = "whole CGLS() calls of 100 iterations at this size (constructor, loop, final norms, result), median of 3; cold: one call after 2 s idle"
            return solves
        except Exception as exc:          # noqa: BLE001
            return {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # Host baselines run AFTER every GPU measurement of this process: NumPy / SciPy work on the host wakes BLAS thread pools whose
    # idle spinning can exhaust the container's CPU quota, and the kernel then parks every thread for the rest of the period — a
    # solver loop enqueueing microsecond kernels stops for ~100 ms (seen here as C4 at 170 instead of 540
    # iterations/s when the C3 host leg ran right before it).
    cpu_jobs = []          # (where the result goes, closure)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        b_host = b.detach().to("cpu")
        cpu_jobs.append((lambda v: res.__setitem__("cpu_baseline", v), lambda: cpu_baseline_cgls(psf, N, b_host, args.cpu_iters)))
    del run
    torch.cuda.empty_cache()
    if not args.no_extras:
        # secondary measurements (never `value`): each guarded so that the main line is always printed — against
        # exceptions by try/except, against a hang (e.g. one rank failing inside a collective) by a watchdog that prints
        # the line with whatever is measured so far and ends the process
        import threading

        def give_up():
            res["extra"]["extras_watchdog"] = f"secondary measurements exceeded {EXTRAS_BUDGET_S} s; abandoned"
            if rank == 0:
                emit_json(json_fd, res)
            os._exit(3)          # the line is printed, but an abandoned run must not look like a finished one

        watchdog = threading.Timer(EXTRAS_BUDGET_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        cpu = rank == 0 and world == 1 and not args.no_cpu_baseline      # host baselines beside every config (rank 0, N = 1)
        for name, fn in (("trk_comm_rccl", lambda: extra_trk_comm(rank, world)),
                         ("c2_blur512_cgls", lambda: extra_c2_blur512(world, cpu_jobs if cpu else None)),
                         ("c3_tomo512_hybrid_lsqr", lambda: extra_c3_tomo(world, cpu_jobs if cpu else None)),
                         ("c4_mmgks_tv_4096", lambda: extra_c4_mmgks(A, b, N, world, cpu_jobs if cpu else None, psf)),
                         ("c5_dynamic_tomo_sharded", lambda: extra_c5_dynamic(rank, world, cpu_jobs if cpu else None)),
                         ("hybrid_gmres_and_lsqr_blur512", lambda: extra_hybrid_blur(world)),
                         ("next_fanbeam512_matvec", lambda: extra_fanbeam(world)),
                         ("next_sparse_dynamic", lambda: extra_sparse_dynamic(world, cpu_jobs if cpu else None))):
            try:
                res["extra"][name] = fn()
            except Exception as exc:          # noqa: BLE001
                res["extra"][name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            barrier(world)
        res["extra"]["cgls_100_iter_solves"] = solves_leg()
        # what a trips.solvers.CGLS-style caller sees, next to `value` (which has a run-in in front of it)
        res["cold_solve_iters_per_sec"] = res["extra"]["cgls_100_iter_solves"].get("cold_100_iter_solve_iters_per_sec")
        res["warm_solve_iters_per_sec"] = res["extra"]["cgls_100_iter_solves"].get("history_off_iters_per_sec")
        c5 = res["extra"].get("c5_dynamic_tomo_sharded")
        if isinstance(c5, dict) and "error" not in c5:
            # the curve north_star asks for at 1/2/4/8 GPUs: C5 at its BASELINE size, frames over ranks (strong scaling)
            res["sharded"] = {"workload": "c5_dynamic_tomo (32 frames x 256^2 x 15 angles per frame, frames over ranks)",
                              "cgls_iters_per_sec": c5.get("cgls_iters_per_sec"), "gks_iters_per_sec": c5.get("gks_iters_per_sec"),
                              "ranks": world, "scaling": "strong", "communicator": c5.get("communicator"),
                              "cgls_reduction_points_per_iteration": c5.get("cgls_one_reduction_reduction_points_per_iteration"),
                              "gks_reduction_points_per_iteration": c5.get("gks_reduction_points_per_iteration")}
        for put, job in cpu_jobs:             # every GPU number is in: now the host legs, each guarded
            put(guarded(job))
        watchdog.cancel()
    else:
        for put, job in cpu_jobs:
            put(guarded(job))
    return res


def extra_trk_comm(rank, world):
    """libtrk.so's own collectives (trk_comm_init / trk_allreduce_f64 / trk_halo_exchange over RCCL, include/trk.h) next to
    torch's: sum of rank + 1 over the ranks and a ring shift of one frame-sized buffer."""
    import torch.distributed as dist
    if world > 1 and dist.get_backend() != "nccl":
        return {"skipped": "ranks share one GPU (gloo debugging run): RCCL needs one GPU per rank"}
    from trips_py_amd.dist import RcclComm
    c = RcclComm(rank, world)
    t = torch.tensor([rank + 1.0], dtype=torch.float64, device="cuda")
    c.allreduce_sum_(t)
    total = float(t.item())
    n = 256 * 256
    a = torch.full((n,), float(rank), device="cuda")
    b = torch.full((n,), -1.0, device="cuda")
    c.shift(a, (rank + 1) % world, b, (rank - 1) % world)
    torch.cuda.synchronize()
    ok = bool(torch.all(b == float((rank - 1) % world)).item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        c.allreduce_sum_(t)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 100 * 1e6
    return {"allreduce_sum": total, "allreduce_expected": world * (world + 1) / 2.0, "halo_shift_ok": ok,
            "allreduce_1double_us": round(us, 2), "ranks": world,
            "note": ("one rank: trk_allreduce_f64 is a no-op by contract; this probe runs with TRK_COMM_FORCE=1 so that the call "
                     "goes through RCCL all the same" if world == 1 else "trk_allreduce_f64 over RCCL, enqueue + device time per call")}


def guarded(fn):
    """A host baseline must never cost the GPU numbers their line."""
    try:
        return fn()
    except Exception as exc:      # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


class no_gc:
    """Timed regions of a few milliseconds run with the cyclic garbage collector off, as `timeit` does: a generation-2 pass over
    a process that holds a few thousand tensors is itself milliseconds."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


def extra_c2_blur512(world, cpu_jobs=None):
    """BASELINE config C2: 2-D Gaussian blur 512^2 fp32, CGLS 100 iterations (the reference's own demo size; BASELINE.md §2
    measured the reference at 21.7 it/s on this problem).  Whole solves through the public CGLS() call, x_true given as in
    the demo (relError history on).  Replicas across ranks."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import CGLS
    N = 512
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    b = A.apply(xt)
    e = torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    b = b + e * (0.01 * torch.linalg.norm(b) / torch.linalg.norm(e))
    x0 = torch.zeros(N * N, device=dev)
    CGLS(A, b, x0, 100, 0, x_true=xt, history=False)
    barrier(world)
    reps = 5
    with no_gc():
        t0 = time.perf_counter()
        for _ in range(reps):
            x, info = CGLS(A, b, x0, 100, 0, x_true=xt, history=False)
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world)
    out = {"solver": "CGLS(max_iter=100, tol=0, x_true)", "iters_per_sec_all_ranks": round(world * reps * 100 / dt, 1),
           "ms_per_solve": round(dt / reps * 1e3, 3), "relError_last": float(info["relError"][-1])}
    if cpu_jobs is not None:
        bh = b.detach().to("cpu")
        cpu_jobs.append((lambda v: out.__setitem__("cpu_baseline", v), lambda: cpu_c2(gauss_psf((9, 9), (3, 3))[0], N, bh)))
    return out


def extra_c3_tomo(world, cpu_jobs=None):
    """BASELINE config C3: parallel-beam tomography 512^2, 180 angles, Hybrid-LSQR 100 iterations (lambda = 1e-2), plus
    the Radon matvec rates.  The Radon operator is gather/ALU-bound, not HBM-bound: taps/s is the honest
    rate, algorithmic GB/s is reported for completeness.  Replicas across ranks."""
    from trips_py_amd.operators import Radon2DParallel
    from trips_py_amd.solvers import Hybrid_LSQR
    Nt, na = 512, 180
    R = Radon2DParallel(Nt, np.linspace(0, np.pi, na, endpoint=False))
    eng = R.engine
    ii, jj = torch.meshgrid(torch.arange(Nt), torch.arange(Nt), indexing="ij")
    ph = ((((ii - 256) / 180.0) ** 2 + ((jj - 256) / 230.0) ** 2) < 1).float() + 0.5 * ((((ii - 300) / 60.0) ** 2 + ((jj - 200) / 40.0) ** 2) < 1).float()
    xt = ph.reshape(-1).to(eng.device)
    bt = R.apply(xt)
    e = torch.randn(bt.numel(), device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(5))
    delta = 0.01 * float(torch.linalg.norm(bt))          # ||noise||, handed to the discrepancy principle
    bt = bt + e * (delta / torch.linalg.norm(e))
    out = {"geometry": f"{Nt}x{Nt}, {na} angles, {Nt} detectors", "taps_per_apply": 2.0 * Nt * Nt * na,
           "alg_bytes_per_apply": 4.0 * (Nt * Nt + na * Nt)}
    y, z = torch.empty_like(bt), torch.empty_like(xt)
    for name, fn in (("fwd", lambda: R.apply(xt, out=y)), ("adj", lambda: R.apply(bt, out=z, transpose=True))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out[f"radon_{name}_us"] = round(ms * 1e3, 1)
        out[f"radon_{name}_Gtaps_per_s"] = round(out["taps_per_apply"] / ms / 1e6, 1)
        out[f"radon_{name}_alg_GBps"] = round(out["alg_bytes_per_apply"] / ms / 1e6, 2)
    # the north_star's second roofline point: the Radon matvec at 4096^2 x 180 angles (6.04 G taps per apply)
    try:
        Nb = 4096
        Rb = Radon2DParallel(Nb, np.linspace(0, np.pi, na, endpoint=False))
        xb = torch.rand(Nb * Nb, device=eng.device)
        yb, zb = torch.empty(Rb.shape[0], device=eng.device), torch.empty(Nb * Nb, device=eng.device)
        big = {"taps_per_apply": 2.0 * Nb * Nb * na, "alg_bytes_per_apply": 4.0 * (Nb * Nb + na * Nb)}
        for name, fn in (("fwd", lambda: Rb.apply(xb, out=yb)), ("adj", lambda: Rb.apply(yb, out=zb, transpose=True))):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            big[f"{name}_ms"] = round(ms, 3)
            big[f"{name}_Gtaps_per_s"] = round(big["taps_per_apply"] / ms / 1e6, 1)
            big[f"{name}_alg_GBps"] = round(big["alg_bytes_per_apply"] / ms / 1e6, 2)
        # Neither direction is bound by HBM.  What a gfx950 SIMD and the LDS sustain for these instruction streams was
        # measured on the chip (tools/microbench/issue_rate.hip, profiles/r04/issue_rate.txt): 2.3 cycles per wave64 instruction for
        # two-source VGPR forms, 4.1-4.2 for an SGPR operand, v_cvt_f32_u32, three-source integer forms and v_pk_fma_f32;
        # ds_read2_b32 4 LDS cycles per wave when no two lanes of a 32-lane group meet on a bank.
        #   forward (k_radon_fwd_quad): four symmetric angles share the taps' arithmetic, so a wave-step serves 256
        #   ray-steps with 4 ds_read2_b32 (16 LDS cycles per CU: the binding pipe, conflict-free by construction) and
        #   8 shared + 4 packed-FMA vector instructions (38.5 cycles on one of the CU's four SIMDs);
        #   adjoint (round 6, k_radon_adj_quad: mirrored tile pairs share the base geometry's weights): 6 vector instructions per
        #   pixel and angle at 4 cycles each (k_radon_adj_tile: 10.25), one ds_read_b128 per pixel and angle (4 LDS cycles per wave).
        # `fwd_ms` / `adj_ms` are whole applies (forward: transpose + clear + k_radon_fwd_quadf + band reduction; adjoint: record
        # pre-pass + k_radon_adj_quad); the kernels alone are in profiles/r06 (rocprofv3).
        steps = float(Nb) * Nb * na
        lds_floor_ms = steps / 256.0 * 16.0 / (256 * 2.4e9) * 1e3
        valu_floor_ms = steps / 256.0 * 38.5 / (1024 * 2.4e9) * 1e3
        big["fwd_roofline"] = {"bound": "lds_gather", "kernel": "k_radon_fwd_quadf (+ k_transpose, clear, k_radon_bands_post in fwd_ms)",
                               "lds_cycles_per_256_ray_steps": 16.0, "units": steps, "floor_ms": round(lds_floor_ms, 4),
                               "frac": round(lds_floor_ms / big["fwd_ms"], 4), "valu_floor_ms": round(valu_floor_ms, 4),
                               "valu_cycles_per_256_ray_steps_per_simd": 38.5}
        floor_ms = steps * 6.0 / 64.0 * 4.0 / (1024 * 2.4e9) * 1e3
        adj_lds_floor_ms = steps / 64.0 * 4.0 / (256 * 2.4e9) * 1e3
        big["adj_roofline"] = {"bound": "valu_issue", "kernel": "k_radon_adj_quad (+ k_radon_adj_prepq in adj_ms)", "instr_per_unit": 6.0,
                               "cycles_per_instr": 4.0, "units": steps, "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / big["adj_ms"], 4),
                               "lds_floor_ms": round(adj_lds_floor_ms, 4), "round5_kernel_floor_ms": round(steps * 10.25 / 64.0 * 4.0 / (1024 * 2.4e9) * 1e3, 4)}
        out["radon_4096x180"] = big
        del Rb, xb, yb, zb
    except Exception as exc:      # noqa: BLE001
        out["radon_4096x180"] = {"error": str(exc)[:200]}
    # every iterate is formed and its relError evaluated (x_true given, as the reference's demos do); only the
    # list of host copies of the iterates is skipped (history=False)
    for tag, reg, kw in (("", 1e-2, {}), ("_gcv", "gcv", {}), ("_dp", "dp", {"delta": delta})):
        # warm-up: one whole solve of the timed size (as C2 does) — a 100-step solve allocates its two bases (150 MB) the first
        # time, which a 5-step warm-up left inside the timed region (13.8k against 15.0k it/s from the third solve on)
        for _ in range(2):
            Hybrid_LSQR(R, bt, 100, reg, x_true=xt, history=False, **kw)
        barrier(world)
        reps = 5
        each = []
        with no_gc():
            t0 = time.perf_counter()
            for _ in range(reps):
                t1 = time.perf_counter()
                _, info = Hybrid_LSQR(R, bt, 100, reg, x_true=xt, history=False, **kw)
                each.append(time.perf_counter() - t1)
            barrier(world)
            dt = max_over_ranks(time.perf_counter() - t0, world)
        out[f"hybrid_lsqr{tag}_iters_per_sec_all_ranks"] = round(world * reps * 100 / dt, 1)
        # the automatic selectors search on a host thread: one solve in eight is a third slower when the host stalls
        # (tools/hybrid_selector_rates.py); the median solve beside the mean
        out[f"hybrid_lsqr{tag}_median_solve_iters_per_sec"] = round(world * 100 / max_over_ranks(float(np.median(each)), world), 1)
        out[f"hybrid_lsqr{tag}_relError_last"] = float(info["relError"][-1])
    # the float64 INSTRUMENT beside the product (Hybrid_LSQR(..., dtype='float64'): csrc/ref64.hip — one thread per ray / per pixel,
    # float64 arithmetic and storage; what the fast path is checked against on the hardware, not a fast path itself)
    try:
        Hybrid_LSQR(R, bt, 10, 1e-2, dtype="float64")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Hybrid_LSQR(R, bt, 30, 1e-2, dtype="float64")
        torch.cuda.synchronize()
        out["hybrid_lsqr_float64_instrument_iters_per_sec"] = round(30 / (time.perf_counter() - t0), 1)
    except Exception as exc:      # noqa: BLE001
        out["hybrid_lsqr_float64_instrument_iters_per_sec"] = f"error: {exc}"[:200]
    if cpu_jobs is not None:
        bh = bt.detach().to("cpu")
        cpu_jobs.append((lambda v: out.__setitem__("cpu_baseline", v),
                         lambda: cpu_c3(Nt, np.linspace(0, np.pi, na, endpoint=False), bh)))
    return out


def extra_hybrid_blur(world):
    """The two hybrid loops of the path on the reference's own deblurring size (512^2, 9 x 9 Gaussian PSF, 1 % noise): Hybrid-GMRES
    (Hybrid_GMRES.py: Arnoldi + projected Tikhonov; no BASELINE config of its own) and Hybrid-LSQR on the same operator, 60-iteration
    solves with x_true, fixed lambda and GCV.  Replicas across ranks."""
    from trips_py_amd.operators import Blur2D
    from trips_py_amd.problems import gauss_psf
    from trips_py_amd.solvers import Hybrid_GMRES, Hybrid_LSQR
    N = 512
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    dev = A.engine.device
    x = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    b = A.apply(x)
    e = torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    delta = 0.01 * float(b.norm())
    b = b + e * (delta / e.norm())
    out = {"problem": "512x512 Gaussian blur 9x9 sigma 3, 1% noise, 60 iterations per solve, x_true given"}
    for name, solver in (("hybrid_gmres", Hybrid_GMRES), ("hybrid_lsqr", Hybrid_LSQR)):
        for reg in (1e-2, "gcv", "dp"):
            kw = {"delta": delta} if reg == "dp" else {}
            solver(A, b, 5, reg, x, **kw)
            solver(A, b, 60, reg, x, **kw)
            torch.cuda.synchronize()
            barrier(world)
            t0 = time.perf_counter()
            for _ in range(3):
                solver(A, b, 60, reg, x, **kw)
            torch.cuda.synchronize()
            barrier(world)
            dt = max_over_ranks(time.perf_counter() - t0, world) / 3
            out[f"{name}_{'fixed_lambda' if not isinstance(reg, str) else reg}_iters_per_sec_all_ranks"] = round(world * 60 / dt, 1)
    return out


def extra_fanbeam(world):
    """SURVEY 8f "next" row, rank 1 (not a BASELINE config): the fan-beam 'line_fanflat' projector of Tomography.py:53-88 at the demo
    geometry scaled to 512^2 (180 views, int(sqrt(2) N) = 724 detectors): time per forward / adjoint apply, ray-steps per second."""
    from trips_py_amd.operators import FanBeam2D
    N, views = 512, 180
    R = FanBeam2D(N, views=views)
    dev = R.engine.device
    x = torch.rand(N * N, device=dev)
    y, z = torch.empty(R.shape[0], device=dev), torch.empty(N * N, device=dev)
    out = {"geometry": f"{N}x{N}, {views} views, {R.n_det} detectors, source 3N / detector N from the centre", "ray_steps_per_apply": float(views) * R.n_det * N}
    for name, fn in (("fwd", lambda: R.apply(x, out=y)), ("adj", lambda: R.apply(y, out=z, transpose=True))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out[f"{name}_us"] = round(us, 1)
        out[f"{name}_Gsteps_per_s"] = round(out["ray_steps_per_apply"] / us * 1e-3, 1)
    out["bound"] = "forward: texture addresser (one scattered 8-byte gather per ray-step: 16.6 cycles per wave-load); adjoint (round 6, k_fan_adj_views: 42 vector instructions per pixel and view, 59 before): the texture data path at 82 % (two scattered 16-byte records per pixel and view) next to the vector unit at 76 % (profiles/r06/fanbeam_adj_pmc.txt)"
    return out


def joseph_block_matrix(Nf, angle_sets, n_det):
    """blkdiag over frames of the parallel-beam Joseph matrix (two linear-interpolation taps per marching step, scale 1 / Nf: the
    convention of trips_py_amd.operators.Radon2DParallel) as scipy.sparse CSR — the build's own stand-in for the sparse forward
    matrix the reference's real-data loaders read from disk (io.py:197-229; the Zenodo files are not reachable offline)."""
    import scipy.sparse as sp
    half, sdh = 0.5 * (Nf - 1), 0.5 * (n_det - 1)
    k = np.arange(Nf)
    s = np.arange(n_det) - sdh
    blocks = []
    for angles in angle_sets:
        rows, cols, vals = [], [], []
        for a, th in enumerate(angles):
            ct, st = np.cos(th), np.sin(th)
            if abs(ct) >= abs(st):
                q = (s[:, None] - (half - k)[None, :] * st) / ct + half
                w = 1.0 / abs(ct)
            else:
                q = half - (s[:, None] - (k - half)[None, :] * ct) / st
                w = 1.0 / abs(st)
            q0 = np.floor(q)
            f = q - q0
            for off, wt in ((0, 1.0 - f), (1, f)):
                t = (q0 + off).astype(np.int64)
                ok = (t >= 0) & (t < Nf) & (wt > 0)
                lin = (k[None, :] * Nf + t) if abs(ct) >= abs(st) else (t * Nf + k[None, :])
                d_idx = np.broadcast_to(np.arange(n_det)[:, None], t.shape)
                rows.append((a * n_det + d_idx)[ok])
                cols.append(lin[ok])
                vals.append((w * wt / Nf)[ok])
        blocks.append(sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(angles) * n_det, Nf * Nf)))
    return blocks


def extra_sparse_dynamic(world, cpu_jobs=None):
    """SURVEY 8f "next" row, rank 4 (not a BASELINE config): the sparse-forward-matrix dynamic path of io.py:197-229 — frames cut out
    of one sparse matrix, applied as ONE CSR operator (csrc/spmv.hip).  (i) CrossPhantom-like: 16 frames of 128^2, 700 rays per frame
    (demos/2_demo_dynamic_CrossPhantom.ipynb): Hybrid-LSQR iterations/s.  (ii) the SpMV's roofline on a ~2e7-non-zero block-diagonal
    matrix (16 frames of 256^2, 10 angles x 256 detectors): algorithmic bytes 8 nnz + 4 (m + n) per apply against the HBM peak."""
    from trips_py_amd.operators import SparseBlockDiag
    from trips_py_amd.solvers import Hybrid_LSQR
    out = {}
    # (ii) first: the big matrix is freed before the solver leg
    T, Nf, na, nd = 16, 256, 10, 256
    blocks = joseph_block_matrix(Nf, [np.deg2rad(t + 18.0 * np.arange(na)) for t in range(T)], nd)
    D = SparseBlockDiag(blocks)
    dev = D.engine.device
    m, n, nnz = D.shape[0], D.shape[1], int(D.matrix.nnz)
    x = torch.rand(n, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    y, z = torch.empty(m, device=dev), torch.empty(n, device=dev)
    alg = 8.0 * nnz + 4.0 * (m + n)
    roof = {"matrix": f"blkdiag of {T} frames, {m} x {n}, {nnz} non-zeros (Joseph weights, {na} angles x {nd} detectors per {Nf}x{Nf} frame)",
            "alg_bytes_per_apply": alg, "alg_bytes_formula": "8 nnz + 4 (m + n)", "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    # COLD (round 6): three handles holding copies of the matrix and their own vectors take turns, so that every
    # apply streams its 170 MB from HBM (3 x 170 MB per direction do not fit the 256 MB memory-side cache); `*_frac` is this
    # number.  WARM (one handle, back to back: round 5's figure, the matrix served by the memory-side cache) stays beside it.
    Ds = [D, SparseBlockDiag(blocks), SparseBlockDiag(blocks)]
    xs = [x] + [x.clone() for _ in range(2)]
    ys = [y] + [torch.empty_like(y) for _ in range(2)]
    zs = [z] + [torch.empty_like(z) for _ in range(2)]
    for name in ("fwd", "adj"):
        for mode, nh in (("", 3), ("_warm", 1)):
            def fn(i, name=name, nh=nh):
                h = i % nh
                if name == "fwd":
                    Ds[h].apply(xs[h], out=ys[h])
                else:
                    Ds[h].apply(ys[h], out=zs[h], transpose=True)
            for i in range(6):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            roof[f"{name}_us{mode}"] = round(us, 1)
            roof[f"{name}_achieved{mode}"] = round(alg / us * 1e-3, 1)
            roof[f"{name}_frac{mode}"] = round(alg / us * 1e-3 / HBM_PEAK_GBPS, 4)
    # A @ V on an (n, k) block (GKS.py:37 / MMGKS.py:44): k = 8 columns in ONE pass over the matrix (k_csr_group<G, 8>) against 8 passes
    try:
        k = 8
        X = torch.rand(k, n, device=dev, generator=torch.Generator(device=dev).manual_seed(13))
        Y = torch.empty(k, m, device=dev)
        for tag, call in (("block8", lambda h: Ds[h].apply(X, out=Y)),
                          ("eight_single", lambda h: [Ds[h].apply(X[j], out=Y[j]) for j in range(k)])):
            for i in range(3):
                call(i % 3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                call(i % 3)
            e1.record()
            torch.cuda.synchronize()
            roof[f"fwd_{tag}_us"] = round(e0.elapsed_time(e1) / 12 * 1e3, 1)
        roof["fwd_block8_alg_bytes"] = 8.0 * nnz + 4.0 * k * (m + n)
        roof["fwd_block8_achieved"] = round(roof["fwd_block8_alg_bytes"] / roof["fwd_block8_us"] * 1e-3, 1)
        del X, Y
    except Exception as exc:      # noqa: BLE001
        roof["fwd_block8_error"] = f"{type(exc).__name__}: {exc}"[:200]
    roof["note"] = ("fwd / adj: COLD — three copies of the 170 MB matrix (and of the vectors) in rotation, every apply streams from HBM; "
                    "*_warm: one copy back to back (it fits the 256 MB memory-side cache: round 5's figure); timed by events on the stream "
                    "the applies are enqueued on")
    del Ds, xs, ys, zs
    out["roofline_spmv"] = roof
    del D, x, y, z
    torch.cuda.empty_cache()
    # (i) the CrossPhantom-like problem through Hybrid-LSQR (the demo's cell 15), fixed lambda, whole solves
    T, Nf, na, nd = 16, 128, 5, 140
    blocks = joseph_block_matrix(Nf, [np.deg2rad(t + 36.0 * np.arange(na)) for t in range(T)], nd)
    F = SparseBlockDiag(blocks)
    xt = torch.zeros(T, Nf, Nf, device=dev)
    for t in range(T):
        xt[t, 30 + 2 * t:70 + 2 * t, 20:90] = 1.0
        xt[t, 80:110, 10 + 4 * t:40 + 4 * t] = 0.5
    xt = xt.reshape(-1)
    b = F.apply(xt)
    e = torch.randn(b.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(12))
    b = b + e * (0.01 * torch.linalg.norm(b) / torch.linalg.norm(e))
    its = 50
    for _ in range(2):                       # (as C3: the first solve allocates, the second still pays first-use costs of the host side)
        Hybrid_LSQR(F, b, its, 1e-2, xt, history=False)
    barrier(world)
    reps = 10
    each = []
    with no_gc():
        t0 = time.perf_counter()
        for _ in range(reps):
            t1 = time.perf_counter()
            xs, info = Hybrid_LSQR(F, b, its, 1e-2, xt, history=False)
            each.append(time.perf_counter() - t1)
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world)
    out.update({"problem": f"{T} frames of {Nf}x{Nf}, {na * nd} rays per frame, {int(F.matrix.nnz)} non-zeros in one CSR handle",
                "solver": f"Hybrid_LSQR(n_iter={its}, regparam=1e-2, x_true)", "iters_per_sec_all_ranks": round(world * reps * its / dt, 1),
                "median_solve_iters_per_sec": round(world * its / max_over_ranks(float(np.median(each)), world), 1),
                "ms_per_solve": round(dt / reps * 1e3, 3), "relError_last": float(info["relError"][-1])})
    if cpu_jobs is not None:
        Fm, bh = F.matrix.astype(np.float64), b.detach().to("cpu")
        cpu_jobs.append((lambda v: out.__setitem__("cpu_baseline", v), lambda: cpu_sparse_dynamic(Fm, bh)))
    return out


def extra_c4_mmgks(A, b, N, world, cpu_jobs=None, psf=None):
    """BASELINE config C4: blur 4096^2, MMGKS + TV (pnorm=2, qnorm=1, projection_dim=3, n_iter=30, lambda=1e-2).
    Replicas across ranks (a static image does not shard)."""
    from trips_py_amd.operators import FirstDerivative2D
    from trips_py_amd.solvers import MMGKS
    L = FirstDerivative2D(N)
    MMGKS(A, b, L, 2, 1, 3, 4, 1e-2, history=False)          # warm-up (allocations, kernels)
    MMGKS(A, b, L, 2, 1, 3, 30, 1e-2, history=False)         # and one whole solve of the timed size, as C2 / C3 / C5
    barrier(world)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        x, info = MMGKS(A, b, L, 2, 1, 3, 30, 1e-2, history=False)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world) / reps          # mean over three solves
    # algorithmic bytes of one MMGKS iteration at basis size k (fp32 vectors, m = n, p = 2n rows of L), DESIGN.md §4.2b:
    #   weighted Grams of AV and LV 4k(m + p) = 12kn | x = V y 4kn | the two re-orthogonalisation sweeps as ONE pair of passes
    #   (V^T r with the Gram row riding along, r - V c) 8kn | weights, stencil applies, weighted residuals, axpys on single
    #   vectors 192n        ->  (24k + 192) n  bytes.   (The reference's own sequence of sweeps — two passes each — is 32k.)
    # What this implementation MOVES per iteration is less (round 3): the fidelity Gram is unweighted for pnorm = 2 and only grows,
    # the re-weighted Gram of L V is formed from V itself (trk_wgram_tv), and its sweep over V also takes the fidelity Gram's new row
    # (trk_wgram_tv_z): ONE pass of 4kn where the count above has 12kn, and L v_j is neither written nor read ->  (16k + 180) n;
    # late round 4: v = r/||r|| with its dot against A^T b in one pass (-4n) and A x - b formed in the blur's store (-8n) ->  (16k + 168) n.
    n = N * N
    its = int(info["its"]) + 1
    alg = sum((24.0 * (3 + i) + 192.0) * n for i in range(its))
    alg_ref = sum((32.0 * (3 + i) + 192.0) * n for i in range(its))
    moved = sum((16.0 * (3 + i) + 168.0) * n for i in range(its))
    gbps = alg / dt / 1e9
    out = {"solver": "MMGKS(pnorm=2,qnorm=1,projection_dim=3,n_iter=30,regparam=1e-2,epsilon=0.1), L = 2-D first derivative",
            "iters_per_sec_all_ranks": round(world * 30 / dt, 2), "seconds_per_solve": round(dt, 4), "its": its,
            "parallelism": "replicas" if world > 1 else "single",
            # frac = bytes this implementation MOVES / time / peak (a count of what another algorithm would have
            # moved is not a roofline fraction); the two other counts stay as named extras
            "roofline": {"bound": "hbm", "bytes_per_iter_formula": "(16 k + 168) n moved, k = 3 + iteration index, n = 4096^2: the two Grams and the new Gram row share one sweep over V, L V is not stored, A x - b leaves the blur, the new vector is scaled and dotted in one pass",
                         "bytes_per_solve": moved, "achieved": round(moved / dt / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(moved / dt / 1e9 / HBM_PEAK_GBPS, 4),
                         "other_counts": {"fused_separate_grams_(24k+192)n_GBps": round(gbps, 1),
                                          "fused_separate_grams_frac": round(gbps / HBM_PEAK_GBPS, 4),
                                          "reference_sweep_passes_(32k+192)n_frac": round(alg_ref / dt / 1e9 / HBM_PEAK_GBPS, 4)},
                         "timed": "whole solve, wall clock incl. the host's projected problems"}}
    # the reference's DEFAULT regparam ('gcv': MMGKS.py:24): lambda chosen on the host every iteration (trk_host_gram_gcv)
    MMGKS(A, b, L, 2, 1, 3, 30, "gcv", history=False)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(reps):
        MMGKS(A, b, L, 2, 1, 3, 30, "gcv", history=False)
    barrier(world)
    out["gcv_iters_per_sec_all_ranks"] = round(world * 30 / (max_over_ranks(time.perf_counter() - t0, world) / reps), 2)
    # GKS on the same problem (GKS.py: the unweighted sibling of this solver): two passes over the basis per iteration (the new vector and
    # the next iterate share one, trk_gemv_orth_iterate) against the pass each of the reference's order, timed side by side
    try:
        from trips_py_amd.solvers import GKS
        gk = {}
        for key, kw in (("gks_iters_per_sec_all_ranks", {}), ("gks_a_pass_each_iters_per_sec_all_ranks", {"fused_orth_iterate": False})):
            GKS(A, b, L, 3, 30, 1e-2, history=False, **kw)
            barrier(world)
            t0 = time.perf_counter()
            for _ in range(reps):
                GKS(A, b, L, 3, 30, 1e-2, history=False, **kw)
            barrier(world)
            gk[key] = round(world * 30 / (max_over_ranks(time.perf_counter() - t0, world) / reps), 2)
        gk["gks_solver"] = "GKS(projection_dim=3, n_iter=30, regparam=1e-2), same A, b, L"
        out.update(gk)
    except Exception as exc:      # noqa: BLE001
        out["gks_error"] = f"{type(exc).__name__}: {exc}"[:200]
    if cpu_jobs is not None:
        bh = b.detach().to("cpu")
        cpu_jobs.append((lambda v: out.__setitem__("cpu_baseline", v), lambda: cpu_c4(psf, N, bh)))
    return out


def extra_c5_dynamic(rank, world, cpu_jobs=None):
    """BASELINE config C5 (dynamic parallel-beam tomography, 256^2 frames, 15 angles per frame shifted by 1 degree per
    frame, space-time derivative) at its BASELINE size: 32 frames in all, 32 / world per rank (STRONG scaling: the same
    problem at every N, also N = 1), global inner products all-reduced over RCCL, one-frame halo exchange for the temporal
    rows of the regulariser."""
    from trips_py_amd.dist import TorchComm, frame_range
    from trips_py_amd.engine import HipEngine
    from trips_py_amd.operators import BlockDiagOp, Radon2DParallel, SpaceTimeDerivative
    from trips_py_amd.solvers import CGLS, GKS
    Nf, nt, na = 256, 32, 15
    if nt % world:
        return {"skipped": f"32 frames do not divide over {world} ranks"}
    per_rank = nt // world
    # world > 1 over RCCL: libtrk's own communicator (trk_allreduce_f64: the sharded CGLS enqueues its all-reduces from C, no
    # Python dispatch per scalar); torch's if that cannot be created, or for the gloo debugging backend
    comm_kind = None
    if world > 1:
        import torch.distributed as dist
        comm = None
        if dist.get_backend() == "nccl" and os.environ.get("TRK_COMM", "rccl") == "rccl":
            # trk_comm_init is collective: a rank that cannot even load librccl must say so BEFORE the others enter it, or they
            # wait there forever.  Every rank probes on its own (trk_comm_unique_id: loads the library, no communication) and the
            # ranks agree on the outcome first.
            probe_ok, why = 1.0, ""
            try:
                from trips_py_amd import _lib as _trk_lib
                _buf = (ctypes.c_char * 128)()
                _trk_lib.check(_trk_lib.load().trk_comm_unique_id(_buf), "trk_comm_unique_id")
            except Exception as exc:      # noqa: BLE001
                probe_ok, why = 0.0, f"{type(exc).__name__}: {exc}"
            flag = torch.tensor([probe_ok], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) < 1.0:
                comm_kind = f"torch.distributed (librccl not usable through libtrk on some rank{': ' + why if why else ''})"[:200]
            else:
                try:
                    from trips_py_amd.dist import RcclComm
                    comm, comm_kind = RcclComm(), "libtrk RCCL communicator (trk_comm_init / trk_allreduce_f64 / trk_halo_exchange2)"
                except Exception as exc:      # noqa: BLE001
                    comm_kind = f"torch.distributed ({type(exc).__name__} from trk_comm_init: {exc})"[:200]
            # every rank must end up on the same communicator: if one could not make its own, all use torch.distributed
            ok = torch.tensor([1.0 if comm is not None else 0.0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0 and comm is not None:
                comm, comm_kind = None, "torch.distributed (another rank could not create the libtrk communicator)"
        if comm is None:
            comm, comm_kind = TorchComm(), comm_kind or f"torch.distributed ({dist.get_backend()})"
        eng = HipEngine(comm=comm)
    else:
        eng = HipEngine()
    lo, hi = frame_range(nt, world, rank)
    ops = [Radon2DParallel(Nf, np.deg2rad(t + 12.0 * np.arange(na)), engine=eng) for t in range(lo, hi)]
    F = BlockDiagOp(ops, engine=eng)
    L = SpaceTimeDerivative(Nf, nt, engine=eng)
    g = torch.Generator(device="cpu").manual_seed(1234)
    frames = []
    for t in range(nt):                                        # two bars drifting linearly in t (every rank builds all, keeps its own)
        img = torch.zeros((Nf, Nf))
        img[60 + 2 * t:100 + 2 * t, 40:200] = 1.0
        img[150:190, 30 + 3 * t:90 + 3 * t] = 0.6
        frames.append(img + 0.05 * torch.rand((Nf, Nf), generator=g))
    xt = torch.cat([f.reshape(-1) for f in frames[lo:hi]]).to(eng.device)
    bl = F.apply(xt)
    e = torch.randn(bl.numel(), device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(77 + rank))
    bl = bl + e * (0.01 * torch.linalg.norm(bl) / torch.linalg.norm(e))
    x0 = torch.zeros(F.shape[1], device=eng.device)
    out = {"frames_total": nt, "frames_per_rank": per_rank, "frame": f"{Nf}x{Nf}", "angles_per_frame": na,
           "scaling": "strong", "ranks": world, "communicator": comm_kind}
    reps = 3
    # CGLS in the arrangement CGLS() picks (ranks > 1: ONE all-reduce per iteration, csrc/cgls_sharded.hip; one rank: the
    # recurrence as written, nothing to exchange) and, beside it, the other one.  "reduction_points_per_iteration" = calls of the
    # engine's all-reduce per iteration, counted on one rank as well (there they move nothing)
    default_one = world > 1
    for key, kw in (("cgls", {}), ("cgls_one_reduction" if not default_one else "cgls_two_reductions", {"one_reduction": not default_one})):
        CGLS(F, bl, x0, 100, 0, history=False, **kw)             # warm-up: a whole solve of the timed size, as C2 / C3
        barrier(world)
        with no_gc():
            t0 = time.perf_counter()
            for _ in range(reps):
                CGLS(F, bl, x0, 100, 0, history=False, **kw)
            barrier(world)
            dt = max_over_ranks(time.perf_counter() - t0, world)
        out[f"{key}_iters_per_sec"] = round(reps * 100 / dt, 1)
    # reduction points per iteration of both arrangements, whatever the rank count (two solves of different length each)
    for key, kw in (("cgls_one_reduction", {"one_reduction": True}), ("cgls_two_reductions", {"one_reduction": False})):
        cnt = []
        for its in (20, 60):
            c0 = eng.reduction_points
            CGLS(F, bl, x0, its, 0, history=False, **kw)
            cnt.append(eng.reduction_points - c0)
        out[f"{key}_reduction_points_per_iteration"] = round((cnt[1] - cnt[0]) / 40.0, 3)
    if world == 1:       # the single-rank form of the recurrence keeps its sums as block partials and never calls the engine's all-reduce
        out["cgls_two_reductions_reduction_points_per_iteration"] = "2 on ranks > 1 (CGLS.py:61,70); 0 calls on one rank"
    GKS(F, bl, L, 3, 50, 1e-2, history=False)
    barrier(world)
    with no_gc():
        t0 = time.perf_counter()
        for _ in range(reps):
            GKS(F, bl, L, 3, 50, 1e-2, history=False)
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world)
    out["gks_iters_per_sec"] = round(reps * 50 / dt, 1)
    # the reference's DEFAULT regparam ('gcv': GKS.py:23): the projected problem visits the host every iteration (trk_host_gram_gcv)
    GKS(F, bl, L, 3, 50, "gcv", history=False)
    barrier(world)
    with no_gc():
        t0 = time.perf_counter()
        for _ in range(reps):
            GKS(F, bl, L, 3, 50, "gcv", history=False)
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world)
    out["gks_gcv_iters_per_sec"] = round(reps * 50 / dt, 1)
    cnt = []
    for its in (10, 30):
        c0, h0 = eng.reduction_points, eng.halo_exchanges
        _, ginfo = GKS(F, bl, L, 3, its, 1e-2, history=False)
        cnt.append((eng.reduction_points - c0, eng.halo_exchanges - h0))
    out["gks_reduction_points_per_iteration"] = round((cnt[1][0] - cnt[0][0]) / 20.0, 3)
    # ranks > 1: the fused space-time stencil takes the neighbours' boundary frames (trk_tv_halo); the iterate's and the new basis
    # vector's come from the basis vectors' (solvers/GKS._HaloTrack), the residual's from ONE two-sided exchange per iteration
    out["gks_halo_exchanges_per_iteration"] = round((cnt[1][1] - cnt[0][1]) / 20.0, 3) if world > 1 else "1 on ranks > 1; none on one rank"
    out["gks_fused_tv_kernels_on_every_rank"] = bool(ginfo.get("fused_tv", True)) and bool(getattr(L, "fused_tv", False))
    if cpu_jobs is not None:
        bh = bl.detach().to("cpu")

        def put(both):
            out["cgls_cpu_baseline"], out["gks_cpu_baseline"] = both if isinstance(both, tuple) else (both, both)
        cpu_jobs.append((put, lambda: cpu_c5(Nf, [np.deg2rad(t + 12.0 * np.arange(na)) for t in range(nt)], bh, nt)))
    return out


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:      # noqa: BLE001
        return None


def cpu_iteration_rate(op, solve, sample_iters):
    """Time `sample_iters` iterations of an ORACLE solver loop (oracle/cpu_ref.py: the reference's algorithm, float64, NumPy /
    SciPy on this box's host cores — the reported baseline, never the thing measured).  Every solver loop of the path applies
    A^T exactly once per iteration (CGLS.py:68, decompositions.py:235, GKS.py:82, MMGKS.py:117), so the applies of `op._adj` are
    stamped: `solve` runs sample_iters + 1 iterations and the time between the last sample_iters + 1 stamps is sample_iters
    whole iterations — set-up (start basis, first products) is outside, as on the GPU side where it is amortised over the solve."""
    stamps = []
    orig = op._adj

    def stamped(y):
        stamps.append(time.perf_counter())
        return orig(y)

    op._adj = stamped
    try:
        solve(sample_iters + 1)
    finally:
        op._adj = orig
    assert len(stamps) >= sample_iters + 1, (len(stamps), sample_iters)
    t = (stamps[-1] - stamps[-1 - sample_iters]) / sample_iters
    return 1.0 / t, t


def cpu_leg(value, t, sample, cores=1):
    return {"value": round(value, 4), "unit": "iters/s", "cores": cores, "kind": "port", "seconds_per_iter": round(t, 4),
            "sample": sample + f"; float64 oracle on the host (NumPy BLAS threads={host_threads()}, {os.cpu_count()} logical cores visible)"}


def cpu_c2(psf, N, b_dev, sample=20):
    """C2 on the host: the oracle's CGLS (CGLS.py:57-80) over scipy.ndimage.convolve (Deblurring2D.py:70-71), same b."""
    from oracle import cpu_ref as O
    Ao = O.Blur2D(psf, N, N)
    bh = b_dev.detach().cpu().numpy().astype(np.float64).reshape(-1, 1)
    v, t = cpu_iteration_rate(Ao, lambda k: O.cgls(Ao, bh, np.zeros((N * N, 1)), k, 0), sample)
    return cpu_leg(v, t, f"{sample} CGLS iterations of the same {N}x{N} problem; scipy.ndimage.convolve is single-threaded")


def cpu_c3(Nt, angles, b_dev, sample=8):
    """C3 on the host: the oracle's Hybrid-LSQR (Hybrid_LSQR.py:73-110, lambda = 1e-2) over the oracle's Joseph projector held as a
    scipy.sparse CSR matrix (ASTRA's CPU projector is not installable; matrix assembly is set-up, outside the sample)."""
    from oracle import cpu_ref as O
    Ro = O.Radon2D(Nt, angles)
    Ro.matrix()
    bh = b_dev.detach().cpu().numpy().astype(np.float64).reshape(-1, 1)
    v, t = cpu_iteration_rate(Ro, lambda k: O.hybrid_lsqr(Ro, bh, k, 1e-2), sample)
    return cpu_leg(v, t, f"iterations 2..{sample + 1} of Hybrid_LSQR(regparam=1e-2) on the same {Nt}x{Nt}, {len(angles)}-angle data "
                         "(the reference's cost per iteration grows with the basis: V y, hstack copies); SpMV single-threaded")


def cpu_sparse_dynamic(Fm, b_dev, sample=20):
    """The sparse dynamic problem on the host: the oracle's Hybrid-LSQR over the same scipy.sparse matrix (what the reference runs:
    demos/2_demo_dynamic_CrossPhantom.ipynb cell 15 hands the loader's scipy matrix straight to Hybrid_LSQR)."""
    from oracle import cpu_ref as O
    Fo = O.MatrixOp(Fm)
    bh = b_dev.detach().cpu().numpy().astype(np.float64).reshape(-1, 1)
    v, t = cpu_iteration_rate(Fo, lambda k: O.hybrid_lsqr(Fo, bh, k, 1e-2), sample)
    return cpu_leg(v, t, f"iterations 2..{sample + 1} of Hybrid_LSQR(regparam=1e-2) on the same sparse matrix; scipy CSR SpMV is single-threaded")


def cpu_c4(psf, N, b_dev):
    """C4 on the host: the oracle's MMGKS (MMGKS.py:55-128: two economic QRs of the re-weighted m x k and p x k images per
    iteration) on the same 4096^2 data — ONE iteration (tens of seconds), the first one (basis size 3; the GPU side's rate is the
    mean over k = 3..32, and the reference's cost grows with k).  Set-up (Golub-Kahan start basis, A V, L V: MMGKS.py:37-44) is
    outside the sample: A is applied 6 times there (3 Golub-Kahan steps, 3 columns of A V), so the 7th apply is `A @ x` at the top
    of the first iteration (MMGKS.py:56) and the sample runs from there to the solver's return."""
    from oracle import cpu_ref as O
    Ao, Lo = O.Blur2D(psf, N, N), O.FirstDerivative2D(N)
    bh = b_dev.detach().cpu().numpy().astype(np.float64).reshape(-1, 1)
    stamps = []
    orig = Ao._fwd

    def stamped(x):
        stamps.append(time.perf_counter())
        return orig(x)

    Ao._fwd = stamped
    try:
        O.mmgks(Ao, bh, Lo, 2, 1, 3, 1, 1e-2, epsilon=0.1)
    finally:
        Ao._fwd = orig
    t_end = time.perf_counter()
    assert len(stamps) == 8, len(stamps)          # 6 in the set-up, A @ x and A @ v_new in the iteration
    t = t_end - stamps[6]
    return cpu_leg(1.0 / t, t, f"the first iteration (basis size 3) of MMGKS(pnorm=2, qnorm=1, projection_dim=3, regparam=1e-2) on the "
                               f"same {N}x{N} problem, set-up excluded; convolution single-threaded, QRs on the BLAS threads",
                   cores=host_threads() or 1)


def cpu_c5(Nf, angle_sets, b_dev, nt, sample_cgls=10, sample_gks=3):
    """C5 on the host (all 32 frames, one process): the oracle's CGLS and GKS (GKS.py:42-96, space-time L) over the block-diagonal
    Joseph projector as scipy.sparse matrices."""
    from oracle import cpu_ref as O
    Fo = O.BlockDiag([O.Radon2D(Nf, a) for a in angle_sets])
    for o in Fo.ops:
        o.matrix()
    Lo = O.SpaceTimeDerivative(Nf, nt)
    bh = b_dev.detach().cpu().numpy().astype(np.float64).reshape(-1, 1)
    v1, t1 = cpu_iteration_rate(Fo, lambda k: O.cgls(Fo, bh, np.zeros((Fo.shape[1], 1)), k, 0), sample_cgls)
    v2, t2 = cpu_iteration_rate(Fo, lambda k: O.gks(Fo, bh, Lo, 3, k, 1e-2), sample_gks)
    what = f"the same {nt} x {Nf}x{Nf} data; SpMV single-threaded, QRs on the BLAS threads"
    return (cpu_leg(v1, t1, f"{sample_cgls} CGLS iterations on " + what),
            cpu_leg(v2, t2, f"iterations 2..{sample_gks + 1} (basis size 4..{sample_gks + 3}) of GKS(projection_dim=3, regparam=1e-2) on " + what,
                    cores=host_threads() or 1))


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
    spawn_ranks_if_needed(args)
    # stdout carries exactly ONE line, the JSON: whatever libraries print there while the bench runs (RCCL's version banner
    # at communicator creation, for one) is sent to stderr — at the file-descriptor level, C code included
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the engine has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("TRK_COMM_FORCE", "1")      # the RCCL latency probe of extra.trk_comm_rccl on one rank (see its note)
    rank, world = dist_setup(args)
    res = run_blur_cgls(args, rank, world, json_fd)
    if rank == 0:
        emit_json(json_fd, res)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
