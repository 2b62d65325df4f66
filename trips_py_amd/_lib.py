"""Build + load libtrk.so (the C-ABI engine, include/trk.h) through ctypes.

There is NO CPU fallback: if the shared library is missing or cannot be loaded, or no GPU is
visible when a kernel is requested, the product path raises.  `build()` cross-compiles for gfx950
with hipcc (no GPU needed) into trips_py_amd/csrc/libtrk.so, in-tree, so the built library
travels with the source snapshot.
"""
import ctypes
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(CSRC, "libtrk.so")
SOURCES = ["core.hip", "vecops.hip", "blur2d.hip", "tvops.hip", "radon2d.hip", "spmv.hip", "fanbeam2d.hip", "projected.hip", "cgls_loop.hip", "comm.hip", "cgls_tiled.hip", "cgls_sharded.hip", "ref64.hip"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]


class TrkError(RuntimeError):
    pass


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise TrkError("hipcc not found: cannot build libtrk.so (ROCm toolchain required)")
    return exe


STAMP_PATH = LIB_PATH + ".stamp"


def _source_hash():
    """Hash of everything the library is built from.  Staleness is decided by content, not by modification times: a copied
    tree (the GPU box receives a snapshot) must not trigger rebuilds, least of all from eight ranks at once."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS[:-2] + SOURCES).encode())
    deps = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for path in [os.path.join(CSRC, f) for f in deps] + [os.path.join(INCLUDE, "trk.h")]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    try:
        with open(STAMP_PATH) as fh:
            return fh.read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libtrk.so.  Returns the library path."""
    if not force and not _stale():
        return LIB_PATH
    # one builder at a time (several ranks may import the package together)
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _file_hash(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for path in paths:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _build_locked(force, verbose):
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "trk.h")]

    def cc(src):
        # an object is reused only if the CONTENT of its source, of every header and the flags are what it was built
        # from (modification times mean nothing in a copied tree)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        key = _file_hash([os.path.join(CSRC, src)] + headers, " ".join(HIPCC_FLAGS[:-2]))
        if not force and os.path.exists(obj) and os.path.exists(obj + ".stamp"):
            with open(obj + ".stamp") as fh:
                if fh.read().strip() == key:
                    return obj
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if os.path.exists(obj + ".stamp"):
            os.unlink(obj + ".stamp")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise TrkError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        with open(obj + ".stamp", "w") as fh:
            fh.write(key + "\n")
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    if os.path.exists(STAMP_PATH):
        os.unlink(STAMP_PATH)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise TrkError(f"link of libtrk.so failed:\n{r.stdout}\n{r.stderr}")
    with open(STAMP_PATH, "w") as fh:
        fh.write(_source_hash() + "\n")
    return LIB_PATH


# ----------------------------------------------------------------------------- ctypes signatures
c_f32p = ctypes.c_void_p     # device pointers travel as integers (tensor.data_ptr())
c_f64p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_op = ctypes.c_void_p
c_stream = ctypes.c_void_p

SIGNATURES = {
    "trk_version": (c_int, []),
    "trk_last_error": (ctypes.c_char_p, []),
    "trk_device_info": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "trk_blur2d_create": (c_int, [ctypes.POINTER(c_dbl), c_int, c_int, c_int, c_int, ctypes.POINTER(c_op)]),
    "trk_radon2d_create": (c_int, [c_int, c_int, ctypes.POINTER(c_dbl), c_int, c_dbl, ctypes.POINTER(c_op)]),
    "trk_radon2d_dynamic_create": (c_int, [c_int, c_int, ctypes.POINTER(c_dbl), c_int, c_int, c_dbl, ctypes.POINTER(c_op)]),
    "trk_fanbeam2d_create": (c_int, [c_int, c_int, c_dbl, c_dbl, c_dbl, ctypes.POINTER(c_dbl), c_int, ctypes.POINTER(c_op)]),
    "trk_deriv2d_create": (c_int, [c_int, ctypes.POINTER(c_op)]),
    "trk_spacetime_create": (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_op)]),
    "trk_spacetime_set_halo": (c_int, [c_op, c_f32p, c_f32p]),
    "trk_csr_create": (c_int, [c_i64, c_i64, c_i64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(c_op)]),
    "trk_blockdiag_create": (c_int, [ctypes.POINTER(c_op), c_int, ctypes.POINTER(c_op)]),
    "trk_op_shape": (c_int, [c_op, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "trk_op_apply": (c_int, [c_op, c_int, c_f32p, c_i64, c_f32p, c_i64, c_int, c_f64p, c_stream]),
    "trk_op_destroy": (c_int, [c_op]),
    "trk_timer_create": (c_int, [c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "trk_timer_reset": (c_int, [ctypes.c_void_p]),
    "trk_timer_read": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), c_int, ctypes.POINTER(c_int)]),
    "trk_timer_destroy": (c_int, [ctypes.c_void_p]),
    "trk_op_set_timer": (c_int, [c_op, ctypes.c_void_p, c_int]),
    "trk_dot": (c_int, [c_f32p, c_f32p, c_i64, c_f64p, c_stream]),
    "trk_nrm2sq": (c_int, [c_f32p, c_i64, c_f64p, c_stream]),
    "trk_diff_nrm2sq": (c_int, [c_f32p, c_f32p, c_i64, c_f64p, c_stream]),
    "trk_axpby": (c_int, [c_i64, c_dbl, c_f64p, c_f64p, c_int, c_f32p, c_dbl, c_f64p, c_f64p, c_int, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_scale_dot": (c_int, [c_i64, c_dbl, c_f64p, c_f64p, c_int, c_f32p, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_mul": (c_int, [c_i64, c_f32p, c_f32p, c_f32p, c_stream]),
    "trk_mul_diff": (c_int, [c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "trk_group_weights": (c_int, [c_f32p, c_i64, c_int, c_dbl, c_dbl, c_int, c_f32p, c_stream]),
    "trk_isotv_weights": (c_int, [c_f32p, c_int, c_int, c_f32p, c_i64, c_dbl, c_dbl, c_f32p, c_stream]),
    "trk_tv_weights": (c_int, [c_op, c_f32p, c_dbl, c_dbl, c_f32p, c_stream]),
    "trk_arnoldi_step": (c_int, [c_op, c_f32p, c_i64, c_int, c_f32p, c_f64p, c_int, c_f64p, c_f64p, c_stream]),
    "trk_host_gram_gcv": (c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 c_int, c_dbl, ctypes.POINTER(c_dbl), ctypes.c_void_p, ctypes.POINTER(c_int)]),
    "trk_hlsqr_select": (c_int, [ctypes.c_void_p, c_int, ctypes.c_void_p, ctypes.c_void_p, c_int, c_dbl, c_dbl, ctypes.c_void_p, c_dbl, c_int,
                                c_f32p, c_i64, c_i64, c_f32p, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_dbl),
                                ctypes.POINTER(c_int), c_stream]),
    "trk_hgmres_create": (c_int, [c_op, c_f32p, c_i64, c_int, c_f32p, c_f64p, c_int, c_f64p, c_f64p, ctypes.c_void_p,
                                 ctypes.POINTER(ctypes.c_void_p), c_int, c_dbl, c_stream,
                                 ctypes.POINTER(ctypes.c_void_p)]),
    "trk_hgmres_destroy": (c_int, [ctypes.c_void_p]),
    "trk_hgmres_start": (c_int, [ctypes.c_void_p]),
    "trk_hgmres_dp": (c_int, [ctypes.c_void_p, c_f32p, c_dbl, c_dbl, c_dbl, ctypes.POINTER(ctypes.POINTER(c_dbl))]),
    "trk_hgmres_fixed_lambda": (c_int, [ctypes.c_void_p, c_dbl]),
    "trk_hgmres_stats": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_dbl)]),
    "trk_hgmres_iter": (c_int, [ctypes.c_void_p, c_int, c_int, c_int, c_f32p, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int),
                               ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)]),
    "trk_hgmres_hessenberg": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(c_dbl)), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "trk_arnoldi_step_post_dot": (c_int, [c_op, c_f32p, c_i64, c_int, c_f32p, c_f64p, c_int, c_f64p, c_f64p, ctypes.c_void_p, c_int, c_int, c_int,
                                         c_int, c_f32p, c_int, c_stream]),
    "trk_arnoldi_step_post_at": (c_int, [c_op, c_f32p, c_i64, c_int, c_f32p, c_f64p, c_int, c_f64p, c_f64p, ctypes.c_void_p, c_int, c_int, c_int,
                                        c_int, c_stream]),
    "trk_mailbox_doubles": (c_int, [ctypes.c_void_p]),
    "trk_mailbox_slots": (c_int, [ctypes.c_void_p]),
    "trk_host_worker_post_hess_fixed": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i64, c_int, c_dbl, c_dbl]),
    "trk_arnoldi_step_post": (c_int, [c_op, c_f32p, c_i64, c_int, c_f32p, c_f64p, c_int, c_f64p, c_f64p, ctypes.c_void_p, c_int, c_int, c_int,
                                     c_stream]),
    "trk_tv_grad": (c_int, [c_op, c_f32p, c_f32p, c_f32p, c_dbl, c_f32p, c_stream]),
    "trk_tv_halo": (c_int, [c_op, c_f32p, c_f32p]),
    "trk_tv_grad_dot": (c_int, [c_op, c_f32p, c_f32p, c_f32p, c_dbl, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_tv_grad_dot_xsq": (c_int, [c_op, c_f32p, c_f32p, c_f32p, c_dbl, c_f32p, c_f32p, c_f64p, c_f64p, c_stream]),
    "trk_mm_weights": (c_int, [c_i64, c_f32p, c_f32p, c_dbl, c_dbl, c_f32p, c_stream]),
    "trk_cgls_update_xr": (c_int, [c_i64, c_i64, c_f64p, c_f64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_op_fused_caps": (c_int, [c_op, ctypes.POINTER(c_int)]),
    "trk_op_axpby_caps": (c_int, [c_op, ctypes.POINTER(c_int)]),
    "trk_op_flush": (c_int, [c_op, c_stream]),
    "trk_op_apply_axpby": (c_int, [c_op, c_int, c_f32p, c_dbl, c_f64p, c_f64p, c_int, c_dbl, c_f64p, c_f64p, c_int, c_f32p,
                                   c_f32p, c_f64p, c_int, c_stream]),
    "trk_op_apply_fused": (c_int, [c_op, c_int, c_f32p, c_f32p, c_dbl, c_f64p, c_int, c_f64p, c_int, c_f32p, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_cgls_x_update": (c_int, [c_i64, c_f64p, c_int, c_f64p, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_finalize_batched": (c_int, [c_f64p, c_int, c_int, c_int, c_f64p, c_int, c_stream]),
    "trk_cgls_update_xr_deferred": (c_int, [c_i64, c_i64, c_f64p, c_f64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_bidiag_tikhonov": (c_int, [c_f64p, c_i64, c_f64p, c_i64, c_int, c_dbl, c_f64p, c_f64p, c_int, c_f64p, c_int,
                                   c_stream]),
    "trk_host_dp_newton": (c_int, [c_f64p, c_f64p, c_int, c_dbl, c_dbl, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int),
                                  ctypes.POINTER(c_int)]),
    "trk_host_gcv_fminbound": (c_int, [c_f64p, c_f64p, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)]),
    "trk_host_dp_bidiag": (c_int, [c_f64p, c_f64p, c_int, c_f64p, c_dbl, c_dbl, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int),
                                   ctypes.POINTER(c_int), ctypes.POINTER(c_dbl)]),
    "trk_host_gcv_bidiag": (c_int, [c_f64p, c_f64p, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int, ctypes.POINTER(c_dbl),
                                    ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)]),
    "trk_cgls_iterate": (c_int, [c_op, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_f32p, c_f32p,
                                 c_f64p, c_f64p, c_int, ctypes.POINTER(c_int), c_f64p, c_f64p, c_int, c_int, c_stream]),
    "trk_cgls_update_xr_src": (c_int, [c_i64, c_i64, c_f64p, c_int, c_f64p, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                       c_f32p, c_f64p, c_f64p, c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_cgls_update_grouping": (c_int, [c_i64]),
    "trk_cgls_r_update": (c_int, [c_i64, c_f64p, c_f64p, c_int, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_cgls_xp_update": (c_int, [c_i64, c_f64p, c_f64p, c_f64p, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_f64p,
                                   c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_cgls_p_update": (c_int, [c_i64, c_f32p, c_f32p, c_f64p, c_int, c_f64p, c_f64p, c_stream]),
    "trk_cgls_iterate_fused": (c_int, [c_op, c_int, c_int, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_i64, c_int,
                                       c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_int, c_f64p, c_int, ctypes.POINTER(c_int),
                                       ctypes.POINTER(c_int), c_stream]),
    "trk_cgls_tiled_caps": (c_int, [c_op, c_int, c_int, ctypes.POINTER(c_int)]),
    "trk_cgls_iterate_tiled": (c_int, [c_op, c_int, c_int, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_int,
                                       c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_int, c_f64p, c_int, ctypes.POINTER(c_int),
                                       ctypes.POINTER(c_int), c_stream]),
    "trk_cgls_iterate_tiled2": (c_int, [c_op, c_int, c_int, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_int,
                                        c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_int, c_f64p, c_int, ctypes.POINTER(c_int),
                                        ctypes.POINTER(c_int), c_stream]),
    "trk_gemv_t": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_gemv_t_x": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f32p, c_f32p, c_f64p, c_f64p, c_stream]),
    "trk_gemv_t2": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_gemv_tn": (c_int, [c_f32p, c_i64, c_int, c_i64, ctypes.POINTER(ctypes.c_void_p), c_int, c_f64p, c_stream]),
    "trk_gram_row_from_sweep": (c_int, [c_f64p, c_int, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_stream]),
    "trk_gram_tikhonov": (c_int, [c_f64p, c_int, c_f64p, c_int, c_f64p, c_int, c_dbl, c_f64p, c_int, c_int, c_f64p, c_stream]),
    "trk_hess_tikhonov": (c_int, [c_f64p, c_int, c_f64p, c_f64p, c_int, c_f64p, c_f64p, c_f64p, c_dbl, c_int, c_dbl, c_int, c_f64p,
                                  c_stream]),
    "trk_cgs_coeffs": (c_int, [c_f64p, c_int, c_f64p, c_f64p, c_int, c_int, c_f64p, c_stream]),
    "trk_gks_rows_solve": (c_int, [c_f64p, c_f64p, c_int, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_dbl,
                                   c_f64p, c_int, c_int, c_f64p, c_stream]),
    "trk_cgs_coeffs_rho": (c_int, [c_f64p, c_int, c_f64p, c_f64p, c_int, c_int, c_f64p, c_f64p, c_f64p, c_stream]),
    "trk_gemv_orth_iterate": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f32p, c_f64p, c_f64p, c_f64p, c_f32p, c_f32p, c_f32p, c_f64p, c_int,
                                      ctypes.POINTER(c_int), c_f64p, c_stream]),
    "trk_gemv_nt": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f64p, c_f32p, c_f32p, c_f64p, c_stream]),
    "trk_gemv_n_err": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f64p, c_f32p, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_host_worker_create": (c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "trk_host_worker_destroy": (c_int, [ctypes.c_void_p]),
    "trk_host_worker_post_gcv_bidiag": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_int, c_dbl, c_dbl, c_dbl, c_dbl,
                                                c_dbl, c_int]),
    "trk_host_worker_post_dp_bidiag": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_int, ctypes.c_void_p, c_dbl, c_dbl]),
    "trk_host_worker_collect": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)]),
    "trk_host_worker_set_lapack": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "trk_host_worker_post_hess_gcv": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i64, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int]),
    "trk_host_worker_post_hess_dp": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i64, c_int, c_dbl, ctypes.c_void_p, c_dbl, c_dbl]),
    "trk_host_worker_collect_vec": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.c_void_p, c_int,
                                            ctypes.POINTER(c_dbl)]),
    "trk_scalars_put": (c_int, [c_f64p, ctypes.c_void_p, c_int, c_stream]),
    "trk_mailbox_create": (c_int, [c_int, c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "trk_mailbox_destroy": (c_int, [ctypes.c_void_p]),
    "trk_mailbox_host": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "trk_mailbox_post": (c_int, [ctypes.c_void_p, c_int, c_f64p, c_int, c_int, c_stream]),
    "trk_mailbox_wait": (c_int, [ctypes.c_void_p, c_int]),
    "trk_mailbox_post_sum": (c_int, [ctypes.c_void_p, c_int, c_f64p, c_int, c_int, c_f64p, c_int, c_f64p, c_int, c_stream]),
    "trk_gk_step_post": (c_int, [c_op, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_int, c_int, c_int, c_f32p, c_f64p, c_int,
                                 ctypes.POINTER(c_int), ctypes.c_void_p, c_int, c_f64p, c_int, c_int, c_f64p, c_int, c_f64p, c_int,
                                 c_stream]),
    "trk_gk_step_lsqr": (c_int, [c_op, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                 c_f64p, c_int, ctypes.POINTER(c_int), c_dbl, c_f64p, c_f64p, c_stream]),
    "trk_gk_step_proj": (c_int, [c_op, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_int, c_int, c_int, c_f32p, c_f64p, c_int,
                                 ctypes.POINTER(c_int), c_stream]),
    "trk_host_bidiag_tikhonov": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int, c_dbl, c_dbl, c_int, ctypes.c_void_p]),
    "trk_gemv_n_hosty": (c_int, [c_f32p, c_i64, c_int, c_i64, ctypes.c_void_p, c_f32p, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int),
                                 c_stream]),
    "trk_gk_step": (c_int, [c_op, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_int, c_int, c_int, c_stream]),
    "trk_wgram_tv_precision": (c_int, [c_int]),
    "trk_wgram_tv_last_probe": (c_int, [ctypes.POINTER(c_dbl)]),
    "trk_radon2d_apply_ref": (c_int, [c_op, c_int, c_int, c_int, ctypes.c_void_p, ctypes.c_void_p, c_stream]),
    "trk_radon2d_set_arithmetic": (c_int, [c_op, c_int]),
    "trk_radon2d_set_ref_sums": (c_int, [c_op, c_int, c_int]),
    "trk_ref_axpby": (c_int, [c_int, c_i64, c_dbl, c_f64p, c_f64p, c_int, ctypes.c_void_p, c_dbl, c_f64p, c_f64p, c_int, ctypes.c_void_p,
                              ctypes.c_void_p, c_f64p, c_stream]),
    "trk_gk_lsqr_chain": (c_int, [c_op, c_int, c_int, ctypes.c_void_p, c_int, c_dbl, ctypes.c_void_p, ctypes.c_void_p, c_f64p, c_f64p,
                                  c_stream]),
    "trk_lsqr_damped_update": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f64p, c_int, ctypes.POINTER(c_int),
                                       c_f64p, c_f64p, c_f64p, c_dbl, c_f64p, c_f64p, c_int, c_stream]),
    "trk_gemv_n": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f64p, c_dbl, c_f32p, c_dbl, c_f32p, c_f64p, c_stream]),
    "trk_comm_unique_id": (c_int, [ctypes.c_void_p]),
    "trk_comm_init": (c_int, [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "trk_comm_attach": (c_int, [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "trk_comm_info": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "trk_comm_destroy": (c_int, [ctypes.c_void_p]),
    "trk_allreduce_f64": (c_int, [ctypes.c_void_p, c_f64p, c_int, c_stream]),
    "trk_halo_exchange": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_f32p, c_int, c_i64, c_stream]),
    "trk_halo_exchange2": (c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_stream]),
    "trk_dot_pair": (c_int, [c_f32p, c_f32p, c_i64, c_f64p, c_stream]),
    "trk_cgls_sharded_update": (c_int, [c_i64, c_i64, c_f64p, c_f64p, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                        c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_int, ctypes.POINTER(c_int), c_stream]),
    "trk_cgls_sharded_scalars": (c_int, [c_f32p, c_f32p, c_i64, c_f64p, c_int, c_f64p, c_stream]),
    "trk_cgls_iterate_sharded": (c_int, [c_op, ctypes.c_void_p, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64,
                                         c_int, c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_int, ctypes.POINTER(c_int), c_f64p, c_int,
                                         ctypes.POINTER(c_int), c_stream]),
    "trk_wgram_tv_z": (c_int, [c_f32p, c_i64, c_int, c_int, c_f32p, c_f64p, c_f32p, c_f64p, c_stream]),
    "trk_wgram_tv": (c_int, [c_f32p, c_i64, c_int, c_int, c_f32p, c_f64p, c_stream]),
    "trk_wgram": (c_int, [c_f32p, c_i64, c_int, c_i64, c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_stream]),
}

_lib = None


def lib_path():
    return LIB_PATH


def load():
    """dlopen libtrk.so (building it first if hipcc is present and the library is stale/missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled libamdhip64.so (same SONAME as ROCm's) must be the HIP runtime of the
    # process, so that torch's streams / allocations and libtrk's kernels live in one runtime.
    import torch  # noqa: F401
    if _stale():
        # a stale library is never loaded silently: either the rebuild succeeds or the caller hears about it
        # (TRK_ALLOW_STALE=1: load what is there, with a warning - for boxes without hipcc)
        try:
            build()
        except TrkError as exc:
            if not os.path.exists(LIB_PATH) or os.environ.get("TRK_ALLOW_STALE", "0") != "1":
                raise TrkError(f"libtrk.so is out of date with its sources and could not be rebuilt: {exc}") from exc
            print(f"trips_py_amd: WARNING: loading a STALE libtrk.so (TRK_ALLOW_STALE=1): {exc}", file=sys.stderr)
    if not os.path.exists(LIB_PATH):
        raise TrkError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                       "There is no CPU fallback for the engine.")
    path = LIB_PATH
    exp = os.environ.get("TRK_EXPERIMENT_LIB")
    if exp:
        # A/B measurements of kernel variants on ONE box (tools/r06_ab_libs.sh builds them with -DTRK_... switches): a library of the same
        # ABI under another path.  Announced on stderr; never set by the product.
        if not os.path.exists(exp):
            raise TrkError(f"TRK_EXPERIMENT_LIB={exp}: no such file")
        print(f"trips_py_amd: loading the EXPERIMENT library {exp} instead of {LIB_PATH}", file=sys.stderr)
        path = exp
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header / library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().trk_last_error().decode("utf-8", "replace")
        exc = ValueError if rc == -1 else (NotImplementedError if rc == -4 else TrkError)   # -5 (RCCL): TrkError
        raise exc(f"{what}: trk error {rc}: {msg}")
