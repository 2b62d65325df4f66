"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  The hot path has exactly two kinds of exchange when a dynamic problem is sharded by time frames:

  * all-reduce(sum) of a handful of float64 scalars (inner products, Gram rows): latency-bound, so reductions of one
    synchronisation point share one call;
  * the boundary frames of a vector between time-neighbours for the temporal rows of the space-time regulariser: ONE two-sided
    exchange per operand of a fused stencil (`exchange2`), or a one-frame shift per direction of a plain L / L^T apply (`shift`).

Everything else (operator applies, axpys, weights) is local to a rank's frames.
"""
import torch
import torch.distributed as dist


class TorchComm:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # gloo cannot move device tensors point-to-point: stage through the host (debug / CPU-test backend only)
        self.host_staging = dist.get_backend(group) == "gloo"

    def allreduce_sum_(self, t):
        """In-place sum over ranks of a float64 tensor (device tensor under nccl, CPU tensor under gloo)."""
        if self.host_staging and t.is_cuda:
            h = t.detach().to("cpu")
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def shift(self, send, send_to, recv, recv_from):
        """Send `send` to rank `send_to` and receive into `recv` from rank `recv_from` (either side may be None)."""
        do_send = send is not None and 0 <= send_to < self.world
        do_recv = recv is not None and recv_from is not None and 0 <= recv_from < self.world
        stage = self.host_staging and ((do_send and send.is_cuda) or (do_recv and recv.is_cuda))
        sbuf = (send.detach().to("cpu") if stage else send.contiguous()) if do_send else None
        rbuf = (torch.empty(recv.shape, dtype=recv.dtype) if stage else recv) if do_recv else None
        ops = []
        # P2POp peers are GLOBAL ranks; send_to / recv_from are ranks of this communicator's group
        if do_send:
            ops.append(dist.P2POp(dist.isend, sbuf, self._global(send_to), self.group))
        if do_recv:
            ops.append(dist.P2POp(dist.irecv, rbuf, self._global(recv_from), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if do_recv and stage:
            recv.copy_(rbuf)

    def exchange2(self, send_prev, recv_prev, send_next, recv_next):
        """Both time-neighbours in ONE batch: send_prev -> rank-1 / recv_prev <- rank-1 (rank > 0), send_next -> rank+1 /
        recv_next <- rank+1 (rank < world-1).  The boundary frames a fused space-time stencil needs of a sharded vector."""
        hp, hn = self.rank > 0, self.rank < self.world - 1
        pairs = ([(send_prev, recv_prev, self.rank - 1)] if hp else []) + ([(send_next, recv_next, self.rank + 1)] if hn else [])
        ops, back = [], []
        for snd, rcv, peer in pairs:
            stage = self.host_staging and (snd.is_cuda or rcv.is_cuda)
            sbuf = snd.detach().to("cpu") if stage else snd.contiguous()
            rbuf = torch.empty(rcv.shape, dtype=rcv.dtype) if stage else rcv
            ops.append(dist.P2POp(dist.isend, sbuf, self._global(peer), self.group))
            ops.append(dist.P2POp(dist.irecv, rbuf, self._global(peer), self.group))
            if stage:
                back.append((rcv, rbuf))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for rcv, rbuf in back:
            rcv.copy_(rbuf)

    def _global(self, group_rank):
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    def barrier(self):
        dist.barrier(group=self.group)


class RcclComm:
    """The same two exchanges through libtrk.so's own RCCL entry points (trk_comm_init / trk_allreduce_f64 / trk_halo_exchange,
    include/trk.h) — what a host without PyTorch binds.  Here torch.distributed (any backend) is used once, to hand rank 0's
    128-byte unique id to the other ranks; world = 1 needs nothing.  Select with TRK_COMM=rccl (`make_comm()`)."""

    host_staging = False

    def __init__(self, rank=None, world=None):
        import ctypes
        from . import _lib
        self.lib = _lib.load()
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world = int(rank), int(world)
        idbuf = (ctypes.c_char * 128)()
        why = None
        if self.rank == 0:
            try:
                _lib.check(self.lib.trk_comm_unique_id(idbuf), "trk_comm_unique_id")
            except Exception as exc:      # noqa: BLE001  (the other ranks are waiting in the broadcast: tell them, then raise)
                why = f"{type(exc).__name__}: {exc}"
        if self.world > 1:
            box = [(why, bytes(idbuf))]
            dist.broadcast_object_list(box, src=0)
            why = box[0][0]
            ctypes.memmove(idbuf, box[0][1], 128)
        if why is not None:
            raise RuntimeError(f"RcclComm: rank 0 could not create the unique id ({why})")
        self._h = ctypes.c_void_p()
        _lib.check(self.lib.trk_comm_init(idbuf, self.rank, self.world, ctypes.byref(self._h)), "trk_comm_init")

    def _stream(self, t):
        return torch.cuda.current_stream(t.device).cuda_stream

    def allreduce_sum_(self, t):
        from . import _lib
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("RcclComm.allreduce_sum_: contiguous float64 device tensor expected")
        _lib.check(self.lib.trk_allreduce_f64(self._h, t.data_ptr(), t.numel(), self._stream(t)), "trk_allreduce_f64")
        return t

    def shift(self, send, send_to, recv, recv_from):
        from . import _lib
        ref = send if send is not None else recv
        if ref is None:
            return
        sb = None if send is None else send.contiguous()
        count = (sb if sb is not None else recv).numel()
        rc = self.lib.trk_halo_exchange(self._h, None if sb is None else sb.data_ptr(), -1 if send_to is None else int(send_to),
                                        None if recv is None else recv.data_ptr(), -1 if recv_from is None else int(recv_from),
                                        count, self._stream(ref))
        _lib.check(rc, "trk_halo_exchange")

    def exchange2(self, send_prev, recv_prev, send_next, recv_next):
        from . import _lib
        hp, hn = self.rank > 0, self.rank < self.world - 1
        if not (hp or hn):
            return
        ref = send_prev if hp else send_next
        sp = send_prev.contiguous() if hp else None
        sn = send_next.contiguous() if hn else None
        rc = self.lib.trk_halo_exchange2(self._h, None if sp is None else sp.data_ptr(), recv_prev.data_ptr() if hp else None,
                                         None if sn is None else sn.data_ptr(), recv_next.data_ptr() if hn else None,
                                         ref.numel(), self._stream(ref))
        _lib.check(rc, "trk_halo_exchange2")

    def barrier(self):
        if dist.is_initialized():
            dist.barrier()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.lib.trk_comm_destroy(h)
            except Exception:
                pass


def make_comm():
    """The communicator of this process: TorchComm (default) or, with TRK_COMM=rccl, libtrk.so's own RCCL communicator."""
    import os
    return RcclComm() if os.environ.get("TRK_COMM", "torch") == "rccl" else TorchComm()


def frame_range(n_frames, world, rank):
    """Frames [lo, hi) owned by `rank` (contiguous, frame-major layout of x and b: io.py:223-225)."""
    if n_frames % world:
        raise ValueError(f"{n_frames} frames do not shard evenly over {world} ranks")
    per = n_frames // world
    return rank * per, (rank + 1) * per
