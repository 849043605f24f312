"""Data-parallel training: one process per MI355X, gradients all-reduced with RCCL over xGMI.

Replaces the reference's single-process `torch.nn.DataParallel` (e.g.
deblurring-diffusion-pytorch/celebA_128.py:102), which re-broadcasts all 226 MB of weights on every
forward and reduces gradients to GPU 0 on every micro-step.  Here every rank owns a full replica
(weights, Adam state, EMA: < 1 GB of 288 GB), draws its own minibatch shard and timesteps, and the
only exchange is ONE sum-all-reduce of the flat gradient arena per optimizer step, issued in
buckets on a side stream as soon as the blocks that own a bucket have finished their backward
(so it overlaps with the rest of the backward pass), on the last accumulation micro-step only.
Launch with `python -m torch.distributed.run --nproc-per-node N ...` (RANK / LOCAL_RANK / WORLD_SIZE).
"""
import os

import torch
import torch.distributed as dist

from . import runtime as rt

_engine = None
BUCKET_BYTES = 32 << 20   # xGMI is per-link bound: few, large messages


def world_size():
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank():
    return int(os.environ.get("RANK", "0"))


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    if world_size() == 1 or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
    dist.init_process_group(backend=backend, rank=rank(), world_size=world_size())


class GradSync:
    """Bucketed, overlapped all-reduce of a FlatArena's gradient buffer."""

    def __init__(self, arena, bucket_bytes=BUCKET_BYTES):
        self.arena = arena
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on_gpu = arena.grad.is_cuda
        n = arena.numel
        per = max(1, bucket_bytes // 4)
        self.bounds = [(lo, min(lo + per, n)) for lo in range(0, n, per)]
        self.bucket_of = {}
        self.count0 = [0] * len(self.bounds)
        for p, o in zip(arena.params, arena.offsets):
            b = min(o // per, len(self.bounds) - 1)
            self.bucket_of[id(p)] = b
            self.count0[b] += 1
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        self.armed = False
        self.pending = None
        self.launched = None
        self.works = []

    def arm(self):
        """Call before the backward pass of the LAST accumulation micro-step."""
        if self.world == 1:
            return
        self.armed = True
        self.pending = list(self.count0)
        self.launched = [False] * len(self.bounds)
        self.works = []

    def _launch(self, b):
        lo, hi = self.bounds[b]
        buf = self.arena.grad[lo:hi]
        self.launched[b] = True
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    def ready(self, params):
        """The gradients of `params` are final (their backward kernels are enqueued)."""
        if not self.armed:
            return
        for p in params:
            b = self.bucket_of.get(id(p))
            if b is None:
                continue
            self.pending[b] -= 1
            if self.pending[b] == 0 and not self.launched[b]:
                self._launch(b)

    def finish(self):
        """After backward returned: reduce whatever is left and make the compute stream wait."""
        if not self.armed:
            return
        for b in range(len(self.bounds)):
            if not self.launched[b]:
                self._launch(b)
        for w in self.works:
            w.wait()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.armed = False


def set_engine(e):
    global _engine
    _engine = e


def grads_ready(params):
    if _engine is not None:
        _engine.ready(params)
