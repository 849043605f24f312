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
import contextlib
import datetime
import os

# dmabuf IPC for RCCL / cross-process device memory on these hosts (the legacy mode fails in hipIpcGetMemHandle); must be in the
# environment before the HIP runtime starts, i.e. before the first torch.cuda call below.  A value the launcher set wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from . import runtime as rt

_engine = None
BUCKET_BYTES = 32 << 20   # xGMI is per-link bound: few, large messages


def world_size():
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank():
    return int(os.environ.get("RANK", "0"))


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def pin_device():
    """One process per GPU: bind this process to `cuda:LOCAL_RANK` so that the `.cuda()` calls of an unmodified reference
    script (e.g. deblurring-diffusion-pytorch/celebA_128.py:100-102) land on the rank's own GPU under
    `python -m torch.distributed.run`.  Called at package import; a no-op without LOCAL_RANK or without a GPU."""
    if "LOCAL_RANK" in os.environ and torch.cuda.is_available():
        n = torch.cuda.device_count()
        if local_rank() >= n and os.environ.get("COLDDIFF_SHARE_GPU", "0") != "1":
            # (RCCL rejects two ranks on one device with an opaque error; COLDDIFF_SHARE_GPU=1 is the test suite's gloo-on-one-GPU run)
            raise RuntimeError(f"LOCAL_RANK={local_rank()} but only {n} GPU(s) are visible: launch one process per GPU "
                               f"(--nproc-per-node <= {n}), or set COLDDIFF_SHARE_GPU=1 to share devices under the gloo backend")
        torch.cuda.set_device(local_rank() % max(1, n))


@contextlib.contextmanager
def _stdout_to_stderr():
    """gloo announces every new group on the PROCESS's stdout ("[Gloo] Rank 0 is connected to ...") from C++; a launcher that reads one
    JSON line from rank 0's stdout (bench.py's contract) must not find that there: file descriptor 1 points at stderr for the duration."""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    if world_size() == 1 or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        assert torch.cuda.device_count() >= lws, (f"RCCL needs one GPU per local rank: LOCAL_WORLD_SIZE={lws}, "
                                                  f"{torch.cuda.device_count()} visible")
        pin_device()
    # default watchdog for the training collectives (a crashed rank / mismatched bucket sequence must surface in minutes); the
    # milestone wait, where rank 0 runs the full T-step sampler + image / checkpoint I/O, goes through milestone_barrier()
    with _stdout_to_stderr():
        dist.init_process_group(backend=backend, rank=rank(), world_size=world_size(),
                                timeout=datetime.timedelta(minutes=int(os.environ.get("COLDDIFF_DIST_TIMEOUT_MIN", "10"))))
        # the long-wait group is created HERE, while every rank is at the same point: created lazily inside the first milestone, its
        # rendezvous itself ran under maximal rank skew (rank 0 still sampling) against the short default timeout
        global _milestone_group
        _milestone_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=2))


_milestone_group = None


def milestone_barrier():
    """Barrier with a 2-hour limit on its own gloo group (host-side wait: no device collective sits in a stream meanwhile).
    EVERY rank-0-only long phase that is followed by a collective must end in this barrier -- Trainer milestones (sampling + checkpoint),
    EvalMixin FID / sampling runs under torchrun, a DeviceImageCache build before the initial broadcast -- because the training
    collectives' own watchdog is COLDDIFF_DIST_TIMEOUT_MIN (10 minutes)."""
    global _milestone_group
    if world_size() == 1 or not dist.is_initialized():
        return
    if _milestone_group is None:                  # (a process group somebody else initialised: all ranks reach their first barrier together)
        with _stdout_to_stderr():
            _milestone_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=2))
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dist.barrier(group=_milestone_group)


def decorrelate_rng():
    """Every rank draws its own timesteps / noise / colours / dropout masks (under DataParallel each replica
    draws its own t, DEBLUR:980).  Called by the Trainer AFTER the replicas' weights are equal."""
    if world_size() > 1:
        torch.manual_seed((torch.initial_seed() + 7919 * rank()) % (1 << 63))


def make_buckets(sizes, per):
    """Greedy runs of WHOLE tensors (arena order), a run is closed once it holds >= `per` elements: no tensor ever crosses
    a bucket edge, so a bucket is final exactly when every tensor in it is.  Returns [(first_tensor, end_tensor)]."""
    out, first, acc = [], 0, 0
    for i, n in enumerate(sizes):
        acc += n
        if acc >= per:
            out.append((first, i + 1))
            first, acc = i + 1, 0
    if first < len(sizes):
        out.append((first, len(sizes)))
    return out


class GradSync:
    """Bucketed, overlapped sum-all-reduce of a FlatArena's gradient buffer.

    * Buckets are runs of whole parameter tensors (`make_buckets`).
    * Readiness is counted per USE: every autograd node announces in forward which parameters its backward will write
      (`used`), and in backward that it has enqueued that contribution (`ready`); a bucket is final when every announced
      use of every tensor in it has reported.  A module that runs twice per loss is therefore waited for twice, a parameter
      the loss does not reach (the blur kernels, DEBLUR:355-359) is final from the start.
    * Collectives are ISSUED in one fixed order on every rank (last bucket first = the order backward finishes them in),
      whatever order readiness arrives in: RCCL requires identical call sequences on all ranks.
    """

    def __init__(self, arena, bucket_bytes=BUCKET_BYTES):
        self.arena = arena
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on_gpu = arena.grad.is_cuda
        sizes = [b - a for a, b in zip(arena.offsets, arena.offsets[1:] + [arena.numel])]        # padded slot sizes
        self.groups = make_buckets(sizes, max(1, bucket_bytes // 4))
        ends = arena.offsets[1:] + [arena.numel]
        self.bounds = [(arena.offsets[i], ends[j - 1]) for i, j in self.groups]
        self.bucket_of = {}
        for b, (i, j) in enumerate(self.groups):
            for p in arena.params[i:j]:
                self.bucket_of[id(p)] = b
        self.order = list(range(len(self.bounds) - 1, -1, -1))
        self.resident_reserve, self.resident_reserve_source = 0, "single rank / CPU: resident grids take every CU"
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        if self.world > 1 and self.on_gpu:
            # The resident ("streaming") GEMM blocks hold every CU for a whole launch with a FIXED share of the tiles each; the collective
            # kernels of the bucket all-reduces run concurrently and need CUs of their own, and a resident block that finds its CU taken
            # would run its whole share after everybody else (the launch then takes up to twice as long).  Multi-rank training therefore
            # runs the SAME kernels as N = 1 with a few CUs left out of the resident grids (cdf_gemm_tuning.resident_reserve; RCCL's
            # channels are one workgroup each, 32 covers its default channel count on this part) -- round 3 switched the resident form
            # off instead, i.e. N > 1 ran other GEMM kernels than the N = 1 line.
            self.resident_reserve, self.resident_reserve_source = resident_reserve()
            rt.tuning().set(resident_reserve=self.resident_reserve)
        self.armed = False
        self.uses = [0] * len(self.bounds)
        self.pending = None
        self.head = 0
        self.works = []
        # optional timing (bench.py): per step (backward-end event on the compute stream, [(start, end) per bucket on the comm stream])
        self.profile = False
        self._prof, self._cur = [], None

    # -- forward side ---------------------------------------------------------------------------------
    def begin(self):
        """Start of a micro-step's forward pass."""
        self.uses = [0] * len(self.bounds)

    def used(self, params):
        for p in params:
            b = self.bucket_of.get(id(p))
            if b is not None:
                self.uses[b] += 1

    # -- backward side --------------------------------------------------------------------------------
    def arm(self):
        """Call after the forward and before the backward pass of the LAST accumulation micro-step."""
        if self.world == 1:
            return
        self.armed = True
        self.pending = list(self.uses)
        self.head = 0
        self.works = []
        self._cur = [] if (self.profile and self.on_gpu) else None
        self._drain()

    def _launch(self, b):
        lo, hi = self.bounds[b]
        buf = self.arena.grad[lo:hi]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self._cur is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
                if self._cur is not None:
                    self.works[-1].wait()                       # (stream-side wait: orders e1 behind the collective on the comm stream)
                    e1.record(self.comm_stream)
                    self._cur.append((e0, e1))
        else:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    def _drain(self, force=False):
        while self.head < len(self.order) and (force or self.pending[self.order[self.head]] == 0):
            self._launch(self.order[self.head])
            self.head += 1

    def ready(self, params):
        """One announced use of each of `params` has its backward kernels enqueued."""
        if not self.armed:
            return
        for p in params:
            b = self.bucket_of.get(id(p))
            if b is not None:
                self.pending[b] -= 1
                assert self.pending[b] >= 0, "a backward node reported gradients it never announced in forward"
        self._drain()

    def finish(self):
        """After backward returned: reduce whatever is left and make the compute stream wait."""
        if not self.armed:
            return
        self._drain(force=True)        # (uses whose backward never ran, e.g. a detached branch: final now that backward is over)
        if self._cur is not None:
            bwd_end = torch.cuda.Event(enable_timing=True)
            bwd_end.record(torch.cuda.current_stream())
            self._prof.append((bwd_end, self._cur))
            self._cur = None
        for w in self.works:
            w.wait()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.armed = False


    def stats(self):
        """Timing of the profiled steps (call after a device synchronize): total all-reduce time on the comm stream, and the part of
        it that ran AFTER the backward pass had finished on the compute stream (= exposed; the rest was hidden under backward)."""
        if not self._prof:
            return None
        comm = exposed = 0.0
        for bwd_end, evs in self._prof:
            comm += sum(a.elapsed_time(b) for a, b in evs)
            exposed += max(0.0, bwd_end.elapsed_time(evs[-1][1]))       # last bucket's end vs the end of backward
        n = len(self._prof)
        nbytes = 4 * sum(hi - lo for lo, hi in self.bounds)
        return {"buckets": len(self.bounds), "bytes_per_step": nbytes, "allreduce_ms_per_step": round(comm / n, 3),
                "exposed_ms_per_step": round(exposed / n, 3), "hidden_ms_per_step": round(max(0.0, comm - exposed) / n, 3),
                "busbw_gbs": round(nbytes * 2 * (self.world - 1) / self.world / (comm / n * 1e-3) / 1e9, 1) if comm > 0 else None,
                "steps": n, "resident_reserve_cus": self.resident_reserve, "resident_reserve_from": self.resident_reserve_source}


RESERVE_MAX = 248          # cdf_gemm_tuning.resident_reserve's range (include/colddiff.h): at least one XCD round of CUs stays in the grid


def resident_reserve(env=None):
    """CUs the resident GEMM grids leave free for the collective kernels that run beside backward -> (count, where it came from).
    An RCCL channel is one workgroup, and a workgroup that finds every CU held by a resident block waits for a whole launch -- so the
    reserve is RCCL's channel count, in whole rounds of the 8 XCDs: COLDDIFF_RESIDENT_RESERVE when set (validated HERE, with the
    variable's name in the error, instead of failing every GEMM call with the library's generic `bad cdf_gemm_tuning`), else
    NCCL_MAX_NCHANNELS (the user's cap on the channel count -- RCCL reads the NCCL_* names), else NCCL_MIN_NCHANNELS if that is larger than
    the default, else 32 (RCCL's default upper channel count per collective on this part)."""
    env = os.environ if env is None else env

    def num(name):
        v = env.get(name)
        if v is None or v == "":
            return None
        try:
            return int(v)
        except ValueError:
            raise ValueError(f"{name}={v!r}: expected an integer") from None

    v = num("COLDDIFF_RESIDENT_RESERVE")
    if v is not None:
        if not 0 <= v <= RESERVE_MAX:
            raise ValueError(f"COLDDIFF_RESIDENT_RESERVE={v}: the resident GEMM grids can leave 0..{RESERVE_MAX} CUs free")
        return v, "COLDDIFF_RESIDENT_RESERVE"
    cap, floor = num("NCCL_MAX_NCHANNELS"), num("NCCL_MIN_NCHANNELS")
    if cap is not None and cap > 0:
        return min(RESERVE_MAX, (max(cap, floor or 0) + 7) // 8 * 8), "NCCL_MAX_NCHANNELS"
    if floor is not None and floor > 32:
        return min(RESERVE_MAX, (floor + 7) // 8 * 8), "NCCL_MIN_NCHANNELS"
    return 32, "default (RCCL's default channel count)"


def set_engine(e):
    global _engine
    _engine = e


def grads_used(params):
    if _engine is not None:
        _engine.used(params)


def grads_ready(params):
    if _engine is not None:
        _engine.ready(params)
