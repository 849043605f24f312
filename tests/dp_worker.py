"""Worker for tests/test_dp.py: one rank of a multi-process gloo data-parallel run on the CPU simulator.

argv: out_path nsteps bucket_bytes mode [hip|emu|rccl] [unet|model] [precision] [fused]
  "hip": both ranks on cuda:0 with the product library, gloo moving CUDA tensors;  "rccl": rank r on cuda:r, RCCL;  "model": the CIFAR `Model` (MODEL2:191-332)
  instead of `Unet`;  precision: bf16x3 (default) | bf16 | f32
  mode 'once'  : loss = L1(x, f(q(x, e, t), t))                         (the denoising package's p_losses)
  mode 'twice' : the network runs TWICE per loss (as RESOL:702-716 'Final_random_mean_and_actual' does):
                 loss = L1(x, f(q(x,e,t), t)) + L1(x, f(q(x,-e,t), t))
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), HERE, REPO):
    sys.path.insert(0, p)

import torch  # noqa: E402
from emu_util import install_emu  # noqa: E402

RCCL = len(sys.argv) > 5 and sys.argv[5] == "rccl"      # one GPU per rank, RCCL carrying the buckets (needs >= 2 devices)
ON_HIP = len(sys.argv) > 5 and sys.argv[5] in ("hip", "rccl")
DEV = "cuda:%d" % (int(os.environ.get("LOCAL_RANK", "0")) if RCCL else 0)
if not ON_HIP:
    install_emu()
from colddiff import parallel, runtime  # noqa: E402
from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet  # noqa: E402
from colddiff.model2 import Model  # noqa: E402

NET = sys.argv[6] if len(sys.argv) > 6 else "unet"
if len(sys.argv) > 7:
    runtime.set_precision(sys.argv[7])

out_path, nsteps, bucket_bytes, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
if bucket_bytes > 0:
    parallel.BUCKET_BYTES = bucket_bytes
    parallel.GradSync.__init__.__defaults__ = (bucket_bytes,)
parallel.init_distributed("nccl" if RCCL else "gloo")
rank, world = parallel.rank(), parallel.world_size()
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    if NET == "model":
        net = Model(ch=32, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(4,), dropout=0.0, in_channels=3, resolution=8)
    else:
        net = Unet(dim=8, dim_mults=(1, 2), channels=3)
if rank >= 1:      # ranks must converge to rank 0's weights through the initial broadcast
    with torch.no_grad():
        for p in net.parameters():
            p.add_(float(rank))
if ON_HIP:
    net = net.to(DEV)
diff = GaussianDiffusion(net, image_size=8, channels=3, timesteps=10)
if ON_HIP:
    diff = diff.to(DEV)
tr = Trainer(diff, None, image_size=8, train_batch_size=2, train_lr=1e-3, train_num_steps=nsteps, gradient_accumulate_every=2,
             dataset="synthetic", results_folder=os.path.join(os.path.dirname(out_path), f"res{rank}"))
# the per-rank RNG streams must differ after construction (every rank draws its own t / noise)
seeds = [None] * world
torch.distributed.all_gather_object(seeds, torch.initial_seed())
assert len(set(seeds)) == world, seeds
# no tensor may cross a bucket edge; record how many buckets were used and when each was launched
sync = tr.sync
edges = {lo for lo, _ in sync.bounds} | {hi for _, hi in sync.bounds}
for p, o in zip(tr.arena.params, tr.arena.offsets):
    assert not any(o < e < o + p.numel() for e in edges), "a parameter straddles a bucket edge"
launch_log = []
orig_launch = sync._launch


def logged_launch(b):
    # at launch time every announced use of the bucket must have reported (or backward is over)
    launch_log.append((b, sync.pending[b]))
    orig_launch(b)


sync._launch = logged_launch
early = []                      # buckets already issued when backward returns (= overlapped with the backward pass)
orig_finish = sync.finish


def logged_finish():
    early.append(len(launch_log) - len(early) * len(sync.bounds))      # every step issues each bucket exactly once
    orig_finish()


sync.finish = logged_finish
g = torch.Generator().manual_seed(1)
# all ranks generate the full schedule and take their own shard: batches[step][rank][micro]
batches = [[[(torch.rand(2, 3, 8, 8, generator=g) * 2 - 1, torch.randn(2, 3, 8, 8, generator=g), torch.randint(0, 10, (2,), generator=g))
             for _ in range(2)] for _ in range(world)] for _ in range(nsteps)]


def loss_of(x, e, t):
    if ON_HIP:
        x, e, t = x.to(DEV), e.to(DEV), t.to(DEV)
    if mode == "once":
        return tr.core.p_losses(x, e, t)
    return tr.core.p_losses(x, e, t) + tr.core.p_losses(x, -e, t)


FUSED = len(sys.argv) > 8 and sys.argv[8] == "fused"      # the rank's two micro-batches as ONE pass (Trainer's fused accumulation)
for s in range(nsteps):
    it = iter(batches[s][rank])
    if FUSED:
        assert mode == "once"

        def micro(it=it):
            x, e, t = next(it)
            if ON_HIP:
                x, e, t = x.to(DEV), e.to(DEV), t.to(DEV)
            return tr.core.prepare(x, e, t=t)
        tr._prepare_micro = micro
        assert tr._can_fuse()
    else:
        tr._loss = lambda batch, it=it: loss_of(*next(it))
    tr.train_step()
    tr.step += 1
assert all(pend == 0 for _, pend in launch_log), launch_log
assert [b for b, _ in launch_log[:len(sync.bounds)]] == sync.order          # same issue order on every rank
if ON_HIP:
    torch.cuda.synchronize()
torch.save({"sd": {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, "buckets": len(sync.bounds),
            "max_uses": max(sync.uses), "early": min(early), "order": [b for b, _ in launch_log]}, out_path + f".rank{rank}")
torch.distributed.barrier()
