"""Where do the ~137 `copyBuffer` dispatches per optimizer step come from?  One profiled train step (torch.profiler, CPU + device activities,
Python stacks): every aten op that ends in a device memcpy / memset, with its innermost colddiff / bench call site.
   python tools/copy_sites.py [--precision bf16]"""
import argparse
import collections
import contextlib
import io
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    from colddiff import runtime
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    if a.precision:
        runtime.set_precision(a.precision)
    dev = torch.device("cuda:0")
    torch.manual_seed(123457)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
        d = GaussianDiffusion(net, image_size=128, channels=3, timesteps=200, loss_type='l1', sampling_routine='x0_step_down').to(dev)
        tr = Trainer(d, None, image_size=128, train_batch_size=32, train_lr=2e-5, train_num_steps=10 ** 9, gradient_accumulate_every=2,
                     dataset='synthetic', results_folder=os.path.join(REPO, "gpurun_out", "bench_results"))
    tr.quiet = True
    for _ in range(3):
        tr.train_step()
        tr.step += 1
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.train_step()
        tr.step += 1
        torch.cuda.synchronize()
    ops = collections.Counter()
    for e in prof.events():
        n = e.name
        if not n.startswith("aten::") or not any(k in n for k in ("copy", "clone", "contiguous", "fill", "zero", "cat", "to_copy", "add", "mul")):
            continue
        st = [s_ for s_ in (e.stack or []) if "colddiff" in s_ or "bench" in s_ or "torch/autograd" in s_]
        ops[(n, st[0].split("/")[-1] if st else "?", tuple(e.input_shapes[0]) if e.input_shapes else None)] += 1
    for (n, where, shp), c in sorted(ops.items(), key=lambda kv: -kv[1])[:40]:
        print("%4d %-28s %s %s" % (c, n, where, shp))
    dev_ev = collections.Counter()
    for e in prof.events():
        if "Memcpy" in e.name or "Memset" in e.name or "copyBuffer" in e.name:
            dev_ev[e.name] += 1
    print(dict(dev_ev))


if __name__ == "__main__":
    main()
