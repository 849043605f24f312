"""Does the bf16-storage engine TRAIN like the parity-grade one?  The same 40 optimizer steps (CelebA-128 denoising, 2 x 32 synthetic images
per step, same seed => same data, t and noise draws) in the parity-grade mode, in the bf16 mode with fp32 tensors (rounds 2-4) and in the
bf16 mode with bf16 activation storage (round 5); prints the loss of every 4th step and the largest relative difference of the three
weight arenas after the last step.   python tools/bf16_trajectory.py"""
import contextlib
import io
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402


def run(precision, storage, steps=40, lr=2e-5):
    from colddiff import runtime
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    os.environ["COLDDIFF_BF16_STORAGE"] = storage
    runtime.set_precision(precision)
    runtime.bump_weights_epoch()
    dev = torch.device("cuda:0")
    torch.manual_seed(123457)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
        d = GaussianDiffusion(net, image_size=128, channels=3, timesteps=200, loss_type='l1', sampling_routine='x0_step_down').to(dev)
        tr = Trainer(d, None, image_size=128, train_batch_size=32, train_lr=lr, train_num_steps=10 ** 9, gradient_accumulate_every=2,
                     dataset='synthetic', results_folder=os.path.join(REPO, "gpurun_out", "bench_results"))
    tr.quiet = True
    torch.manual_seed(7)
    losses = []
    for _ in range(steps):
        losses.append(tr.train_step())
        tr.step += 1
    torch.cuda.synchronize()
    return [float(v) for v in losses], tr.arena.data.clone()


def main():
    out = {}
    for name, prec, sto in (("parity-grade (bf16x3)", "bf16x3", "1"), ("bf16, fp32 tensors", "bf16", "0"), ("bf16, bf16 activation storage", "bf16", "1")):
        out[name] = run(prec, sto)
    names = list(out)
    print("step  " + "  ".join("%-30s" % n for n in names))
    for i in range(0, len(out[names[0]][0]), 4):
        print("%4d  " % i + "  ".join("%-30.6f" % out[n][0][i] for n in names))
    ref = out[names[0]][1]
    for n in names[1:]:
        dlt = (out[n][1] - ref).abs()
        print("%s vs parity-grade after %d steps: largest weight difference %.3e (lr %.0e: Adam moves a weight by ~lr per step), mean %.3e; largest loss difference %.2e"
              % (n, len(out[n][0]), dlt.max().item(), 2e-5, dlt.mean().item(), max(abs(a - b) for a, b in zip(out[n][0], out[names[0]][0]))))


if __name__ == "__main__":
    main()
