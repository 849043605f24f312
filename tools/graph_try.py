"""Can a whole sampler loop be captured into one hipGraph through torch.cuda.graph?  (experiment)"""
import contextlib, io, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "cold-diffusion-models_amd"), REPO]
import torch
from denoising_diffusion_pytorch import GaussianDiffusion, Unet
dev = torch.device("cuda:0")
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
T = int(os.environ.get("T", "200"))
d = GaussianDiffusion(net, image_size=128, channels=3, timesteps=T, sampling_routine="x0_step_down").to(dev)
for B in (1, 4, 16):
    noise = torch.randn(B, 3, 128, 128, device=dev)
    with torch.no_grad():
        d.gen_sample(batch_size=B, img=noise, t=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); _, _, ref = d.gen_sample(batch_size=B, img=noise); torch.cuda.synchronize(); te = time.perf_counter() - t0
        static = noise.clone()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            d.gen_sample(batch_size=B, img=static, t=2)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            _, _, out = d.gen_sample(batch_size=B, img=static)
        torch.cuda.synchronize(); tc = time.perf_counter() - t0
        static.copy_(noise)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    print(f"B={B}: eager {1000*te/B:.1f} ms/img, graph {1000*tg/B:.1f} ms/img (capture {tc:.1f}s), max|diff| {(out-ref).abs().max().item():.2e}", flush=True)
    del g
