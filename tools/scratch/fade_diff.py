import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "cold-diffusion-models_amd"), REPO]
from defading_diffusion_pytorch import GaussianDiffusion
g = torch.load(os.path.join(REPO, "tests/golden/fullsize.pt"), weights_only=False)
for key, c in g.items():
    d = GaussianDiffusion(torch.nn.Identity(), image_size=128, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=c["kernel_std"],
                          initial_mask=c["initial_mask"], fade_routine="Random_Incremental", discrete=key.endswith("/1"))
    d._offsets = lambda b, dev, c=c: (c["rand_x"].to(dev), c["rand_y"].to(dev))
    x = (c["levels"].float() / 255 * 2 - 1).cuda()
    with torch.no_grad():
        q = d.q_sample(x, c["t"].cuda()).cpu()
    ne = (q != c["q"]).nonzero()
    print(key, "mismatches", ne.shape[0])
    for idx in ne[:20].tolist():
        a, b = q[tuple(idx)].item(), c["q"][tuple(idx)].item()
        print("  ", idx, "gpu %.9e" % a, "ref %.9e" % b)
    # torch's own GPU elementwise product as a second opinion
    m = d.fade_kernels.cuda()
    z = x[0:1].clone()
    rx, ry = int(c["rand_x"][0]), int(c["rand_y"][0])
    for i in range(int(c["t"][0]) + 1):
        z = m[i][rx:rx + 128, ry:ry + 128] * z
    print("   torch-on-gpu vs ref mismatches (sample 0, before quantise):", (z.cpu()[0] != c["q"][0]).sum().item() if not key.endswith("/1") else "n/a")
