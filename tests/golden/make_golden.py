"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference
(/root/reference, imported through oracle/ref_shim.py) on CPU with seed 123457
(deblurring-diffusion-pytorch/celebA_128_test.py:14).  Build-container only; the .pt fixtures it
writes are committed and travel to the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import ref_shim  # noqa: E402

SEED = 123457


def images(B, C, H, g):
    return torch.randint(0, 256, (B, C, H, H), generator=g).float() / 255 * 2 - 1


def unet_case(ref):
    torch.manual_seed(SEED)
    net = ref.Unet(dim=8, dim_mults=(1, 2, 4), channels=3)
    g = torch.Generator().manual_seed(SEED)
    x, t = images(2, 3, 16, g), torch.tensor([3, 17])
    y = net(x, t)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    return {"cfg": dict(dim=8, dim_mults=(1, 2, 4), channels=3), "sd": {k: v.clone() for k, v in net.state_dict().items()},
            "x": x, "t": t, "y": y.detach(), "gy": gy, "grads": grads}


def model_case(ref):
    torch.manual_seed(SEED)
    cfg = dict(resolution=8, in_channels=3, out_ch=3, ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(4,), dropout=0.0)
    net = ref.Model(**cfg)
    g = torch.Generator().manual_seed(SEED)
    x, t = images(2, 3, 8, g), torch.tensor([0, 9])
    y = net(x, t)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    return {"cfg": cfg, "sd": {k: v.clone() for k, v in net.state_dict().items()}, "x": x, "t": t, "y": y.detach(), "gy": gy,
            "grads": {k: p.grad.clone() for k, p in net.named_parameters()}}


def diffusion_cases():
    out = {}
    g = torch.Generator().manual_seed(SEED)
    # ---- deblurring ------------------------------------------------------------------------------
    ref = ref_shim.load("deblurring")
    torch.manual_seed(SEED)
    net = ref.Unet(dim=8, dim_mults=(1, 2), channels=3)
    for routine, ks, std in (("Incremental", 3, 0.4), ("Constant", 5, 1.0), ("Exponential_reflect", 5, 0.2), ("Special_6_routine", 11, 0)):
        for sampling in ("default", "x0_step_down"):
            T = 4
            d = ref.GaussianDiffusion(net, image_size=16, device_of_kernel="cpu", channels=3, timesteps=T, kernel_std=std, kernel_size=ks,
                                      blur_routine=routine, sampling_routine=sampling)
            x, t = images(3, 3, 16, g), torch.tensor([0, 3, 2])
            with torch.no_grad():
                xq = d.q_sample(x, t)
                xt, direct, img = d.sample(batch_size=3, img=x)
            out[f"deblur/{routine}/{sampling}"] = dict(T=T, ks=ks, std=std, x=x, t=t, q=xq, xt=xt, direct=direct, img=img,
                                                       kernels=[m.weight.detach().clone() for m in d.gaussian_kernels],
                                                       modes=[m.padding_mode for m in d.gaussian_kernels])
    out["deblur/net_sd"] = {k: v.clone() for k, v in net.state_dict().items()}
    # ---- denoising ------------------------------------------------------------------------------------
    ref = ref_shim.load("denoising")
    for sampling in ("x0_step_down", "ddim"):
        T = 5
        d = ref.GaussianDiffusion(net, image_size=16, channels=3, timesteps=T, sampling_routine=sampling)
        x, eps, t = images(3, 3, 16, g), torch.randn(3, 3, 16, 16, generator=g), torch.tensor([0, 4, 2])
        with torch.no_grad():
            xq = d.q_sample(x, eps, t)
            n1, d1, i1 = d.gen_sample(batch_size=3, img=eps)
            n2, d2, i2 = d.sample(batch_size=3, img=eps)
        out[f"denoise/{sampling}"] = dict(T=T, x=x, eps=eps, t=t, q=xq, gen=i1, sample=i2, ca=d.sqrt_alphas_cumprod.clone(),
                                          cb=d.sqrt_one_minus_alphas_cumprod.clone())
    # ---- resolution -----------------------------------------------------------------------------------
    ref = ref_shim.load("resolution")
    for routine in ("Incremental", "Incremental_area", "Incremental_factor_2", "Incremental_area_factor_2", "Incremental_bilinear_factor_2"):
        T = 3
        d = ref.GaussianDiffusion(net, image_size=16, device_of_kernel="cpu", channels=3, timesteps=T, resolution_routine=routine,
                                  sampling_routine="x0_step_down")
        x, t = images(3, 3, 16, g), torch.tensor([0, 2, 1])
        with torch.no_grad():
            xq = d.q_sample(x, t)
            xt, direct, img = d.sample(batch_size=3, img=x)
        out[f"resolution/{routine}"] = dict(T=T, x=x, t=t, q=xq, xt=xt, img=img)
    # ---- defading -------------------------------------------------------------------------------------
    ref = ref_shim.load("defading")
    for sampling in ("default", "x0_step_down"):
        T = 5
        d = ref.GaussianDiffusion(net, image_size=16, device_of_kernel="cpu", channels=3, timesteps=T, kernel_std=0.6, initial_mask=1,
                                  fade_routine="Incremental", sampling_routine=sampling)
        x, t = images(3, 3, 16, g), torch.tensor([0, 4, 2])
        with torch.no_grad():
            xq = d.q_sample(x, t)
            xt, direct, img = d.sample(batch_size=3, faded_recon_sample=x)
        out[f"defade/{sampling}"] = dict(T=T, x=x, t=t, q=xq, xt=xt, img=img, masks=d.fade_kernels.clone())
    return out


def main():
    assert ref_shim.available(), "needs /root/reference (build container)"
    ref = ref_shim.load("deblurring")
    torch.save(unet_case(ref), os.path.join(HERE, "unet_dim8.pt"))
    torch.save(model_case(ref), os.path.join(HERE, "model_ch32.pt"))
    torch.save(diffusion_cases(), os.path.join(HERE, "diffusion.pt"))
    # torchgeometry boundary: values observed when the reference builds its kernels through the shim
    k = ref_shim.get_gaussian_kernel2d((11, 11), (7.0, 7.0))
    print("k=11 sigma=7 centre %.10f corner %.10f" % (k[5, 5].item(), k[0, 0].item()))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
