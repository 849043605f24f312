"""The CPU oracle (oracle/cold_oracle.py) is pinned against the reference itself:
  * always: against the committed golden vectors the unmodified reference produced
    (tests/golden/make_golden.py), and the torchgeometry probe values recorded in SURVEY.md §8(c)
  * in the build container (where /root/reference exists): live, bit-for-bit on CPU.
"""
import os

import pytest
import torch

from oracle import cold_oracle as O
from oracle import ref_shim

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def test_gaussian_kernel_probe_values():
    k = O.gaussian_kernel2d((11, 11), (7.0, 7.0))
    assert abs(k[5, 5].item() - 0.0100549823) < 1e-9 and abs(k[0, 0].item() - 0.0060367407) < 1e-9
    assert abs(O.gaussian_kernel2d((15, 15), (1.0, 1.0))[7, 7].item() - 0.1591549516) < 2e-8
    assert abs(O.gaussian_kernel2d((11, 11), (0.35, 0.35))[5, 5].item() - 0.9357516) < 1e-6


def test_unet_matches_golden():
    g = load("unet_dim8.pt")
    ps = {k: v.clone().requires_grad_() for k, v in g["sd"].items()}
    y = O.unet_forward(ps, g["x"], g["t"])
    assert torch.equal(y, g["y"])
    y.backward(g["gy"])
    for k, ref in g["grads"].items():
        assert (ps[k].grad - ref).abs().max() <= 1e-6 * max(1.0, ref.abs().max().item()), k


def test_model_matches_golden():
    g = load("model_ch32.pt")
    ps = {k: v.clone().requires_grad_() for k, v in g["sd"].items()}
    y = O.model_forward(ps, g["x"], g["t"], num_res_blocks=1, num_resolutions=2)
    assert (y - g["y"]).abs().max() <= 1e-6
    y.backward(g["gy"])
    for k, ref in g["grads"].items():
        assert (ps[k].grad - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()), k


def test_degradations_and_samplers_match_golden():
    g = load("diffusion.pt")
    sd = g["deblur/net_sd"]
    net = lambda im, st: O.unet_forward(sd, im, st)
    for key, c in g.items():
        if key.startswith("deblur/") and key != "deblur/net_sd":
            _, routine, sampling = key.split("/")
            sig = O.blur_sigmas(routine, c["T"], c["ks"], c["std"])
            for i, (k, s, mode) in enumerate(sig):         # the restated generator reproduces the reference's kernels
                w = O.gaussian_kernel2d((k, k), (s, s))[None, None].repeat(3, 1, 1, 1)
                assert torch.equal(w, c["kernels"][i]) and mode == c["modes"][i]
            assert torch.equal(O.blur_q_sample(c["x"], c["t"], c["kernels"], c["modes"], c["T"]), c["q"])
            _, _, img = O.cold_sample(net, lambda z, i: O.blur_step(z, c["kernels"][i], c["modes"][i]), c["x"], c["T"], sampling)
            assert (img - c["img"]).abs().max() <= 1e-6, key
        elif key.startswith("denoise/"):
            ca, cb = O.cosine_tables(c["T"])
            assert torch.equal(ca, c["ca"]) and torch.equal(cb, c["cb"])
            assert torch.equal(O.noise_q_sample(c["x"], c["eps"], c["t"], ca, cb), c["q"])
            fixed = key.endswith("x0_step_down")
            assert (O.noise_sample(net, c["eps"], c["T"], ca, cb, fixed)[2] - c["gen"]).abs().max() <= 1e-6
            assert (O.noise_sample(net, c["eps"], c["T"], ca, cb, False)[2] - c["sample"]).abs().max() <= 1e-6
        elif key.startswith("resolution/"):
            routine = key.split("/")[1]
            mode = "area" if "_area" in routine else ("bilinear" if "_bilinear" in routine else "bicubic")
            sizes = O.pixelate_sizes(routine, c["T"], 16)
            assert torch.equal(O.pixelate_q_sample(c["x"], c["t"], sizes, mode), c["q"])
            _, _, img = O.cold_sample(net, lambda z, i: O.pixelate_step(z, sizes[i], mode), c["x"], c["T"], "x0_step_down")
            assert (img - c["img"]).abs().max() <= 1e-6, key
        elif key.startswith("defade/"):
            masks = O.fade_kernels("Incremental", c["T"], 16, 0.6, 1)
            assert torch.equal(masks, c["masks"])
            assert torch.equal(O.fade_q_sample(c["x"], c["t"], masks), c["q"])
            _, _, img = O.cold_sample(net, lambda z, i: masks[i] * z, c["x"], c["T"], key.split("/")[1])
            assert (img - c["img"]).abs().max() <= 1e-6, key


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree only exists in the build container")
def test_oracle_bit_exact_vs_live_reference():
    ref = ref_shim.load("deblurring")
    torch.manual_seed(7)
    net = ref.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=1)
    x, t = torch.randn(2, 1, 32, 32), torch.tensor([5, 11])
    with torch.no_grad():
        assert torch.equal(net(x, t), O.unet_forward(net.state_dict(), x, t))
    m = ref.Model(resolution=16, in_channels=3, out_ch=3, ch=32, ch_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(8,), dropout=0.1).eval()
    x = torch.randn(2, 3, 16, 16)
    with torch.no_grad():
        assert (m(x, t) - O.model_forward(m.state_dict(), x, t, num_res_blocks=2, num_resolutions=3)).abs().max() <= 1e-6
