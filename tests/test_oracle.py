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


def _close_lists(a, b, tol=1e-6):
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert (u - v).abs().max() <= tol


def test_sampler_variants_match_golden():
    """forward_and_backward(_2), sample_from_blur, all_sample, gen_sample(times) of the reference."""
    g = load("variants.pt")
    sd = load("diffusion.pt")["deblur/net_sd"]
    net = lambda im, st: O.unet_forward(sd, im, st)
    for key, c in g.items():
        if key.startswith("deblur/"):
            sampling = key.split("/")[2]
            step = lambda z, i: O.blur_step(z, c["kernels"][i], c["modes"][i])
            F1, B1, i1 = O.cold_forward_and_backward(net, step, c["x"], c["T"], sampling)
            _close_lists(F1, c["fab"][0]), _close_lists(B1, c["fab"][1]), _close_lists([i1], [c["fab"][2]])
            F2, Ba, Bb, ia, ib = O.blur_forward_and_backward_2(net, step, c["x"], c["T"])
            _close_lists(F2, c["fab2"][0]), _close_lists(Ba, c["fab2"][1]), _close_lists(Bb, c["fab2"][2])
            _close_lists([ia, ib], list(c["fab2"][3:]))
            _close_lists(list(O.cold_sample_from(net, step, c["half"], c["T"], sampling, start=2)), list(c["from_blur"]))
            X0, Xt, _ = O.cold_all_sample(net, step, c["x"], c["T"], sampling, times=3)
            _close_lists(X0, c["all_sample"][0][:3]), _close_lists(Xt, c["all_sample"][1])
        elif key == "denoise/fab":
            ca, cb = O.cosine_tables(c["T"])
            F1, B1, i1 = O.noise_forward_and_backward(net, c["x"], c["noise"], c["T"], ca, cb)
            _close_lists(F1, c["fab"][0]), _close_lists(B1, c["fab"][1]), _close_lists([i1], [c["fab"][2]])
        elif key.startswith("resolution/Incremental"):
            _, routine, sampling = key.split("/")
            mode = "area" if "_area" in routine else "bicubic"
            sizes = O.pixelate_sizes(routine, c["T"], 16)
            step = lambda z, i: O.pixelate_step(z, sizes[i], mode)
            X0, Xt, _ = O.cold_all_sample(net, step, c["x"], c["T"], sampling)
            _close_lists(X0, c["all_sample"][0]), _close_lists(Xt, c["all_sample"][1])
            F1, B1, i1 = O.cold_forward_and_backward(net, step, c["x"], c["T"], sampling)
            _close_lists(F1, c["fab"][0]), _close_lists(B1, c["fab"][1]), _close_lists([i1], [c["fab"][2]])
            # gen_sample(times=2): no forward process, two reverse updates from the given image (RESOL:460-505)
            img, direct = c["x"], None
            for tt in (2, 1):
                x = net(img, torch.full((2,), tt - 1, dtype=torch.long))
                direct = x if direct is None else direct
                img = O._reverse_update(step, img, x, tt, sampling)
            _close_lists([c["x"], direct, img], list(c["gen_times2"]))


def test_resolution_train_routines_match_golden():
    """RESOL:655-760 incl. the `t - 1 = -1` indexing of 'Step' (SURVEY §8 row A6)."""
    g = load("variants.pt")
    sd0 = load("diffusion.pt")["deblur/net_sd"]
    sizes = O.pixelate_sizes("Incremental_factor_2", 3, 16)
    for key, c in g.items():
        if not key.startswith("resolution/train/"):
            continue
        _, _, tr, loss_type = key.split("/")
        ps = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd0.items()}
        net = lambda im, st: O.unet_forward(ps, im, st)
        loss = O.pixelate_p_losses(net, c["x"], c["t"], sizes, "bicubic", tr, loss_type, noise=c["noise"], new_mean=c["new_mean"])
        assert (loss - c["loss"]).abs() <= 1e-6 * max(1.0, c["loss"].abs().item()), key
        if c["grads"] is not None:
            loss.backward()
            for k, ref in c["grads"].items():
                assert (ps[k].grad - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()), (key, k)
    with pytest.raises(RuntimeError):                                 # 'Step' with every t = 0: the reference stacks an empty list
        O.pixelate_q_sample_ref(g["resolution/train/Step/l1"]["x"], torch.tensor([-1, -1, -1]), sizes, "bicubic")


def test_mixing_packages_match_golden():
    """demixing and defading generation: tables, q_sample, every sampler, loss + gradients."""
    g = load("mixing.pt")
    sd0 = load("diffusion.pt")["deblur/net_sd"]
    net = lambda im, st: O.unet_forward(sd0, im, st)
    c = g["demix"]
    ca, cb = O.cosine_tables(c["T"])
    assert torch.equal(O.noise_q_sample(c["x1"], c["x2"], c["t"], ca, cb), c["q"])
    _close_lists(list(O.noise_sample(net, c["x2"], c["T"], ca, cb, True)), list(c["gen"]))
    _close_lists(list(O.noise_sample(net, c["x2"], c["T"], ca, cb, False)), list(c["sample"]))
    F1, B1, i1 = O.noise_forward_and_backward(net, c["x1"], c["x2"], c["T"], ca, cb)
    _close_lists(F1, c["fab"][0]), _close_lists(B1, c["fab"][1]), _close_lists([i1], [c["fab"][2]])
    for key in ("defgen/0", "defgen/1"):
        c = g[key]
        al, om = O.blend_tables(c["T"], 16, c["kernel_std"], c["initial_mask"], reverse=key.endswith("1"))
        assert torch.equal(al, c["alphas"]) and torch.equal(om, c["one_minus"])
        assert torch.equal(O.blend_q_sample(c["x1"], c["x2"], c["t"], al, om), c["q"])
        direct, img = O.blend_sample(net, c["x2"], c["x2"], c["T"], al, om)
        _close_lists([c["x2"], direct, img], list(c["sample"]))
        _close_lists([c["x2"], direct, img], list(c["gen"]))            # noise_level = 0: the same walk
        X0, Xt = [], []
        O.blend_sample(net, c["x2"], c["x2"], c["T"], al, om, collect=lambda a, b: (X0.append(a), Xt.append(b)))
        _close_lists(X0, c["all_sample"][0]), _close_lists(Xt, c["all_sample"][1])
        F1 = [c["x1"]] + [O.blend_q_sample(c["x1"], c["x2"], torch.full((2,), i, dtype=torch.long), al, om) for i in range(c["T"])]
        _close_lists(F1, c["fab"][0]), _close_lists(Xt, c["fab"][1]), _close_lists([img], [c["fab"][2]])
        ps = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd0.items()}
        loss = O.loss_fn(c["x1"], O.unet_forward(ps, O.blend_q_sample(c["x1"], c["x2"], c["t"], al, om), c["t"]))
        assert (loss - c["loss"]).abs() <= 1e-6
        if c["grads"] is not None:
            loss.backward()
            for k, ref in c["grads"].items():
                assert (ps[k].grad - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()), k


def masks_from_g1(g1):
    """[T, n] normalised 1-D Gaussians -> the fade masks of DEFADE:328-352 ((1 - g g^T / max)[1:, 1:] per step): everything after the
    exp is exactly rounded IEEE arithmetic.  tests/golden/fullsize.pt carries the REFERENCE's own 1-D Gaussians because torch.exp on
    CPU is not correctly rounded and differs by an ulp between vector ISAs (make_golden.py::fullsize_cases)."""
    ks = []
    for g in g1:
        k = torch.matmul(g.unsqueeze(-1), g.unsqueeze(-1).t())
        ks.append((torch.ones_like(k) - k / torch.max(k))[1:, 1:])
    return torch.stack(ks)


def test_fullsize_random_fade_matches_golden():
    """Defading 'Random_Incremental' (+- discrete) at 128 x 128 with the README schedule (README.md:125-126), reference-generated
    (make_golden.py::fullsize_cases): the restated masks, q_sample with the replayed crop offsets (bit-exact) and the six-step
    Algorithm-2 walk (DEFADE:354-425)."""
    g = load("fullsize.pt")
    sd = load("diffusion.pt")["deblur/net_sd"]
    net = lambda im, st: O.unet_forward(sd, im, st)
    ref_masks = masks_from_g1(g["defade128/g1d"])
    for key, c in g.items():
        if key.endswith("g1d"):
            continue
        discrete = key.endswith("/1")
        x = c["levels"].float() / 255 * 2 - 1
        # the restated mask generator, on THIS host's exp: equal to the reference's table to the last ulp or two of the taps ...
        own = O.fade_kernels("Random_Incremental", c["T"], 128, c["kernel_std"], c["initial_mask"])
        assert (own - ref_masks).abs().max() <= 2.4e-7
        masks = ref_masks                                     # ... and the chain itself is checked bit-exactly on the reference's own table
        rx, ry = c["rand_x"], c["rand_y"]
        assert torch.equal(O.fade_q_sample(x, c["t"], masks, rx, ry, discrete=discrete), c["q"]), key
        crop = lambda i: torch.stack([masks[i][rx[b]:rx[b] + 128, ry[b]:ry[b] + 128] for b in range(x.shape[0])]).unsqueeze(1)
        st = c["sample_t"]
        z = x
        for i in range(st):
            z = crop(i) * z
        z = O.quantise8(z) if discrete else z
        assert torch.equal(z, c["xt"]), key
        with torch.no_grad():
            # (cold_sample degrades first; here the start is already faded + quantised: walk back with the shared reverse update)
            img, t, direct = z, st, None
            while t:
                r = net(img, torch.full((x.shape[0],), t - 1, dtype=torch.long))
                direct = r if direct is None else direct
                img = O._reverse_update(lambda u, i: crop(i) * u, img, r, t, "x0_step_down")
                t -= 1
        assert (direct - c["direct"]).abs().max() <= 1e-6 and (img - c["img"]).abs().max() <= 2e-6, key


def _same(a, b, path=""):
    if isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), path
    elif isinstance(a, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            _same(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (u, v) in enumerate(zip(a, b)):
            _same(u, v, f"{path}[{i}]")
    else:
        assert a == b, (path, a, b)


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree only exists in the build container")
def test_make_golden_regenerates_every_committed_fixture():
    """tests/golden/*.pt ARE what the unmodified reference produces: re-run every generator of make_golden.py against
    /root/reference and require bit-equality with the committed files (every tensor, key and scalar)."""
    import contextlib
    import io
    import sys
    sys.path.insert(0, GOLD)
    try:
        import make_golden as M
    finally:
        sys.path.remove(GOLD)
    # (the committed fixtures were written with torch's default thread count of the 8-core build container; the CPU suite runs its
    # workers with OMP_NUM_THREADS=2, and a different thread count changes the summation order of the conv weight gradients)
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        _regenerate_and_compare(M)
    finally:
        torch.set_num_threads(threads)
    made = {"unet_dim8.pt", "model_ch32.pt", "model_noconv.pt", "diffusion.pt", "variants.pt", "mixing.pt", "extras.pt", "evaluation.pt", "fullsize.pt"}
    assert made == {f for f in os.listdir(GOLD) if f.endswith(".pt")}          # no fixture without a generator


def _regenerate_and_compare(M):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        ref = ref_shim.load("deblurring")
        _same(M.unet_case(ref), load("unet_dim8.pt"), "unet_dim8")
        _same(M.model_case(ref), load("model_ch32.pt"), "model_ch32")
        _same(M.model_case(ref, resamp_with_conv=False), load("model_noconv.pt"), "model_noconv")
        dc = M.diffusion_cases()
        _same(dc, load("diffusion.pt"), "diffusion")
        sd = dc["deblur/net_sd"]
        _same(M.variant_cases(sd), load("variants.pt"), "variants")
        _same(M.mixing_cases(sd), load("mixing.pt"), "mixing")
        _same(M.extra_cases(sd), load("extras.pt"), "extras")
        _same(M.evaluation_cases(sd), load("evaluation.pt"), "evaluation")
        _same(M.fullsize_cases(sd), load("fullsize.pt"), "fullsize")


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree only exists in the build container")
def test_oracle_vs_live_reference_degradations_and_samplers():
    """q_sample + sample of the four packages, live against the imported reference at a size and schedule no fixture holds
    (24 x 24, its own seeds): blur (DEBLUR:393-455, 927-960), noise (DENOISE:342-434, 517-522), pixelation (RESOL:417-459, 630-652),
    fading (DEFADE:354-425, 496-535)."""
    import contextlib
    import io
    S, B = 24, 2
    g = torch.Generator().manual_seed(4242)
    x = torch.randint(0, 256, (B, 3, S, S), generator=g).float() / 255 * 2 - 1
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())
    ref = ref_shim.load("deblurring")
    torch.manual_seed(99)
    with quiet():
        rnet = ref.Unet(dim=8, dim_mults=(1, 2), channels=3)
    sd = {k: v.clone() for k, v in rnet.state_dict().items()}
    net = lambda im, st: O.unet_forward(sd, im, st)
    with torch.no_grad(), quiet():
        # ---- blur
        for routine, ks, std in (("Incremental", 5, 0.3), ("Exponential_reflect", 7, 0.05)):
            for sampling in ("default", "x0_step_down"):
                T, t = 6, torch.tensor([5, 2])
                d = ref.GaussianDiffusion(rnet, image_size=S, device_of_kernel="cpu", channels=3, timesteps=T, kernel_std=std, kernel_size=ks,
                                          blur_routine=routine, sampling_routine=sampling)
                ws = [m.weight.detach() for m in d.gaussian_kernels]
                modes = [m.padding_mode for m in d.gaussian_kernels]
                assert torch.equal(d.q_sample(x, t), O.blur_q_sample(x, t, ws, modes, T))
                xt, direct, img = d.sample(batch_size=B, img=x)
                oxt, odirect, oimg = O.cold_sample(net, lambda z, i: O.blur_step(z, ws[i], modes[i]), x, T, sampling)
                assert torch.equal(xt, oxt) and torch.equal(direct, odirect) and (img - oimg).abs().max() <= 1e-6
        # ---- noise
        ref = ref_shim.load("denoising")
        T, t = 7, torch.tensor([6, 1])
        e = torch.randn(B, 3, S, S, generator=g)
        ca, cb = O.cosine_tables(T)
        for sampling in ("x0_step_down", "ddim"):
            d = ref.GaussianDiffusion(rnet, image_size=S, channels=3, timesteps=T, sampling_routine=sampling)
            assert torch.equal(d.q_sample(x, e, t), O.noise_q_sample(x, e, t, ca, cb))
            for got, want in ((d.gen_sample(batch_size=B, img=e), O.noise_sample(net, e, T, ca, cb, fixed_noise=sampling == "x0_step_down")),
                              (d.sample(batch_size=B, img=e), O.noise_sample(net, e, T, ca, cb, fixed_noise=False))):
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and (got[2] - want[2]).abs().max() <= 2e-6
        # ---- pixelation
        ref = ref_shim.load("resolution")
        for routine, mode in (("Incremental", "bicubic"), ("Incremental_area_factor_2", "area")):
            T = 3
            t = torch.tensor([2, 0])
            d = ref.GaussianDiffusion(rnet, image_size=S, device_of_kernel="cpu", channels=3, timesteps=T, resolution_routine=routine,
                                      sampling_routine="x0_step_down")
            sizes = O.pixelate_sizes(routine, T, S)
            assert torch.equal(d.q_sample(x, t), O.pixelate_q_sample(x, t, sizes, mode))
            xt, direct, img = d.sample(batch_size=B, img=x)
            oxt, odirect, oimg = O.cold_sample(net, lambda z, i: O.pixelate_step(z, sizes[i], mode), x, T, "x0_step_down")
            assert torch.equal(xt, oxt) and torch.equal(direct, odirect) and (img - oimg).abs().max() <= 1e-6
        # ---- fading
        ref = ref_shim.load("defading")
        for routine in ("Incremental", "Constant"):
            for sampling in ("default", "x0_step_down"):
                T, t = 5, torch.tensor([4, 1])
                d = ref.GaussianDiffusion(rnet, image_size=S, device_of_kernel="cpu", channels=3, timesteps=T, kernel_std=0.4, initial_mask=2,
                                          fade_routine=routine, sampling_routine=sampling)
                masks = O.fade_kernels(routine, T, S, 0.4, 2)
                assert torch.equal(d.fade_kernels, masks)
                assert torch.equal(d.q_sample(x, t), O.fade_q_sample(x, t, masks))
                xt, direct, img = d.sample(batch_size=B, faded_recon_sample=x)
                oxt, odirect, oimg = O.cold_sample(net, lambda z, i: masks[i] * z, x, T, sampling)
                assert torch.equal(xt, oxt) and torch.equal(direct, odirect) and (img - oimg).abs().max() <= 1e-6


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
