"""The drop-in boundary, checked against the LIVE reference (build container only: /root/reference does not exist on the GPU box).

The reference has no FFI; its boundary for the hot path is the Python class API of its packages.  Here every exported class
of every package and every public method the reference defines on it is compared by `inspect.signature` with the class of the same name in
this repository's package of the same name: the reference's parameters are a PREFIX of ours with the same names, order, kinds and
defaults (a drop-in must accept every call the reference's driver scripts make, by position or keyword, and mean the same); what this
engine adds -- e.g. Trainer(num_workers=, device_data=) -- comes after, keyword-capable and defaulted, and a parameter the reference
requires may carry a default here.  The driver call it must accept is spelled out too
(deblurring-diffusion-pytorch/mnist_train.py:64-104: Unet(...).cuda(), GaussianDiffusion(...), DataParallel wrap, Trainer(..., fp16=,
load_path=, dataset='mnist')).
"""
import importlib
import inspect
import os
import sys

import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="the reference tree only exists in the build container")

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cold-diffusion-models_amd")

# reference package (ref_shim name) -> (import path of the drop-in package, exported classes)
PACKAGES = {
    "deblurring": ("deblurring_diffusion_pytorch", ("Unet", "Model", "GaussianDiffusion", "Trainer")),
    "denoising": ("denoising_diffusion_pytorch", ("Unet", "GaussianDiffusion", "Trainer")),                  # (the reference exports no Model there)
    "resolution": ("resolution_diffusion_pytorch", ("Unet", "Model", "GaussianDiffusion", "Trainer")),
    "defading": ("defading_diffusion_pytorch", ("Unet", "Model", "GaussianDiffusion", "Trainer")),
    "demixing": ("demixing_diffusion_pytorch", ("Unet", "GaussianDiffusion", "Trainer")),
    "defading_generation": ("defading_generation.defading_diffusion_pytorch", ("Unet", "GaussianDiffusion", "Trainer")),
}
# of `Trainer`, the constructor and train / save / load / step_ema / reset_parameters are the contract, "test / figure
# methods optional".  Those REQUIRED names must exist with a compatible signature; every other Trainer method the reference defines is
# compared when this repository has it too (a wrong signature is a failure) and LISTED when it does not (printed, not a failure).
# Unet / Model / GaussianDiffusion: every public method the reference defines is required.
TRAINER_REQUIRED = ("__init__", "train", "save", "load", "step_ema", "reset_parameters")
# ... and since round 6 the only Trainer methods of the reference without a counterpart are figure code (cv2 titles, the paper_* montages:
# DESIGN.md section 7, out of scope by contract) -- anything else missing is a failure
FIGURE_CODE = {"add_title", "paper_invert_section_images", "paper_showing_diffusion_images", "paper_showing_diffusion_images_cover_page",
               "paper_showing_diffusion_images_diff", "paper_showing_sampling_diff_images", "paper_showing_diffusion_images_cover_page_both_sampling"}


def _mine(name):
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    return importlib.import_module(name)


def _params(fn):
    return [(p.name, p.kind, p.default) for p in inspect.signature(fn).parameters.values()]


def _accepts_every_reference_call(rfn, mfn):
    r, m = _params(rfn), _params(mfn)
    if len(m) < len(r):
        return False
    for (rn, rk, rd), (mn, mk, md) in zip(r, m):
        if rn != mn or rk != mk:
            return False
        if rd is not inspect.Parameter.empty and rd != md:      # same default wherever the reference has one
            return False
    return all(md is not inspect.Parameter.empty or mk in (inspect.Parameter.VAR_POSITIONAL, inspect.Parameter.VAR_KEYWORD) for _, mk, md in m[len(r):])


def _forwards_keywords(rfn, mfn):
    """A constructor written as (self, <the reference's positional parameters>, *, <some of its keywords>, **kw) that hands **kw to a base
    class of this repository: it accepts the reference's calls iff its named parameters agree with the reference's of the same name."""
    r = {n: (k, d) for n, k, d in _params(rfn)}
    m = _params(mfn)
    if not any(k == inspect.Parameter.VAR_KEYWORD for _, k, _ in m):
        return False
    pos_r = [n for n, k, _ in _params(rfn) if k == inspect.Parameter.POSITIONAL_OR_KEYWORD]
    pos_m = [n for n, k, _ in m if k == inspect.Parameter.POSITIONAL_OR_KEYWORD]
    if pos_m != pos_r:
        return False
    return all(n not in r or r[n] == (k, d) for n, k, d in m if k == inspect.Parameter.KEYWORD_ONLY)


def _own_methods(cls):
    """public callables the class itself (not nn.Module / object) defines, plus __init__ and forward"""
    out = {}
    for klass in cls.__mro__:
        if klass.__module__.startswith("torch") or klass is object:
            continue
        for name, v in vars(klass).items():
            if callable(v) and (not name.startswith("_") or name == "__init__") and name not in out:
                out[name] = v
    return out


@pytest.mark.parametrize("which", sorted(PACKAGES))
def test_signatures_equal_the_live_reference(which):
    ref = ref_shim.load(which)
    mine = _mine(PACKAGES[which][0])
    missing, different, optional = [], [], []
    for cname in PACKAGES[which][1]:
        rc, mc = getattr(ref, cname), getattr(mine, cname)
        mm = _own_methods(mc)
        for mname, rfn in _own_methods(rc).items():
            if mname not in mm and not hasattr(mc, mname):
                (optional if (cname == "Trainer" and mname not in TRAINER_REQUIRED) else missing).append(cname + "." + mname)
                continue
            mfn = mm.get(mname, getattr(mc, mname))
            if not (_accepts_every_reference_call(rfn, mfn) or (mname == "__init__" and _forwards_keywords(rfn, mfn))):
                different.append((cname + "." + mname, str(inspect.signature(rfn)), str(inspect.signature(mfn))))
    print(which, "- figure methods of the reference Trainer that are not built:", optional)
    assert {o.split(".")[1] for o in optional} <= FIGURE_CODE, "Trainer methods without a counterpart that are not figure code: %s" % optional
    assert not missing, "reference methods without a counterpart: %s" % missing
    assert not different, "signatures differ from the reference:\n" + "\n".join("%s\n   ref  %s\n   here %s" % d for d in different)


def test_reference_driver_call_is_accepted(tmp_path):
    """mnist_train.py:64-104 verbatim in its keywords (the tensors live wherever .cuda() puts them: no device in the CPU container, so the
    constructor calls are bound -- `inspect.signature(...).bind` -- not executed; execution on hardware is tests/test_modules.py's Trainer tests)."""
    mine = _mine("deblurring_diffusion_pytorch")
    inspect.signature(mine.Unet.__init__).bind(None, dim=64, dim_mults=(1, 2, 4, 8), channels=1)
    inspect.signature(mine.GaussianDiffusion.__init__).bind(None, object(), image_size=32, device_of_kernel='cuda', channels=1, timesteps=20,
                                                            loss_type='l1', kernel_std=7.0, kernel_size=11, blur_routine='Constant',
                                                            train_routine='Final', sampling_routine='x0_step_down', discrete=False)
    inspect.signature(mine.Trainer.__init__).bind(None, object(), './root_mnist/', image_size=32, train_batch_size=32, train_lr=2e-5,
                                                  train_num_steps=700000, gradient_accumulate_every=2, ema_decay=0.995, fp16=False,
                                                  results_folder=str(tmp_path), load_path=None, dataset='mnist')
    # cifar10_train.py:71-96 (BASELINE config 2)
    inspect.signature(mine.Model.__init__).bind(None, resolution=32, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2,
                                                attn_resolutions=(16,), dropout=0.1)
    for name in ("train", "save", "load", "test_from_data"):
        assert callable(getattr(mine.Trainer, name))
