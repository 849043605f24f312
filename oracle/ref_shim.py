"""TEST INFRASTRUCTURE — import the *unmodified* reference packages from /root/reference.

Only usable inside the build container (the GPU box has no /root/reference).  It is used to
 (1) validate the CPU restatement in oracle/cold_oracle.py against the reference itself, and
 (2) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py).
Nothing in the product imports this module.

The reference modules import comet_ml / torchvision / torchgeometry / cv2 / imageio /
pytorch_msssim at module top; none of them is installed here, and only
`torchgeometry.image.get_gaussian_kernel2d` is used on the path.  They are replaced by stub modules
in sys.modules; `get_gaussian_kernel2d` is restated from torchgeometry 0.1.2 (image/gaussian.py):
    gauss(x) = exp(-(x - ksize//2)^2 / (2 sigma^2)) per tap (python float -> fp32 tensor), / sum;
    kernel2d = kx[:, None] @ ky[None, :]
The reference pins no torchgeometry version => this boundary is "parity unpinned" (DESIGN.md §oracle).
"""
import importlib
import math
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"

PACKAGES = {
    "deblurring": ("deblurring-diffusion-pytorch", "deblurring_diffusion_pytorch"),
    "denoising": ("denoising-diffusion-pytorch", "denoising_diffusion_pytorch"),
    "resolution": ("resolution-diffusion-pytorch", "resolution_diffusion_pytorch"),
    "defading": ("defading-diffusion-pytorch", "defading_diffusion_pytorch"),
    # the two forward(x1, x2) packages (the second shares its package NAME with "defading")
    "demixing": ("demixing-diffusion-pytorch", "demixing_diffusion_pytorch"),
    "defading_generation": ("defading-generation-diffusion-pytorch", "defading_diffusion_pytorch"),
}


def available():
    return os.path.isdir(REFERENCE_ROOT)


def gaussian_1d(ksize, sigma):
    # torchgeometry 0.1.2 image/gaussian.py:9-17: fp32 exponent, fp32 torch.exp, per tap
    g = torch.stack([torch.exp(torch.tensor(-(x - ksize // 2) ** 2 / float(2 * sigma ** 2))) for x in range(ksize)])
    return g / g.sum()


def get_gaussian_kernel2d(ksize, sigma):
    kx = gaussian_1d(ksize[0], sigma[0])
    ky = gaussian_1d(ksize[1], sigma[1])
    return torch.matmul(kx.unsqueeze(-1), ky.unsqueeze(-1).t())


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, k):
        return _Anything()


def install_stubs():
    if "torchgeometry" in sys.modules and getattr(sys.modules["torchgeometry"], "_cdf_stub", False):
        return
    _stub("comet_ml", Experiment=_Anything)
    tr = _stub("torchvision.transforms", **{n: _Anything for n in
                                            ("Compose", "Resize", "RandomCrop", "CenterCrop", "RandomHorizontalFlip", "ToTensor", "Lambda")})
    ut = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    ds = _stub("torchvision.datasets", LSUN=_Anything)
    _stub("torchvision", transforms=tr, utils=ut, datasets=ds)
    img = _stub("torchgeometry.image", get_gaussian_kernel2d=get_gaussian_kernel2d)
    _stub("torchgeometry", image=img, _cdf_stub=True)
    _stub("cv2")
    _stub("imageio")
    _stub("pytorch_msssim", ssim=lambda *a, **k: None)
    _stub("pycave")
    _stub("pycave.bayes", GMM=_Anything)
    # the reference hard-codes .cuda(); on a CPU-only box it is the identity
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self


def load(which):
    """Import one reference package ('deblurring' | 'denoising' | 'resolution' | 'defading' | 'demixing' | 'defading_generation')."""
    assert available(), "reference tree not present (this only works in the build container)"
    install_stubs()
    folder, pkg = PACKAGES[which]
    path = os.path.join(REFERENCE_ROOT, folder)
    # two packages share module names (Model2, defading_diffusion_pytorch): import fresh each time -- and leave sys.modules as it was
    # found: the drop-in packages of this repository carry the SAME names, and a test that runs after this call must get those, not
    # the reference (which would make its parity check compare the reference with itself)
    mine = lambda name: name == pkg or name.startswith(pkg + ".")
    saved = {name: m for name, m in sys.modules.items() if mine(name)}
    for name in saved:
        del sys.modules[name]
    sys.path.insert(0, path)
    try:
        mod = importlib.import_module(pkg)
    finally:
        sys.path.remove(path)
        ref_modules = {n: m for n, m in sys.modules.items() if mine(n)}
        for name in ref_modules:
            del sys.modules[name]
        sys.modules.update(saved)
    mod._cdf_ref_modules = ref_modules          # (the reference's own submodules by dotted name, e.g. for patching module globals)
    return mod


def submodule(mod, dotted):
    """The reference submodule `dotted` (e.g. Trainer.__module__) of a package returned by load()."""
    return mod._cdf_ref_modules[dotted]
