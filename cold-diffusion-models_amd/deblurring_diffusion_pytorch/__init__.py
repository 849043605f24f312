"""Drop-in for the reference package `deblurring_diffusion_pytorch`
(deblurring-diffusion-pytorch/deblurring_diffusion_pytorch/__init__.py:1-2) on the MI355X engine."""
from colddiff.diffusion import DeblurDiffusion as GaussianDiffusion
from colddiff.unet import Unet
from colddiff.model2 import Model
from colddiff.trainer import Trainer

__all__ = ["GaussianDiffusion", "Unet", "Trainer", "Model"]
