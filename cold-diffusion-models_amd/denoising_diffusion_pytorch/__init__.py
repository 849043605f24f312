"""Drop-in for the reference package `denoising_diffusion_pytorch`
(denoising-diffusion-pytorch/denoising_diffusion_pytorch/__init__.py) on the MI355X engine."""
from colddiff.diffusion import DenoiseDiffusion as GaussianDiffusion
from colddiff.unet import Unet
from colddiff.model2 import Model
from colddiff.trainer import DenoiseTrainer as Trainer

__all__ = ["GaussianDiffusion", "Unet", "Trainer", "Model"]
