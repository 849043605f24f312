"""Drop-in for the reference package `resolution_diffusion_pytorch`
(resolution-diffusion-pytorch/resolution_diffusion_pytorch/__init__.py:1-2) on the MI355X engine."""
from colddiff.diffusion import ResolutionDiffusion as GaussianDiffusion
from colddiff.unet import Unet
from colddiff.model2 import Model
from colddiff.trainer import ResolutionTrainer as Trainer

__all__ = ["GaussianDiffusion", "Unet", "Trainer", "Model"]
