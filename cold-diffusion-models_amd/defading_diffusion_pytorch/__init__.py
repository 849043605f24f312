"""Drop-in for the reference package `defading_diffusion_pytorch`
(defading-diffusion-pytorch/defading_diffusion_pytorch/__init__.py:1-2) on the MI355X engine."""
from colddiff.diffusion import DefadeDiffusion as GaussianDiffusion
from colddiff.unet import Unet
from colddiff.model2 import Model
from colddiff.trainer import DefadeTrainer as Trainer

__all__ = ["GaussianDiffusion", "Unet", "Trainer", "Model"]
