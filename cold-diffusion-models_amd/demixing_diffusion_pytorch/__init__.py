"""Drop-in for the reference package `demixing_diffusion_pytorch`
(demixing-diffusion-pytorch/demixing_diffusion_pytorch/__init__.py) on the MI355X engine: the cosine schedule mixes an
image of one dataset into an image of another; `Trainer(diffusion_model, folder1, folder2, ...)`."""
from colddiff.diffusion import DemixDiffusion as GaussianDiffusion
from colddiff.unet import Unet
from colddiff.model2 import Model
from colddiff.trainer import DemixTrainer as Trainer

__all__ = ["GaussianDiffusion", "Unet", "Trainer", "Model"]
