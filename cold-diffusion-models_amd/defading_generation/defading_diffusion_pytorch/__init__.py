"""Drop-in for the `defading_diffusion_pytorch` package of defading-generation-diffusion-pytorch/ (the reference ships TWO
packages of that name; the inpainting one lives one directory up).  Put `cold-diffusion-models_amd/defading_generation` on
sys.path to get this one, exactly as the reference's celebA_128.py relies on its own directory:
per-pixel Gaussian-mask blend of an image into a solid-colour image, `forward(x1, x2)`."""
from colddiff.diffusion import DefadeGenDiffusion as GaussianDiffusion
from colddiff.unet import Unet
from colddiff.model2 import Model
from colddiff.trainer import DefadeGenTrainer as Trainer

__all__ = ["GaussianDiffusion", "Unet", "Trainer", "Model"]
