"""Host-side mirror of the reference's cold-diffusion classes over the HIP kernel library."""
from . import parallel as _parallel

# one process per GPU: under `python -m torch.distributed.run` an unmodified reference script calls a bare `.cuda()`
# (e.g. deblurring-diffusion-pytorch/celebA_128.py:100-102); bind it to the rank's own device before anything allocates.
_parallel.pin_device()
