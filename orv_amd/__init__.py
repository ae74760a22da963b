"""orv_amd - MI355X (gfx950) native implementation of ORV's diffusion denoising hot path.

Drop-in for ``orv.models.cogvideox_control`` / ``orv.models.components`` on that path only:

    from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj, CogVideoXImageToVideoPipelineTraj
    from orv_amd.schedulers import CogVideoXDPMScheduler, CogVideoXDDIMScheduler

All arithmetic runs in hand-written HIP kernels (orv_amd/csrc -> liborv_mi355.so, C ABI in include/orv_mi355.h).
There is no CPU or eager-PyTorch fallback; importing works anywhere, running needs an MI355X and the built library.
"""
__version__ = "0.1.0"
