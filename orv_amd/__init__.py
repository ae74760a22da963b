"""orv_amd - MI355X (gfx950) native implementation of ORV's diffusion denoising hot path.

Drop-in for ``orv.models.cogvideox_control`` / ``orv.models.components`` on that path only:

    from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj, CogVideoXImageToVideoPipelineTraj
    from orv_amd.schedulers import CogVideoXDPMScheduler, CogVideoXDDIMScheduler

or, with no edit to ORV at all, ``import orv_amd; orv_amd.install()`` ahead of the reference's own import lines.

All arithmetic runs in hand-written HIP kernels (orv_amd/csrc -> liborv_mi355.so, C ABI in include/orv_mi355.h).
There is no CPU or eager-PyTorch fallback; importing works anywhere, running needs an MI355X and the built library.
"""
__version__ = "0.1.0"


def install(verbose: bool = False):
    """Alias ORV's import paths (``orv.models.cogvideox_control``, ``orv.models.components``, the two
    ``diffusers.schedulers.scheduling_*_cogvideox`` modules, ``orv.utils.prepare_rotary_positional_embeddings``) to this package
    in ``sys.modules`` so the reference's entry points run unchanged - see ``orv_amd/dropin.py``."""
    from .dropin import install as _install
    return _install(verbose=verbose)
