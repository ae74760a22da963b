"""``orv_amd.install()`` - make ORV's own import lines resolve to this package, so that its entry points run UNCHANGED.

The reference's entry points import the hot path by module path
(/root/reference/orv/pipeline/inference_control_to_video.py:7-17, evaluation_control_to_video.py:17-23,
train_cogvideox_control_to_video_sft.py:98):

    from diffusers.schedulers.scheduling_dpm_cogvideox import CogVideoXDPMScheduler
    from orv.models.cogvideox_control import CogVideoXTransformer3DModelTraj, CogVideoXImageToVideoPipelineTraj
    from orv.utils import CONSOLE

``install()`` puts aliases into ``sys.modules`` BEFORE those lines run (one line at the top of a launcher, ``python -c "import
orv_amd; orv_amd.install(); import runpy; runpy.run_path('orv/pipeline/inference_control_to_video.py', run_name='__main__')"``,
or ``sitecustomize``): no file of ORV is edited.

  orv.models.cogvideox_control                          -> orv_amd.cogvideox_control
  orv.models.components                                 -> orv_amd.components
  orv.utils.prepare_rotary_positional_embeddings        -> orv_amd.utils.prepare_rotary_positional_embeddings (orv/utils.py:178-239)
  diffusers.schedulers.scheduling_dpm_cogvideox         -> CogVideoXDPMScheduler of orv_amd.schedulers
  diffusers.schedulers.scheduling_ddim_cogvideox        -> CogVideoXDDIMScheduler of orv_amd.schedulers

Everything else of ``orv`` / ``diffusers`` stays whatever the environment provides.  Where the environment provides NOTHING (this
build's containers have neither diffusers nor the reference on the path) the missing parents are filled with the smallest stand-ins the
import blocks need, and only then: ``orv.utils`` (``CONSOLE`` + the RoPE helper), ``orv.dataset.dataset`` (this package's
``CollateFunctionControl`` / ``BucketSampler`` / latent loaders; the raw-video dataset classes are out of scope and raise on
construction), ``diffusers.configuration_utils.FrozenDict`` and ``diffusers.utils.export_utils.export_to_video``.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types
from typing import Dict, List

_installed: Dict[str, types.ModuleType] = {}
_MISSING = object()
# what install() displaced, so that uninstall() can put it back: name -> (previous sys.modules entry or _MISSING, previous
# attribute of the parent package or _MISSING); and (object, attribute name, previous value) for attributes patched in place
_previous: Dict[str, tuple] = {}
_patched: list = []


def _have(name: str) -> bool:
    """Is ``name`` importable from the environment (not from one of our own stand-ins)?"""
    if name in sys.modules:
        return not getattr(sys.modules[name], "__orv_amd_standin__", False)
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError, AttributeError):
        return False


def _standin(name: str, doc: str) -> types.ModuleType:
    m = types.ModuleType(name, doc)
    m.__orv_amd_standin__ = True
    m.__path__ = []                        # a package: submodules may hang below it
    return m


def _ensure_parent(name: str) -> types.ModuleType:
    """The package ``name`` from the environment if it exists, else a stand-in; registered in ``sys.modules`` either way."""
    if name in sys.modules:
        return sys.modules[name]
    if _have(name):
        try:
            return importlib.import_module(name)
        except Exception:                  # present but broken here (missing optional dependency): stand in for it
            pass
    m = _standin(name, f"stand-in created by orv_amd.install(): `{name}` is not importable in this environment")
    sys.modules[name] = m
    _installed[name] = m
    if "." in name:
        parent, _, leaf = name.rpartition(".")
        setattr(_ensure_parent(parent), leaf, m)
    return m


def _alias(name: str, module: types.ModuleType) -> None:
    parent, _, leaf = name.rpartition(".")
    pkg = _ensure_parent(parent)
    if name not in _previous:              # first install only: a repeated install() must not record its own aliases as "previous"
        _previous[name] = (sys.modules.get(name, _MISSING), getattr(pkg, leaf, _MISSING))
    sys.modules[name] = module
    setattr(pkg, leaf, module)
    _installed[name] = module


def _scheduler_module(name: str, cls) -> types.ModuleType:
    m = types.ModuleType(name, f"orv_amd.install(): `{cls.__name__}` of orv_amd.schedulers under diffusers' module path")
    setattr(m, cls.__name__, cls)
    m.__all__ = [cls.__name__]
    return m


class _OutOfScope:
    """Placeholder for a reference class outside this build (raw-video datasets): importable, not constructible."""
    _name = "?"

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{self._name} belongs to the reference's raw-video data stack (decord / cv2 / torchvision), which orv_amd "
                                  "does not rebuild: put the reference's `orv` package on sys.path, orv_amd.install() leaves it in place")


def install(verbose: bool = False) -> List[str]:
    """Install the aliases (idempotent).  Returns the module names now served by ``orv_amd`` or its stand-ins."""
    from . import cogvideox_control, components, data, schedulers, utils

    _alias("orv.models.cogvideox_control", cogvideox_control)
    _alias("orv.models.components", components)
    _alias("diffusers.schedulers.scheduling_dpm_cogvideox",
           _scheduler_module("diffusers.schedulers.scheduling_dpm_cogvideox", schedulers.CogVideoXDPMScheduler))
    _alias("diffusers.schedulers.scheduling_ddim_cogvideox",
           _scheduler_module("diffusers.schedulers.scheduling_ddim_cogvideox", schedulers.CogVideoXDDIMScheduler))
    for cls in (schedulers.CogVideoXDPMScheduler, schedulers.CogVideoXDDIMScheduler):     # `from diffusers import CogVideoXDPMScheduler`
        for pkg in ("diffusers.schedulers", "diffusers"):
            if getattr(sys.modules.get(pkg), "__orv_amd_standin__", False):
                setattr(sys.modules[pkg], cls.__name__, cls)

    # orv.utils: the environment's module keeps everything but the RoPE helper; without one, CONSOLE + the helper
    ou = None
    if _have("orv.utils"):
        try:
            ou = importlib.import_module("orv.utils")
        except Exception:
            ou = None
    if ou is None:
        ou = types.ModuleType("orv.utils", "stand-in created by orv_amd.install(): CONSOLE + prepare_rotary_positional_embeddings")
        ou.__orv_amd_standin__ = True
        try:
            from rich.console import Console
            ou.CONSOLE = Console(width=120)                      # orv/utils.py:15
        except Exception:                                        # pragma: no cover
            class _Plain:
                def log(self, *a, **k):
                    print(*a)
                print = log
            ou.CONSOLE = _Plain()
        _alias("orv.utils", ou)
    if not getattr(ou, "__orv_amd_standin__", False) and not any(o is ou for o, _, _ in _patched):
        _patched.append((ou, "prepare_rotary_positional_embeddings", getattr(ou, "prepare_rotary_positional_embeddings", _MISSING)))
    ou.prepare_rotary_positional_embeddings = utils.prepare_rotary_positional_embeddings
    if "orv.utils" not in _installed:
        _installed["orv.utils"] = ou
        _previous.setdefault("orv.utils", (ou if not getattr(ou, "__orv_amd_standin__", False) else _MISSING, _MISSING))

    # orv.dataset.dataset: only when the reference's is not there
    if not _have("orv.dataset.dataset"):
        od = types.ModuleType("orv.dataset.dataset", "stand-in created by orv_amd.install(): orv_amd.data's collate / sampler / latent loaders")
        od.__orv_amd_standin__ = True
        for k in ("CollateFunctionControl", "BucketSampler", "load_latent_clip", "load_latent_controls"):
            if hasattr(data, k):
                setattr(od, k, getattr(data, k))

        def _missing(name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (_OutOfScope,), {"_name": f"orv.dataset.dataset.{name}"})
        od.__getattr__ = _missing
        _alias("orv.dataset.dataset", od)

    # diffusers leaves the import blocks name besides the schedulers - only when diffusers itself is absent
    if getattr(sys.modules.get("diffusers"), "__orv_amd_standin__", False):
        cu = types.ModuleType("diffusers.configuration_utils", "stand-in created by orv_amd.install()")
        cu.__orv_amd_standin__ = True
        cu.FrozenDict = cogvideox_control.FrozenConfig           # `FrozenDict(**cfg)` at inference_control_to_video.py:108
        _alias("diffusers.configuration_utils", cu)
        eu = types.ModuleType("diffusers.utils.export_utils", "stand-in created by orv_amd.install()")
        eu.__orv_amd_standin__ = True
        eu.export_to_video = _export_to_video
        _alias("diffusers.utils.export_utils", eu)
    if verbose:
        for k in sorted(_installed):
            print(f"[orv_amd.install] {k} -> {getattr(_installed[k], '__name__', _installed[k])}"
                  f"{' (stand-in)' if getattr(_installed[k], '__orv_amd_standin__', False) else ''}")
    return sorted(_installed)


def uninstall() -> None:
    """Undo ``install()``: aliases and stand-ins leave ``sys.modules`` / their parent packages, whatever they displaced (a real
    ``diffusers.schedulers.scheduling_*_cogvideox``, the reference's own ``orv.utils.prepare_rotary_positional_embeddings``) comes back."""
    for obj, attr, prev in _patched:
        if prev is _MISSING:
            if hasattr(obj, attr):
                delattr(obj, attr)
        else:
            setattr(obj, attr, prev)
    _patched.clear()
    for k, m in list(_installed.items()):
        prev_mod, prev_attr = _previous.get(k, (_MISSING, _MISSING))
        if sys.modules.get(k) is m:
            if prev_mod is _MISSING:
                del sys.modules[k]
            else:
                sys.modules[k] = prev_mod
        parent, _, leaf = k.rpartition(".")
        if parent in sys.modules and getattr(sys.modules[parent], leaf, None) is m:
            if prev_attr is _MISSING:
                try:
                    delattr(sys.modules[parent], leaf)
                except AttributeError:
                    pass
            else:
                setattr(sys.modules[parent], leaf, prev_attr)
    _installed.clear()
    _previous.clear()


def _export_to_video(video_frames, output_video_path: str = None, fps: int = 10, **_):
    """Minimal stand-in for diffusers' ``export_to_video`` (used at inference_control_to_video.py:149 after the hot path): writes the PIL / numpy
    frames with imageio when present, else as an animated GIF next to the requested path."""
    import numpy as np
    frames = [np.asarray(f) if not isinstance(f, np.ndarray) else f for f in video_frames]
    frames = [(f * 255).round().astype("uint8") if f.dtype.kind == "f" else f for f in frames]
    try:
        import imageio
        imageio.mimsave(output_video_path, frames, fps=fps)
        return output_video_path
    except Exception:
        from PIL import Image
        path = output_video_path.rsplit(".", 1)[0] + ".gif"
        ims = [Image.fromarray(f) for f in frames]
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=int(1000 / max(fps, 1)), loop=0)
        return path
