"""Import the REAL reference (``/root/reference/orv``) in the build container to generate golden vectors.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (never on the GPU box; nothing
in ``tests/`` imports this at run time - the fixtures it produced are committed under tests/golden/).

The reference's hot path subclasses ``diffusers`` classes; diffusers is absent and not installable
here.  This module installs stand-in ``diffusers.*`` / ``omegaconf`` / ``torchvision`` modules into
``sys.modules`` whose leaf classes are oracle/leaf.py's restatement, then imports
``orv.models.cogvideox_control`` / ``orv.models.components`` unmodified, so that ORV's *own*
``CogVideoXLayerNormZero.forward``, ``AdaLayerNorm.forward``, ``CogVideoXAttnProcessor2_0.__call__``,
``CogVideoXBlock.forward``, ``MVBlock.forward``, ``CogVideoXTransformer3DModelTraj.forward``,
``ActionEmbed``/``ActionRecon``, ``prepare_latents`` and the ``__call__`` denoise loop execute verbatim.
What this pins: everything ORV authored.  What it does not pin: the diffusers leaf arithmetic.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import importlib.machinery
import inspect
import sys
import types
from typing import Any, Dict

import numpy as np
import torch

REF = REFERENCE_ROOT if "REFERENCE_ROOT" in globals() else "/root/reference"
from torch import nn

from . import leaf

REFERENCE_ROOT = "/root/reference"


class _Config(dict):
    """diffusers FrozenDict stand-in: attribute + mapping access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _register_to_config(init):
    sig = inspect.signature(init)

    def wrapper(self, *args, **kwargs):
        cfg = {k: p.default for k, p in sig.parameters.items()
               if k != "self" and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL)}
        names = [k for k in sig.parameters if k != "self"]
        for n, a in zip(names, args):
            cfg[n] = a
        cfg.update(kwargs)
        object.__setattr__(self, "_orv_config", _Config(cfg))
        init(self, *args, **kwargs)

    wrapper.__wrapped__ = init
    return wrapper


class _ModelMixin:
    @property
    def config(self):
        return self.__dict__["_orv_config"]

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class _CogVideoXTransformer3DModel(leaf.CogVideoXTransformer3DModelBase):
    pass


class _DiffusionPipeline:
    pass


class _FakeVAE:
    def __init__(self, scaling_factor=1.15258426, latent_channels=16, invert_scale_latents=False):
        self.config = _Config(scaling_factor=scaling_factor, latent_channels=latent_channels,
                              invert_scale_latents=invert_scale_latents, block_out_channels=(128, 256, 256, 512),
                              temporal_compression_ratio=4)


class _CogVideoXImageToVideoPipeline(_DiffusionPipeline):
    """Only the helpers ORV's ``__call__`` touches when prompt_embeds are given and output_type='latent'."""

    def __init__(self, tokenizer, text_encoder, vae, transformer, scheduler):
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.transformer, self.scheduler = transformer, scheduler
        self.vae_scale_factor_spatial = 2 ** (len(vae.config.block_out_channels) - 1)
        self.vae_scale_factor_temporal = vae.config.temporal_compression_ratio
        self.vae_scaling_factor_image = vae.config.scaling_factor
        self._interrupt = False

    guidance_scale = property(lambda self: self._guidance_scale)
    interrupt = property(lambda self: self._interrupt)
    _execution_device = property(lambda self: torch.device("cpu"))

    def check_inputs(self, *a, **k):
        pass

    def encode_prompt(self, prompt=None, negative_prompt=None, do_classifier_free_guidance=False,
                      num_videos_per_prompt=1, prompt_embeds=None, negative_prompt_embeds=None, **_):
        if prompt_embeds is None:
            raise NotImplementedError("the golden harness always passes prompt_embeds (T5 is out of scope)")
        return prompt_embeds, negative_prompt_embeds

    def prepare_extra_step_kwargs(self, generator, eta):
        params = set(inspect.signature(self.scheduler.step).parameters)
        extra = {}
        if "eta" in params:
            extra["eta"] = eta
        if "generator" in params:
            extra["generator"] = generator
        return extra

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield types.SimpleNamespace(update=lambda *a, **k: None)

    def maybe_free_model_hooks(self):
        pass


class _VideoProcessor:
    def __init__(self, vae_latent_channels=16, vae_scale_factor=8, **_):
        self.config = _Config(do_resize=True, do_normalize=True, do_binarize=False, do_convert_rgb=False,
                              do_convert_grayscale=False, vae_latent_channels=vae_latent_channels,
                              vae_scale_factor=vae_scale_factor)


def _is_valid_image(image):
    import PIL.Image
    return isinstance(image, PIL.Image.Image) or (isinstance(image, (np.ndarray, torch.Tensor)) and image.ndim in (2, 3))


class _Logging:
    @staticmethod
    def get_logger(name):
        import logging
        return logging.getLogger(name)


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []  # behave as a package so dotted children resolve
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _mod(parent)
        setattr(sys.modules[parent], child, m)
    return m


_INSTALLED = False


def install_stubs() -> None:
    global _INSTALLED
    if _INSTALLED:
        return
    import transformers.models.t5  # noqa: F401  (before any torchvision stand-in exists)
    from dataclasses import dataclass

    @dataclass
    class Transformer2DModelOutput:
        sample: torch.Tensor

    @dataclass
    class CogVideoXPipelineOutput:
        frames: Any = None

    _mod("diffusers")
    _mod("diffusers.utils.constants", USE_PEFT_BACKEND=False)
    _mod("diffusers.utils.import_utils", is_torch_version=lambda op, v: True, logging=_Logging)
    _mod("diffusers.utils.peft_utils", scale_lora_layers=lambda *a: None, unscale_lora_layers=lambda *a: None)
    _mod("diffusers.utils.torch_utils", randn_tensor=leaf.randn_tensor)
    _mod("diffusers.configuration_utils", register_to_config=_register_to_config)
    _mod("diffusers.models.transformers.cogvideox_transformer_3d", CogVideoXBlock=leaf.CogVideoXBlock,
         CogVideoXTransformer3DModel=_CogVideoXTransformer3DModel)
    _mod("diffusers.models.embeddings", CogVideoXPatchEmbed=leaf.CogVideoXPatchEmbed,
         get_3d_sincos_pos_embed=leaf.get_3d_sincos_pos_embed, apply_rotary_emb=leaf.apply_rotary_emb,
         get_3d_rotary_pos_embed=leaf.get_3d_rotary_pos_embed)
    _mod("diffusers.models.modeling_utils", ModelMixin=_ModelMixin)
    _mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=Transformer2DModelOutput)
    _mod("diffusers.models.autoencoders.vae", DiagonalGaussianDistribution=leaf.DiagonalGaussianDistribution)
    _mod("diffusers.models.attention_processor", Attention=leaf.Attention,
         CogVideoXAttnProcessor2_0=leaf.CogVideoXAttnProcessor2_0)
    _mod("diffusers.models.autoencoders.autoencoder_kl_cogvideox", AutoencoderKLCogVideoX=_FakeVAE)
    _mod("diffusers.models.normalization", AdaLayerNorm=leaf.AdaLayerNorm,
         CogVideoXLayerNormZero=leaf.CogVideoXLayerNormZero)
    _mod("diffusers.pipelines.cogvideo.pipeline_output", CogVideoXPipelineOutput=CogVideoXPipelineOutput)
    _mod("diffusers.pipelines.cogvideo.pipeline_cogvideox_image2video",
         CogVideoXImageToVideoPipeline=_CogVideoXImageToVideoPipeline, retrieve_latents=None)
    _mod("diffusers.pipelines.cogvideo.pipeline_cogvideox", retrieve_timesteps=leaf.retrieve_timesteps)
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=_DiffusionPipeline)
    _mod("diffusers.schedulers.scheduling_ddim_cogvideox", CogVideoXDDIMScheduler=leaf.CogVideoXDDIMScheduler)
    _mod("diffusers.schedulers.scheduling_dpm_cogvideox", CogVideoXDPMScheduler=leaf.CogVideoXDPMScheduler)
    _mod("diffusers.video_processor", VideoProcessor=_VideoProcessor)
    _mod("diffusers.image_processor", PipelineImageInput=Any, is_valid_image=_is_valid_image)
    _mod("omegaconf.dictconfig", DictConfig=dict)
    _mod("torchvision.transforms.functional")
    _INSTALLED = True


def load_reference():
    """Returns (cogvideox_control module, components module, utils module) of the real reference."""
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    cc = importlib.import_module("orv.models.cogvideox_control")
    comp = importlib.import_module("orv.models.components")
    utils = importlib.import_module("orv.utils")
    return cc, comp, utils


@contextlib.contextmanager
def forced_action_mask(values):
    """Make ActionEmbed's ``torch.rand(B) < 0.1`` (components.py:67) return ``values`` (bool list)."""
    real = torch.rand

    def fake(*size, **kw):
        n = size[0] if not isinstance(size[0], (tuple, list)) else size[0][0]
        v = torch.tensor([0.0 if m else 1.0 for m in values], dtype=torch.float32)
        assert v.numel() == n, (v.numel(), n)
        return v

    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = real


def make_pipeline(cc, transformer, scheduler, scaling_factor=1.15258426):
    return cc.CogVideoXImageToVideoPipelineTraj(tokenizer=None, text_encoder=None,
                                                vae=_FakeVAE(scaling_factor=scaling_factor),
                                                transformer=transformer, scheduler=scheduler)


def load_reference_class(rel_path: str, class_name: str, extra_globals=None):
    """Exec ONE pure-torch class of the reference in isolation (its module imports decord / torchvision / PIL, absent
    here): the class definition is AST-sliced out of the file and compiled with the few names it needs (SURVEY §8c,
    "AST-extract pure-torch classes").  Build-container use only (golden-vector generation)."""
    import ast
    import itertools
    import typing
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == class_name)
    mod = ast.Module(body=[node], type_ignores=[])
    glb = {"torch": torch, "Any": typing.Any, "Optional": typing.Optional, "chain": itertools.chain, "Tensor": torch.Tensor}
    glb.update(extra_globals or {})
    exec(compile(mod, os.path.join(REF, rel_path), "exec"), glb)
    return glb[class_name]
