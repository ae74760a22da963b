"""Cached-latent data format on either side of the hot path: what the reference's dataset hands the SFT
loop when ``load_tensors`` is on, and how a list of such samples becomes the batch ``orv_amd.sft.prepare_batch`` consumes.

Mirrors (behaviour, not code) /root/reference/orv/dataset/dataset.py:
* ``load_latent_clip``       :655-694   ``_get_frames`` (latent branch): ``torch.load`` of the VAE moments of a clip and of its
                                         reference frame(s); 3-D-VAE files are ``[2C, F, h, w]`` -> ``[F, 2C, h, w]``
* ``load_latent_controls``   :785-847   ``_get_cond_frames`` (latent branch): per-view depth / semantic-label moments, views
                                         stacked along the frame axis
* ``CollateFunctionControl`` :2053-2142 list of samples -> ``{prompt_embeds, latents, images, controls{actions,
                                         latents_depth, latents_label}, num_views, num_frames, image_width/height}``
                                         with the ``[B, F, C, h, w] -> [B, C, F, h, w]`` permute the loop expects (:864-898)

Raw-video / PIL branches (decord, torchvision transforms) belong to data preparation and are out of scope.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Sequence

import torch

VAE_SCALE_FACTOR_SPATIAL = 8
_LATENT_KEYS = (("latents", "latents"), ("image", "images"))
_CONTROL_KEYS = ("latents_depth", "latents_label", "depths", "labels")


def _load(path: str) -> torch.Tensor:
    with open(path, "rb") as f:
        return torch.load(f, weights_only=True)


def load_latent_clip(data_root: str, latent_video_path: str, latent_ref_path: str, frame_ids: Sequence[int],
                     use_3dvae: bool = True, is_sliced: bool = True) -> Dict[str, torch.Tensor]:
    """Moments of one clip (``latents`` [n_frame, 2C, h, w]) and of its reference frames (``image``)."""
    video = _load(os.path.join(data_root, latent_video_path))
    ref = _load(os.path.join(data_root, latent_ref_path))
    ids = list(frame_ids)
    if use_3dvae:                       # stored [2C, F, h, w]; RGB frame ids map to latent frames 4:1
        video, ref = video.permute(1, 0, 2, 3), ref.permute(1, 0, 2, 3)
        ids = sorted({i // 4 for i in ids})
    if is_sliced:                       # the file already holds exactly this clip
        ids = list(range(video.size(0)))
    if video.shape[0] <= max(ids):
        raise RuntimeError(f'Got mismatched latent video and frame ids: {tuple(video.shape)} v.s. {ids}, path: {latent_video_path}.')
    return {"latents": video[ids], "image": ref}


def load_latent_controls(data_root: str, control_keys: Sequence[str], latent_depth_paths=None, latent_label_paths=None,
                         use_3dvae: bool = True) -> Dict[str, torch.Tensor]:
    """Occupancy-derived depth / semantic-label moments, one file per view, views concatenated along frames."""
    out = {}
    for key, paths, name in (("depth", latent_depth_paths, "latents_depth"), ("label", latent_label_paths, "latents_label")):
        if key not in control_keys:
            continue
        if paths is None:
            raise AssertionError(f"Invalid latent_{key}_paths={paths}.")
        if not isinstance(paths, (list, tuple)):
            paths = [paths]
        views = []
        for p in paths:
            t = _load(os.path.join(data_root, p))
            views.append(t.permute(1, 0, 2, 3) if use_3dvae else t)
        out[name] = torch.stack(views).flatten(0, 1)        # [n_view * n_frame, 2C, h, w]
    return out


class CollateFunctionControl:
    """Batch assembly for the control-to-video loop (cached-latent keys; ``videos`` of raw frames also pass through)."""

    def __init__(self, weight_dtype: torch.dtype, load_tensors: bool) -> None:
        self.weight_dtype = weight_dtype
        self.load_tensors = load_tensors

    def _stack(self, data: List[Dict[str, Any]], key: str) -> torch.Tensor:
        return torch.stack([x[key] for x in data]).to(dtype=self.weight_dtype, non_blocking=True)

    def __call__(self, data: List[Dict[str, Any]]) -> Dict[str, Any]:
        keys = data[0].keys()
        out: Dict[str, Any] = {"controls": {}, "prompts": [x["prompt"] for x in data]}
        if "prompt_embeds" in keys:
            out["prompt_embeds"] = self._stack(data, "prompt_embeds")
        if "actions" in keys:
            out["controls"]["actions"] = self._stack(data, "actions")
        if "videos" in keys:
            out["videos"] = self._stack(data, "videos").permute(0, 2, 1, 3, 4)
        for src, dst in _LATENT_KEYS:
            if src in keys:
                out[dst] = self._stack(data, src).permute(0, 2, 1, 3, 4)            # -> [B, C, F, h, w]
        if "image" in keys:
            fh, fw = out["images"].shape[-2:]
            out["image_width"], out["image_height"] = int(fw * VAE_SCALE_FACTOR_SPATIAL), int(fh * VAE_SCALE_FACTOR_SPATIAL)
        for key in _CONTROL_KEYS:
            if key in keys:
                out["controls"][key] = self._stack(data, key).permute(0, 2, 1, 3, 4)
        out["metainfos"] = [x["metainfo"] for x in data]
        out["num_views"] = out["metainfos"][0]["num_view"]
        out["num_frames"] = out["metainfos"][0]["num_frame"]
        return out


class BucketSampler(torch.utils.data.Sampler):
    """Groups samples into same-shape batches: every ``batch_size`` consecutive yields share (reference-frame count, view
    count), so a data-parallel step gets a rectangular batch on every rank (the part of SURVEY §8f rank 3 that keeps 8 GPUs
    fed).  Mirror of /root/reference/orv/dataset/dataset.py:1972-2050, behaviour for behaviour:

    * yields ``(index, ref_num, n_view)`` TUPLES (the reference's datasets index by that tuple), drawn with the GLOBAL
      ``random`` module - seeding ``random`` reproduces the reference's order exactly (golden: tests/golden/bucket_sampler.json);
    * a bucket is flushed (and shuffled again) the moment it holds ``batch_size`` entries;
    * left-over partial buckets are emitted at the end only when ``drop_last`` is False AND ``shuffle`` is True - with
      ``shuffle=False`` the reference silently drops them (:2036-2043, the ``extend`` sits inside ``if self.shuffle``); kept,
      because the order of an epoch is part of what "same batches as the reference" means;
    * ``__len__`` is ``ceil(len(data_source) / batch_size)`` whatever ``drop_last`` says (:2012-2018).

    ``data_source`` needs ``resolutions`` (iterable of (ref_num, n_view) keys), ``get_ref_nums_for_all_samples()`` and
    ``get_n_views_for_all_samples(train=...)``."""

    def __init__(self, data_source, batch_size: int = 8, shuffle: bool = True, drop_last: bool = False, train: bool = True) -> None:
        self.data_source = data_source
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.drop_last = drop_last
        self.train = train
        self.buckets = {resolution: [] for resolution in data_source.resolutions}

    def __len__(self):
        return (len(self.data_source) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        import random
        order = []
        entries = [(i, r, v) for i, (r, v) in enumerate(zip(self.data_source.get_ref_nums_for_all_samples(),
                                                            self.data_source.get_n_views_for_all_samples(train=self.train)))]
        if self.shuffle:
            random.shuffle(entries)
        for entry in entries:
            key = (entry[1], entry[2])
            bucket = self.buckets[key]
            bucket.append(entry)
            if len(bucket) == self.batch_size:
                if self.shuffle:
                    random.shuffle(bucket)
                order.extend(bucket)
                del self.buckets[key]          # re-inserted at the END of the dict, as in the reference: the order in which
                self.buckets[key] = []         # left-over buckets are emitted below depends on it
        if not self.drop_last and self.shuffle:
            for key, bucket in list(self.buckets.items()):
                if bucket:
                    random.shuffle(bucket)
                    order.extend(bucket)
                    del self.buckets[key]
                    self.buckets[key] = []
        yield from order
