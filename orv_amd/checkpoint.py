"""safetensors (de)serialisation of a transformer directory: ``diffusion_pytorch_model*.safetensors`` (+ index json),
the layout written by the reference's ``save_pretrained(safe_serialization=True, max_shard_size='5GB')``
(/root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:396-409) and read back by ``from_pretrained``."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Union

import torch

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
INDEX_NAME = "diffusion_pytorch_model.safetensors.index.json"


def _parse_size(s: Union[int, str]) -> int:
    if isinstance(s, int):
        return s
    s = s.upper().strip()
    for suf, mul in (("GB", 10 ** 9), ("MB", 10 ** 6), ("KB", 10 ** 3)):
        if s.endswith(suf):
            return int(float(s[: -len(suf)]) * mul)
    return int(s)


def load_state_dict_dir(directory: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    index = os.path.join(directory, INDEX_NAME)
    if os.path.exists(index):
        with open(index, "r", encoding="utf-8") as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = [os.path.basename(p) for p in sorted(glob.glob(os.path.join(directory, "diffusion_pytorch_model*.safetensors")))]
    if not files:
        raise RuntimeError(f"no diffusion_pytorch_model*.safetensors under {directory}")
    state: Dict[str, torch.Tensor] = {}
    for fn in files:
        state.update(load_file(os.path.join(directory, fn)))
    return state


def save_state_dict_dir(state: Dict[str, torch.Tensor], directory: str, max_shard_size: Union[int, str] = "5GB") -> None:
    from safetensors.torch import save_file
    limit = _parse_size(max_shard_size)
    shards, cur, cur_bytes = [], {}, 0
    for k, v in state.items():
        nb = v.numel() * v.element_size()
        if cur and cur_bytes + nb > limit:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = v.detach().cpu().contiguous()
        cur_bytes += nb
    shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(directory, WEIGHTS_NAME), metadata={"format": "pt"})
        return
    weight_map, total = {}, 0
    for i, sh in enumerate(shards):
        fn = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(directory, fn), metadata={"format": "pt"})
        for k, v in sh.items():
            weight_map[k] = fn
            total += v.numel() * v.element_size()
    with open(os.path.join(directory, INDEX_NAME), "w", encoding="utf-8") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
