"""Multi-GPU inference = independent replicas over clips (/root/reference/orv/pipeline/evaluation_control_to_video.py:
212-222: each rank slices `dataset.samples`); the only cross-rank traffic is rendezvous, the wall-clock max and the final
result merge - no collective on the data path."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_clips(clips: Sequence, rank: int, world_size: int) -> List:
    """Strided slice `clips[rank::world_size]`, the reference's partitioning."""
    return list(clips[rank::world_size])


def merge_rank_results(results: Dict, wall: torch.Tensor) -> Tuple[Dict, float]:
    """Gather per-rank {clip_id: result} dicts on every rank and the max wall time (what bench.py reports)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(results), float(wall.item())
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, results)
    merged: Dict = {}
    for part in gathered:
        overlap = set(merged) & set(part)
        if overlap:
            raise RuntimeError(f"clips processed by more than one rank: {sorted(overlap)}")
        merged.update(part)
    w = wall.clone()
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return merged, float(w.item())
