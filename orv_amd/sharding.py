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


# ---- training: data parallel gradient all-reduce (the ONE collective of the path) ----------------------------------------
def allreduce_flat_(flat: torch.Tensor, chunk_elems: int = 128 * 1024 * 1024, group=None, average: bool = True) -> int:
    """In-place sum / average of ONE flat gradient buffer (``FusedAdamW``'s) over the data-parallel group, ``chunk_elems``
    elements per collective (256 MB of bf16: ring all-reduce over xGMI is per-link bound, so few large messages).
    A parameter without a gradient on some rank holds zeros in its segment.  Returns the number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    n = 0
    for s0 in range(0, flat.numel(), chunk_elems):
        piece = flat[s0:s0 + chunk_elems]
        dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group)
        n += 1
    if average:
        flat.mul_(1.0 / world)
    return n


class FlatGradReducer:
    """Overlaps the data-parallel gradient exchange with the backward: segments of ONE flat gradient buffer are summed over
    the group as soon as a contiguous run of them is final (``ready``), on the collective stream, while the hand-written
    backward keeps producing earlier layers; ``finish`` sends what is left, waits, and averages.

    The reference gets this from DDP's bucket hooks (accelerate, `train_cogvideox_control_to_video_sft.py:750,1093`).  Here
    runs are coalesced up to ``max_elems`` (256 MB of bf16) and launched once they reach ``min_elems`` (16 MB): xGMI rings are
    per-link bound, so messages stay large.  Round 6: the optimizer lays the six big weights of every block out first and contiguously (the model
    tags them: they are the gradients that are final when the backward leaves the block), so a block's early gradients are ONE run of 88 MB (2B)
    that leaves during the backward; biases / LayerNorm affines / AdaLN linears / embeddings (final at the end) follow as one tail that
    ``finish`` sends in 256-MB pieces: 30 + ~4 collectives per 2B step.  Rounds 3-5 interleaved both kinds in model order: 121 collectives, four
    per block (two early FeedForward weights, two late runs), the attention weights exposed in ``finish`` (`profiles/r5_multi_rank_check.txt`)
    - against DDP's ~140 buckets of 25 MB.
    """

    def __init__(self, flat: torch.Tensor, seg_start, min_elems: int = 8 << 20, max_elems: int = 128 << 20, group=None):
        self.flat, self.group = flat, group
        self.start = [int(x) for x in seg_start]            # nseg + 1 element offsets (segments padded, contiguous)
        n = len(self.start) - 1
        self.ready_flag, self.sent = [False] * n, [False] * n
        self.min_elems, self.max_elems = min_elems, max_elems
        self.handles, self.n_collectives = [], 0
        self.bytes_sent = 0                                   # payload bytes handed to all-reduce (per rank)
        self.exposed = None                                   # (event, event) on the compute stream around finish()'s waits
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def _send(self, a: int, b: int):
        """all-reduce segments [a, b) in pieces of at most max_elems."""
        lo, hi = self.start[a], self.start[b]
        for s0 in range(lo, hi, self.max_elems):
            piece = self.flat[s0:min(hi, s0 + self.max_elems)]
            if self.world > 1:
                self.handles.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.n_collectives += 1
            self.bytes_sent += piece.numel() * piece.element_size()
        for i in range(a, b):
            self.sent[i] = True

    def _scan(self, force: bool):
        n = len(self.sent)
        i = 0
        while i < n:
            if self.sent[i] or not (self.ready_flag[i] or force):
                i += 1
                continue
            j = i
            while j < n and not self.sent[j] and (self.ready_flag[j] or force):
                j += 1
            if force or self.start[j] - self.start[i] >= self.min_elems:
                self._send(i, j)
            i = j

    def ready(self, segments):
        """Mark segments (indices) final; launches every contiguous final run that has reached ``min_elems``."""
        for k in segments:
            self.ready_flag[k] = True
        self._scan(force=False)

    def finish(self, average: bool = True) -> int:
        """Send everything not sent yet (segments never marked count as final: zeros or already-copied gradients), wait for
        all collectives, average.  Returns the number of collectives issued."""
        self._scan(force=True)
        timed = self.flat.is_cuda and self.world > 1
        if timed:                                             # what the compute stream waits for here is the NON-overlapped part of the exchange
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self.handles:
            h.wait()
        if timed:
            e1.record()
            self.exposed = (e0, e1)
        self.handles = []
        if average and self.world > 1:
            self.flat.mul_(1.0 / self.world)
        return self.n_collectives

    def stats(self) -> Dict:
        """Exchange bookkeeping of the finished step: collectives, payload bytes per rank, exposed (non-overlapped) milliseconds on the
        compute stream (None without a GPU group; synchronises on the second event)."""
        ms = None
        if self.exposed is not None:
            self.exposed[1].synchronize()
            ms = self.exposed[0].elapsed_time(self.exposed[1])
        return {"collectives": self.n_collectives, "bytes": self.bytes_sent, "exposed_ms": ms}
