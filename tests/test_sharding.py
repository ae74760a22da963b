"""CPU, world_size 2 over gloo: the N>1 path of the denoise job is replica sharding - each rank takes a strided slice of
the clip list (evaluation_control_to_video.py:212-222), runs alone, and only the wall-clock max / result merge cross
ranks.  No data-path collective exists in inference."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orv_amd.sharding import merge_rank_results, shard_clips


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_clips(list(range(n_clips)), rank, world)
    # stand-in for the per-clip denoise: a deterministic function of the clip id
    results = {i: float(i) * 2.0 + 1.0 for i in mine}
    wall = torch.tensor([1.0 + rank], dtype=torch.float64)
    merged, wall_max = merge_rank_results(results, wall)
    if rank == 0:
        out.put((merged, wall_max))
    dist.destroy_process_group()


def test_two_rank_replica_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    [p.start() for p in procs]
    merged, wall_max = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert merged == {i: float(i) * 2.0 + 1.0 for i in range(7)}        # every clip exactly once
    assert wall_max == 2.0                                              # max over ranks, as bench.py reports


def test_shard_is_a_partition():
    for n, w in [(0, 2), (1, 2), (7, 2), (8, 8), (5, 8)]:
        parts = [shard_clips(list(range(n)), r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # the flat-buffer form used by FusedAdamW: one buffer, several collectives, averaged in place
    from orv_amd.sharding import allreduce_flat_
    flat = torch.arange(10000, dtype=torch.float32) * (rank + 1)
    nf = allreduce_flat_(flat, chunk_elems=4096)
    # overlap form: segments become final out of order (as the backward walks the layers), runs are coalesced
    from orv_amd.sharding import FlatGradReducer
    starts = [0, 100, 1000, 1300, 5000, 5004, 9000, 10000]
    buf = torch.arange(10000, dtype=torch.float32) * (rank + 1)
    red = FlatGradReducer(buf, starts, min_elems=900, max_elems=3000)
    red.ready([5])                 # 3996 elements: sent at once (2 pieces)
    red.ready([1])                 # exactly 900: sent
    red.ready([3])                 # 3700 elements: sent (2 pieces)
    red.ready([2])                 # 300 < min and its neighbours are already sent: waits for finish
    nr = red.finish()
    if rank == 0:
        # numpy copies, not tensors: a tensor travels as a shared-memory fd that dies with this process (flaky FileNotFoundError)
        out.put((nf, flat.numpy().copy(), nr, buf.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce():
    """training's one collective: all-reduce(avg) of the flat gradient buffer (whole, and overlapped in coalesced runs),
    world_size 2 over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    nf, flat, nr, buf = q.get(timeout=120)
    flat, buf = torch.from_numpy(flat), torch.from_numpy(buf)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert nf == 3 and torch.allclose(flat, torch.arange(10000, dtype=torch.float32) * 1.5)
    assert nr == 2 + 1 + 2 + 4 and torch.allclose(buf, torch.arange(10000, dtype=torch.float32) * 1.5)


def test_block_gradients_leave_as_one_collective_per_block():
    """Round 6: FusedAdamW lays the parameters the model tags as "final when the block's backward ends" (the six big weights of a block) out
    first and contiguously, so FlatGradReducer sends ONE run per block during the backward and one tail in finish() - not two early + two late
    pieces per block (121 collectives per 2B step in rounds 3-5).  CPU, no process group: the reducer counts what it would send."""
    import torch
    from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj
    from orv_amd.optim import FusedAdamW
    from orv_amd.sharding import FlatGradReducer
    L = 3
    m = CogVideoXTransformer3DModelTraj(num_attention_heads=2, attention_head_dim=64, in_channels=8, out_channels=4, num_layers=L,
                                        text_embed_dim=32, time_embed_dim=32, sample_width=12, sample_height=8, sample_frames=9,
                                        max_text_seq_length=8).to(torch.bfloat16)
    opt = FusedAdamW(m.parameters(), lr=1e-3)
    early = [p for p in opt.params if getattr(p, "_orv_grad_early", False)]
    assert len(early) == 6 * L and opt.params[:len(early)] == early                      # tagged first, model order kept
    assert sorted(map(id, opt.params)) == sorted(id(p) for p in m.parameters() if p.requires_grad)
    opt._build()
    f = opt._flat
    index = {id(p): i for i, p in enumerate(opt.params)}
    red = FlatGradReducer(f["g"], f["reduce_starts"], min_elems=1, max_elems=1 << 40)
    for blk in reversed(list(m.transformer_blocks)):                                      # the backward walks the blocks from the last to the first
        before = red.n_collectives
        ws = [blk.attn1.to_q.weight, blk.attn1.to_k.weight, blk.attn1.to_v.weight, blk.attn1.to_out[0].weight,
              blk.ff.net[0].proj.weight, blk.ff.net[2].weight]
        seg = sorted(index[id(w)] for w in ws)
        assert seg == list(range(seg[0], seg[0] + 6))                                     # contiguous in the flat buffer
        red.ready(seg)
        assert red.n_collectives == before + 1                                            # one run, one collective
    n = red.finish(average=False)
    assert n == L + 1                                                                     # + ONE tail: everything that is final at the end
