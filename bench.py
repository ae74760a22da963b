#!/usr/bin/env python
"""Headline benchmark of the ORV denoising hot path on MI355X (contract: see the task brief / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" is ONE denoise step of BASELINE.json configs[1] - CogVideoX-2B (ORV "1.7b": D=1920, 30 heads x 64, 30 layers,
FFN 7680) on 320x480x17-frame clips = latents [B,5,32,40,60] (16 noisy + 16 image-condition channels), 226 text tokens +
3000 video tokens, per-frame action (trajectory) modulation, guidance 1.0, bf16: channel-concat -> 3-D DiT forward ->
fused DDIM update, for a batch of B clips (B = 4 = the reference's eval batch, config/eval_traj_image_2b_finetune.yaml:32).
Inputs are synthetic (seed 42), weights random N(0, 0.02^2) of the real architecture, all resident in HBM
before the timed region.  N > 1 = N independent replicas (inference has no collective, evaluation_control_to_video.py:212-222);
value = N*B*K / max-over-ranks(wall).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: algorithmic FLOPs / live HIP-event duration vs the
2.5 PFLOP/s dense bf16 MFMA peak) and, at N=1, `cpu_baseline` (the CPU oracle timed on the host cores - a reported
baseline, not the target).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
CFG_2B = dict(num_attention_heads=30, attention_head_dim=64, in_channels=32, out_channels=16, time_embed_dim=512,
              text_embed_dim=4096, num_layers=30, sample_width=60, sample_height=40, sample_frames=17, patch_size=2,
              max_text_seq_length=226, modulate_encoder_hidden_states=True,
              loaded_pretrained_model_name_or_path="THUDM/CogVideoX-2b")
# BASELINE configs[4]: CogVideoX1.5-5B-I2V (config/traj_image_5b_finetune.yaml:15) on DROID 256x384x29f clips
CFG_5B = dict(num_attention_heads=48, attention_head_dim=64, in_channels=32, out_channels=16, time_embed_dim=512,
              text_embed_dim=4096, num_layers=42, sample_width=48, sample_height=32, sample_frames=29, patch_size=2, patch_size_t=2,
              ofs_embed_dim=512, use_rotary_positional_embeddings=True, max_text_seq_length=226,
              modulate_encoder_hidden_states=True, loaded_pretrained_model_name_or_path="THUDM/CogVideoX1.5-5b-I2V")
SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
             clip_sample=False, set_alpha_to_one=True, prediction_type="v_prediction", rescale_betas_zero_snr=True,
             snr_shift_scale=3.0, timestep_spacing="trailing")


def flops_per_sample(cfg, S):
    """Algorithmic forward FLOPs per sample per step of attention + FFN (the north-star numerator)."""
    D, L = cfg["num_attention_heads"] * cfg["attention_head_dim"], cfg["num_layers"]
    return L * (24 * S * D * D + 4 * S * S * D)


def synthetic_inputs(B, dev, dtype, frames=5, h=40, w=60):
    g = torch.Generator().manual_seed(42)
    latents = torch.randn(B, frames, 16, h, w, generator=g)
    image_latents = torch.zeros(B, frames, 16, h, w)
    image_latents[:, 0] = torch.randn(B, 16, h, w, generator=g) * 1.15258426
    prompt = torch.randn(B, 226, 4096, generator=g) * 0.2
    actions = torch.randn(B, 4 * (frames - 1), 7, generator=g) * torch.tensor([20.0] * 6 + [1.0])
    actions[..., 6].clamp_(0, 1)
    return (latents.to(dev, dtype), image_latents.to(dev, dtype), prompt.to(dev, dtype), actions.to(dev))


def build_model(cfg, dev):
    from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj
    torch.manual_seed(42)
    with torch.device(dev):
        m = CogVideoXTransformer3DModelTraj(**cfg)
    g = torch.Generator(device=dev).manual_seed(42)
    for name, p in m.named_parameters():
        if p.ndim >= 2:
            p.data.normal_(0, 0.02, generator=g)          # also makes zero-init layers non-trivial
        elif name.endswith("bias"):
            p.data.normal_(0, 0.02, generator=g)
    m = m.to(torch.bfloat16).eval()
    m.action_embed.forced_mask = torch.zeros(64, dtype=torch.bool)[:0]   # placeholder, set per batch below
    return m


def cpu_baseline(cfg, layers, threads):
    """Oracle (kind 'port': the reference's own Python cannot run without diffusers) timed on the host cores: ONE fp32
    denoise-step forward for ONE clip through `layers` of the 30 blocks, scaled to 30 layers."""
    from oracle import dit
    torch.set_num_threads(threads)
    c = {**cfg, "num_layers": layers}
    D, E = 1920, 512
    g = torch.Generator().manual_seed(0)
    sd = {}

    def lin(name, o, i):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * 0.02
        sd[name + ".bias"] = torch.zeros(o)

    def ln(name, d):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(d), torch.zeros(d)

    sd["patch_embed.proj.weight"] = torch.randn(D, 32, 2, 2, generator=g) * 0.02
    sd["patch_embed.proj.bias"] = torch.zeros(D)
    lin("patch_embed.text_proj", D, 4096), lin("time_embedding.linear_1", E, D), lin("time_embedding.linear_2", E, E)
    for i in range(layers):
        p = f"transformer_blocks.{i}."
        lin(p + "norm1.linear", 6 * D, E), ln(p + "norm1.norm", D), lin(p + "norm2.linear", 6 * D, E), ln(p + "norm2.norm", D)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + "attn1." + n, D, D)
        ln(p + "attn1.norm_q", 64), ln(p + "attn1.norm_k", 64)
        lin(p + "ff.net.0.proj", 4 * D, D), lin(p + "ff.net.2", D, 4 * D)
    ln("norm_final", D), lin("norm_out.linear", 2 * D, E), ln("norm_out.norm", D), lin("proj_out", 64, D)
    lin("action_embed.mlp.0", 4 * E, 28), lin("action_embed.mlp.3", E, 4 * E)
    sd["action_embed.mask_embed.weight"] = torch.zeros(1, E)
    x = torch.randn(1, 5, 32, 40, 60, generator=g)
    e = torch.randn(1, 226, 4096, generator=g) * 0.2
    a = torch.randn(1, 16, 7, generator=g)
    t = torch.tensor([500])

    def timed(n_layers, n_threads, dtype):
        torch.set_num_threads(n_threads)
        keep = lambda k: not k.startswith("transformer_blocks.") or int(k.split(".")[1]) < n_layers
        w = {k: v.to(dtype) for k, v in sd.items() if keep(k)}
        with torch.no_grad():
            t0 = time.perf_counter()
            dit.dit_forward(w, {**cfg, "num_layers": n_layers}, x.to(dtype), e.to(dtype), t, actions=a.to(dtype),
                            is_mask=torch.zeros(1, dtype=torch.bool))
            d = time.perf_counter() - t0
        return d, d * 30.0 / n_layers

    # thread sweep on ONE block each (the 128-thread run of earlier rounds was the slowest point of this curve: oversubscribed),
    # then the sample proper - `layers` of the 30 blocks - with the best count; bf16 beside fp32
    sweep = {}
    for nthr in sorted({n for n in (8, 16, 32, 64, threads) if n <= max(threads, 8)}):
        sweep[nthr] = timed(1, nthr, torch.float32)[0]
    best = min(sweep, key=sweep.get)
    dt, per_step = timed(layers, best, torch.float32)
    dt_bf, per_bf = timed(max(1, layers // 3), best, torch.bfloat16)
    torch.set_num_threads(threads)
    return {"value": 1.0 / per_step, "unit": "denoise-steps/s", "cores": best, "kind": "port",
            "sample": f"1 clip x 1 step, fp32 eager PyTorch oracle, {layers}/30 blocks timed ({dt:.2f} s)" + ("" if layers == 30 else " and scaled to 30") + "; "
                      f"threads = best of a one-block sweep",
            "s_per_step": per_step, "thread_sweep_s_per_block": {str(k): round(v, 3) for k, v in sweep.items()},
            "variants": [{"dtype": "bf16", "cores": best, "s_per_step": per_bf, "sample": f"{max(1, layers // 3)}/30 blocks ({dt_bf:.2f} s)"}]}


def live_pmc(dom_symbol, timeout_s=420):
    """MFMA-busy fraction, clock and fabric bytes per launch of the dominant kernel symbol, measured NOW: three separate
    `rocprofv3 --pmc` passes (SQ + GRBM | FETCH_SIZE | WRITE_SIZE; no tracing flags - MI355X_MICROARCH.md "rocprofv3 PMC slots")
    of a two-step eager run of this same script in a child process.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64
    B) and WRITE_SIZE calibrated on ln_mod_kernel's known write volume, as tools/pmc_bench.py does.  Returns None when
    rocprofv3 is not on PATH or a pass fails; the caller then falls back to the committed profile."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    out = tempfile.mkdtemp(prefix="orv_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--eager", "--no-legs", "--no-vae",
             "--no-cpu-baseline", "--no-pmc"]
    sym = lambda name: name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    res = {}
    t_start = time.time()
    try:
        for name, ctrs in (("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
            left = timeout_s - (time.time() - t_start)
            if left < 30:
                break
            subprocess.run([exe, "--pmc", *ctrs, "--output-format", "csv", "-d", out, "-o", name, "--", *child], cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, timeout=left, check=True)
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            dur, seen = collections.defaultdict(float), set()
            n = collections.defaultdict(int)
            for f in glob.glob(os.path.join(out, "**", f"{name}_counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for r in csv.DictReader(fh):
                        k = sym(r["Kernel_Name"])
                        if k != dom_symbol and not k.startswith("ln_mod_kernel"):
                            continue
                        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                        if r["Dispatch_Id"] not in seen:
                            seen.add(r["Dispatch_Id"])
                            n[k] += 1
                            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            res[name] = (agg, dur, n)
        if "sq" not in res or dom_symbol not in res["sq"][0]:
            return None
        agg, dur, n = res["sq"]
        gui = agg[dom_symbol]["GRBM_GUI_ACTIVE"] / 8.0
        r = {"mfma_busy": round(agg[dom_symbol]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui), 4) if gui else None,
             "clock_ghz": round(gui / dur[dom_symbol], 3) if dur[dom_symbol] else None, "launches_under_pmc": n[dom_symbol],
             "avg_us_under_pmc": round(dur[dom_symbol] / n[dom_symbol] / 1e3, 2), "traffic": None}
        if "fetch" in res and "write" in res and dom_symbol in res["fetch"][0] and dom_symbol in res["write"][0]:
            fa, _, fn = res["fetch"]
            wa, _, wn = res["write"]
            cal = 1.0
            lnk = next((k for k in wa if k.startswith("ln_mod_kernel")), None)
            if lnk and wn[lnk]:
                known = 4 * 3226 * 1920 * 2
                meas = wa[lnk]["WRITE_SIZE"] / wn[lnk] * 1024
                if meas > 0 and 0.9 < known / meas < 1.1:
                    cal = known / meas
            fb = 2.0 * 1024 * fa[dom_symbol]["FETCH_SIZE"] / fn[dom_symbol]
            wb = cal * 1024 * wa[dom_symbol]["WRITE_SIZE"] / wn[dom_symbol]
            r.update(traffic=int(fb + wb), fetch_bytes=int(fb), write_bytes=int(wb), write_calibration=round(cal, 4))
        return r
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def ranks_seen(world, dev):
    """World size as the process group (RCCL) actually sees it: every rank adds 1 (collective; call on all ranks)."""
    if world == 1:
        return 1
    import torch.distributed as dist
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    return int(t.item())


def vae_decode_leg(latents, dev, loop_wall, steps, world, B, barrier):
    """SURVEY §8(f) rank 1, reported BESIDE the headline (never inside `value`): the MI355X VAE (orv_amd.vae, real 2B geometry,
    random weights) decodes the B clips [B,16,5,40,60] -> [B,3,17,320,480]; frames/s of a whole 50-step sample including it =
    17 B / (50 x step time + decode time)."""
    from orv_amd.vae import AutoencoderKLCogVideoX
    torch.manual_seed(1)
    with torch.device(dev):
        vae = AutoencoderKLCogVideoX()
    g = torch.Generator(device=dev).manual_seed(1)
    for p in vae.parameters():
        if p.ndim > 1:
            p.data.normal_(0, 1.0 / (p[0].numel() ** 0.5), generator=g)
    vae = vae.to(torch.bfloat16).eval()
    z = (latents.permute(0, 2, 1, 3, 4) / 1.15258426).contiguous()            # decode_latents layout (:1476-1479)

    def timed_decode():
        vae.decode(z)                      # untimed: weight packing, allocator growth and code-object loads at THIS batch size
        barrier()
        t0 = time.perf_counter()
        out = vae.decode(z).sample
        barrier()
        dt = time.perf_counter() - t0
        assert out.shape == (B, 3, 17, 320, 480) and torch.isfinite(out.float()).all()
        return dt
    dt_plain = timed_decode()
    vae.enable_slicing(); vae.enable_tiling()          # what the reference's entry points do (inference_control_to_video.py:98-99):
    dt = timed_decode()                                # a 40 x 60 latent is 4 tiles of <= 30 x 45 (1.29 x the voxels) + seam blends
    step_s = loop_wall / steps
    return {"ms_per_clip": round(1e3 * dt / B, 2), "ms_per_clip_untiled": round(1e3 * dt_plain / B, 2),
            "tiling": "enable_tiling() as the reference: 4 tiles 30x45 / 30x24 / 15x45 / 15x24 latent, blend 40 / 72 px",
            "parity": "no reference fixture can exist (diffusers absent); oracle/vae.py and the HIP decode are pinned by an independent naive derivation (tools/make_vae_naive.py, tests/test_vae_naive_golden.py)",
            "frames_per_sec_50_steps_incl_decode": round(17.0 * world * B / (50 * step_s + dt), 3),
            "frames_per_sec_50_steps_excl_decode": round(17.0 * world * B / (50 * step_s), 3)}


def train_mode(args, model, latents, image_latents, prompt, actions, sched, dev, rank, world, cfg=None, leg=None):
    """BASELINE configs[2]: CogVideoX-2B SFT step (train_cogvideox_control_to_video_sft.py:1005-1104), B clips per GPU, bf16
    params/grads, data parallel: forward+backward through the HIP kernels, ONE bucketed RCCL all-reduce of the gradients,
    global-norm clip + fused AdamW.  value = trained clips per second (all GPUs)."""
    import torch.distributed as dist
    from orv_amd.optim import FusedAdamW
    B = args.batch
    model.train()
    opt = FusedAdamW(model.parameters(), lr=1e-5, betas=(0.9, 0.95), weight_decay=1e-3, max_grad_norm=1.0)
    from orv_amd import sft
    g = torch.Generator(device=dev).manual_seed(42 + rank)
    batch = sft.Batch(latents, image_latents, prompt, actions, None, None,
                      torch.ones(latents.shape[1], dtype=torch.bool, device=dev), 1)

    is5b = cfg is not None and cfg.get("patch_size_t") is not None
    if getattr(args, "grad_ckpt", False):
        model.enable_gradient_checkpointing()       # block-level activation recompute (BASELINE configs[4] names it)

    def step():
        # orv_amd.sft.sft_step = train script :1005-1104 (noise + timestep draw, add_noise, forward, x0 loss, backward,
        # gradient all-reduce when world > 1, global-norm clip, fused AdamW)
        return sft.sft_step(model, sched, opt, batch, generator=g, data_parallel=world > 1, use_rope=is5b, is_ofs_embed=is5b)[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    n_warm, n_steps = (leg if leg is not None else (args.warmup, args.steps))
    torch.cuda.reset_peak_memory_stats()
    for _ in range(n_warm):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        loss = step()
    barrier()
    wall = time.perf_counter() - t0
    if leg is not None:            # `train` leg of the default (denoise) line: configs[2] at B clips, N = 1, a few steps
        c_ = cfg or {**CFG_2B, "num_layers": args.layers}
        fl = 3.0 * flops_per_sample(c_, 226 + 3000)
        return {"workload": "configs[2]: CogVideoX-2B SFT step (fwd + bwd + global-norm clip + fused AdamW), bf16 params + grads",
                "batch_per_gpu": B, "warmup": n_warm, "steps": n_steps, "ms_per_step": round(1e3 * wall / n_steps, 2),
                "clips_per_sec": round(B * n_steps / wall, 3), "achieved_tflops_attn_ffn": round(B * n_steps / wall * fl / 1e12, 1),
                "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "final_loss": float(loss)}
    if world > 1:
        w = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    seen = ranks_seen(world, dev)
    assert seen == world, f"{seen} of {world} ranks answered the census"
    # the gradient exchange of the LAST step (sharding.FlatGradReducer): collectives, payload, exposed (non-overlapped) time
    exch = getattr(opt, "last_exchange", None)
    exch = exch.stats() if exch is not None else {"collectives": 0, "bytes": 0, "exposed_ms": None}
    if world > 1 and exch["exposed_ms"] is not None:
        w = torch.tensor([exch["exposed_ms"]], device=dev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        exch["exposed_ms"] = float(w.item())
    if rank == 0:
        c_ = cfg or {**CFG_2B, "num_layers": args.layers}
        S_ = 226 + (latents.shape[1] // (c_.get("patch_size_t") or 1)) * (latents.shape[3] // 2) * (latents.shape[4] // 2)
        fl = 3.0 * flops_per_sample(c_, S_)
        value = world * B * args.steps / wall
        print(json.dumps({
            "metric": "train-clips/sec", "value": round(value, 3), "unit": "clips/s (SFT step: fwd+bwd+allreduce+AdamW)",
            "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "achieved_tflops_attn_ffn": round(value * fl / 1e12, 1), "final_loss": float(loss),
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "exchange": {"collective": "all-reduce (SUM) of the flat bf16 gradient buffer, RCCL, overlapped with the backward",
                         "collectives_per_step": exch["collectives"], "bytes_per_rank_per_step": exch["bytes"],
                         "exposed_ms_max_over_ranks": None if exch["exposed_ms"] is None else round(exch["exposed_ms"], 3),
                         "note": "N = 1: no exchange is issued"},
            "config": {"workload": ("configs[4]: CogVideoX1.5-5B SFT step, DROID 256x384x29f latents [B,8,16,32,48], p_t=2, RoPE, "
                                    "ofs, bf16 params+grads, " + ("activation checkpointing (block-level recompute)" if getattr(args, "grad_ckpt", False) else "all activations resident (no recompute)")) if is5b else
                                   "configs[2]: CogVideoX-2B SFT step, 320x480x17f latents, bf16 params+grads, DP",
                       "batch_per_gpu": B, "num_layers": c_["num_layers"], "seq_len": S_,
                       "parallelism": f"dp{world} (one RCCL all-reduce/step)",
                       "valid": c_["num_layers"] == (42 if is5b else 30)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def self_launch(n: int) -> int:
    """Re-run this command line as N ranks of `python -m torch.distributed.run` on this node (the shape the driver itself
    uses for N > 1): one process per GPU, rendezvous on 127.0.0.1, HIP_VISIBLE_DEVICES untouched (each rank picks
    cuda:LOCAL_RANK)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def dry_run(rank: int, world: int, args) -> int:
    """Launcher check without a GPU: every rank joins a gloo group and contributes 1 to an all-reduce."""
    seen = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "ranks_seen": seen, "mode": args.mode}), flush=True)
    return 0 if seen == args.gpus else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU per step (reference eval batch = 4, demo = 1)")
    ap.add_argument("--layers", type=int, default=30, help="debug only; anything but 30 marks the line INVALID")
    ap.add_argument("--mode", choices=["denoise", "train"], default="denoise",
                    help="denoise (headline, BASELINE configs[1]) or train (configs[2]: one SFT step = fwd+bwd+all-reduce+AdamW)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="(default since round 3) replay the transformer forward from a HIP graph")
    ap.add_argument("--eager", action="store_true", help="time the eager launch path instead of the HIP-graph replay")
    ap.add_argument("--no-legs", action="store_true", help="skip the b1 / train legs reported beside the headline (N = 1 only)")
    ap.add_argument("--cpu-baseline-layers", type=int, default=30, help="blocks of the CPU oracle that are timed (30 = the whole step)")
    ap.add_argument("--model", choices=["2b", "5b"], default="2b",
                    help="2b = the headline (BASELINE configs[1]/[2]); 5b = configs[4] (CogVideoX1.5-5B, DROID 256x384x29f, p_t=2, "
                         "RoPE, ofs) - train mode only")
    ap.add_argument("--grad-ckpt", dest="grad_ckpt", action="store_true",
                    help="train mode: gradient checkpointing (block inputs kept, activations recomputed in the backward)")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE-decode leg (frames/s including decode)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the three live rocprofv3 --pmc passes for roofline.traffic / mfma_busy")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch the ranks, rendezvous (gloo, no GPU work), print the ranks seen and exit: checks the launcher")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start N ranks (one per GPU) of this same command under torch.distributed.run
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(rank, world, args)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ORV_DIST_BACKEND", "nccl")     # "gloo": functional test of the N > 1 path on ONE GPU (ranks share it)
        if backend != "nccl" or os.environ.get("ORV_SAME_GPU"):   # ORV_SAME_GPU=1: the ranks share the visible GPUs also under nccl (tools/multi_rank_check.sh)
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL; only used for barrier/max
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from orv_amd import ops, schedulers
    from orv_amd._lib import LIB_PATH, check, lib
    check(lib().orv_device_check(local), "orv_device_check")

    B = args.batch
    if args.model == "5b":
        assert args.mode == "train", "--model 5b is the configs[4] training line (use --mode train)"
        cfg = {**CFG_5B, "num_layers": 42 if args.layers == 30 else args.layers}
        model = build_model(cfg, dev)
        model.action_embed.forced_mask = torch.zeros(B, dtype=torch.bool)
        latents, image_latents, prompt, actions = synthetic_inputs(B, dev, torch.bfloat16, frames=8, h=32, w=48)
        sched = schedulers.CogVideoXDDIMScheduler(**{**SCHED, "snr_shift_scale": 1.0})
        return train_mode(args, model, latents, image_latents, prompt, actions, sched, dev, rank, world, cfg=cfg)
    cfg = {**CFG_2B, "num_layers": args.layers}
    model = build_model(cfg, dev)
    model.action_embed.forced_mask = torch.zeros(B, dtype=torch.bool)       # is_mask forced False
    latents, image_latents, prompt, actions = synthetic_inputs(B, dev, torch.bfloat16)
    sched = schedulers.CogVideoXDDIMScheduler(**SCHED)
    sched.set_timesteps(50)
    ts = sched.timesteps.tolist()
    controls = {"actions": actions}

    # The timed path is what the sampler runs: the transformer forward replayed from a HIP graph (GraphedTransformer: bit-identical to
    # the eager launches, tests/test_gpu_model.py), which removes ~1.2 ms of launch gaps per step.  The per-kernel timeline (HIP
    # events around every GEMM / attention launch) cannot be taken inside a graph: it comes from a SECOND, eager loop after the
    # timed region.  --eager times the eager path instead.
    use_graph = not args.eager
    from orv_amd.cogvideox_control import CogVideoXImageToVideoPipelineTraj

    def default_forward(m):
        """The transformer forward exactly as ``CogVideoXImageToVideoPipelineTraj.__call__`` makes it (its ``transformer_forward``): by
        DEFAULT - nothing enabled by the caller - that is HIP-graph replay; --eager forbids it the way a user would
        (``enable_hip_graph(False)``)."""
        p_ = CogVideoXImageToVideoPipelineTraj(transformer=m, scheduler=sched)
        if not use_graph:
            p_.enable_hip_graph(False)
        return p_.transformer_forward

    fwd = default_forward(model)

    def make_step(fn, Bn, lat_img, prm, ctl):
        @torch.no_grad()                    # the reference sampler runs under torch.no_grad (cogvideox_control.py:1228)
        def step(i, lat):
            t = ts[i % len(ts)]
            model_in = torch.cat([lat, lat_img], dim=2)                       # cogvideox_control.py:1409-1413
            tvec = torch.full((Bn,), t, device=dev, dtype=torch.int64)
            v = fn(hidden_states=model_in, encoder_hidden_states=prm, timestep=tvec,
                   controls_or_guidances=ctl, return_dict=False)[0]
            return sched.step(v, t, lat, return_dict=False)[0]
        return step

    step = make_step(fwd, B, image_latents, prompt, controls)
    eager_step = make_step(model, B, image_latents, prompt, controls)

    if args.mode == "train":
        return train_mode(args, model, latents, image_latents, prompt, actions, sched, dev, rank, world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    lat = latents
    for i in range(max(args.warmup, 3 if use_graph else 0)):        # a graph needs one eager + one capturing call before it replays
        lat = step(i, lat)
    barrier()
    # per-step HIP events on the launch stream (median + p10 / p90): K + 1 events around the K steps; recording an event
    # is a stream marker, not a synchronisation - the loop below is still one uninterrupted submission sequence
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    lat = latents
    evs[0].record()
    for i in range(args.steps):
        lat = step(i, lat)
        evs[i + 1].record()
    barrier()
    wall = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    pct = lambda q: step_ms[min(len(step_ms) - 1, max(0, int(round(q * (len(step_ms) - 1)))))]
    step_stats = {"median_ms": round(pct(0.5), 3), "p10_ms": round(pct(0.1), 3), "p90_ms": round(pct(0.9), 3),
                  "min_ms": round(step_ms[0], 3), "max_ms": round(step_ms[-1], 3)}
    assert torch.isfinite(lat.float()).all(), "non-finite latents"
    # per-kernel timeline: eager loop, outside the timed region (same kernels, same shapes, same stream)
    n_tl = args.steps if not use_graph else min(args.steps, 10)
    lat2 = latents
    for i in range(2):
        lat2 = eager_step(i, lat2)
    torch.cuda.synchronize()
    ops.start_timeline()
    t1 = time.perf_counter()
    for i in range(n_tl):
        lat2 = eager_step(i, lat2)
    torch.cuda.synchronize()
    eager_ms = 1e3 * (time.perf_counter() - t1) / n_tl
    timeline = ops.stop_timeline()
    if world > 1:
        w = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    seen = ranks_seen(world, dev)
    softmax_census = model.softmax_kernel_census(rope=False)      # which softmax form the timed weights selected, layer by layer
    vae_leg = None
    if not args.no_vae and args.layers == 30:
        vae_leg = vae_decode_leg(lat, dev, wall, args.steps, world, B, barrier)
    legs = {}
    if world == 1 and not args.no_legs and args.layers == 30:
        # b1: the demo shape (inference_control_to_video.py runs ONE clip), graph-replayed
        l1, il1, p1, a1 = synthetic_inputs(1, dev, torch.bfloat16)
        model.action_embed.forced_mask = torch.zeros(1, dtype=torch.bool)
        s1 = make_step(default_forward(model), 1, il1, p1, {"actions": a1})
        x1 = l1
        for i in range(3):
            x1 = s1(i, x1)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(20):
            x1 = s1(i, x1)
        torch.cuda.synchronize()
        legs["b1"] = {"workload": "configs[1] at B = 1 (the demo entry point), 20 steps", "ms_per_step": round(1e3 * (time.perf_counter() - tb) / 20, 3),
                      "timed_path": "hip-graph replay" if use_graph else "eager"}
        del s1, x1
        model.action_embed.forced_mask = torch.zeros(B, dtype=torch.bool)

        def timed_steps(stepfn, x0, n_warm, n):
            x = x0
            for i in range(n_warm):
                x = stepfn(i, x)
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for i in range(n):
                x = stepfn(i, x)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t_) / n

        # attn_online: the same headline step when the qk-LayerNorm parameters give NO usable score bound (a checkpoint whose
        # max|gamma_q| max|gamma_k| exceeds ~7.6: the fixed-shift softmax is not valid and every block runs the online-softmax
        # kernel) - Attention.score_bound is overridden to "unknown" for this leg only
        from orv_amd import cogvideox_control as cc
        keep_sb = cc.Attention.score_bound
        try:
            cc.Attention.score_bound = lambda self, scale, rope=True: None
            so = make_step(default_forward(model), B, image_latents, prompt, controls)
            legs["attn_online"] = {"workload": "configs[1] at B = %d with the online-softmax attention kernel in every block (no score bound), 10 steps" % B,
                                   "ms_per_step": round(timed_steps(so, latents, 3, 10), 3),
                                   "fixed_shift_valid_up_to_log2_units": 90.0, "random_init_bound_log2_units": round(float(keep_sb(model.transformer_blocks[0].attn1, 0.125)), 2)}
            del so
        finally:
            cc.Attention.score_bound = keep_sb
        # attn_mixed: trained-LIKE qk-LayerNorm gains (no checkpoint is reachable here): per layer max|gamma_q gamma_k| drawn log-uniformly in
        # [1, 16] so that the layers' bounds straddle the fixed-shift limit; the parameters are restored afterwards
        try:
            gmix = torch.Generator().manual_seed(46)
            saved = []
            with torch.no_grad():
                for blk in model.transformer_blocks:
                    at_ = blk.attn1
                    saved.append((at_.norm_q.weight.detach().clone(), at_.norm_k.weight.detach().clone()))
                    tgt = float(torch.exp(torch.rand(1, generator=gmix) * math.log(16.0)))
                    pert = 1.0 + 0.15 * torch.randn(2, 64, generator=gmix)
                    at_.norm_q.weight.copy_((pert[0].abs() * math.sqrt(tgt)).to(dev, torch.bfloat16))
                    at_.norm_k.weight.copy_((pert[1].abs() * math.sqrt(tgt)).to(dev, torch.bfloat16))
            census = model.softmax_kernel_census(rope=False)
            sm = make_step(default_forward(model), B, image_latents, prompt, controls)
            legs["attn_mixed"] = {"workload": "configs[1] at B = %d with trained-like qk-LayerNorm gains (per-layer max|gamma_q gamma_k| log-uniform in [1, 16]), 10 steps" % B,
                                  "ms_per_step": round(timed_steps(sm, latents, 3, 10), 3), **census}
            del sm
        except Exception as e:
            legs["attn_mixed"] = {"error": repr(e)[:300]}
        finally:
            with torch.no_grad():
                for blk, (wq_, wk_) in zip(model.transformer_blocks, saved):
                    blk.attn1.norm_q.weight.copy_(wq_)
                    blk.attn1.norm_k.weight.copy_(wk_)
        # cond: BASELINE configs[3] (config/traj_image_condfull_2b_finetune.yaml: visual_guidance, control_keys depth + label;
        # cogvideox_control.py:828-858) - a second 2B model with the guidance fuse, depth / label maps through the same patch-embed
        try:
            torch.manual_seed(43)
            mc = build_model({**cfg, "visual_guidance": True, "num_control_keys": 2}, dev)
            mc.action_embed.forced_mask = torch.zeros(B, dtype=torch.bool)
            gcond = torch.Generator().manual_seed(44)
            depths = torch.randn(B, 5, 32, 40, 60, generator=gcond).to(dev, torch.bfloat16)
            labels = torch.randn(B, 5, 32, 40, 60, generator=gcond).to(dev, torch.bfloat16)
            sc = make_step(default_forward(mc), B, image_latents, prompt,
                           {"actions": actions, "depths": depths, "labels": labels})
            legs["cond"] = {"workload": "configs[3]: occupancy-conditioned CogVideoX-2B (visual_guidance, depth + label control maps), B = %d, 5 steps" % B,
                            "ms_per_step": round(timed_steps(sc, latents, 3, 5), 3), "timed_path": "hip-graph replay" if use_graph else "eager"}
            del sc, mc, depths, labels
        except Exception as e:      # a leg must never take the headline line down
            legs["cond"] = {"error": repr(e)[:300]}
        # train: configs[2] on this GPU (the fused optimizer moves the parameters into its flat buffers: after the inference legs)
        legs["train"] = train_mode(args, model, latents, image_latents, prompt, actions, sched, dev, rank, world, leg=(3, 12))      # 12 timed steps: resolves a 2 % change
        # train_5b_ckpt: configs[4] (CogVideoX1.5-5B, DROID 256x384x29f, p_t = 2, RoPE, ofs, activation checkpointing), 1 + 2 steps
        try:
            import copy
            import gc
            del fwd, step, eager_step
            model = None
            gc.collect()
            torch.cuda.empty_cache()
            a5 = copy.copy(args)
            a5.grad_ckpt = True
            cfg5 = {**CFG_5B, "num_layers": 42}
            m5 = build_model(cfg5, dev)
            m5.action_embed.forced_mask = torch.zeros(B, dtype=torch.bool)
            l5, il5, p5, ac5 = synthetic_inputs(B, dev, torch.bfloat16, frames=8, h=32, w=48)
            sched5 = schedulers.CogVideoXDDIMScheduler(**{**SCHED, "snr_shift_scale": 1.0})
            r5 = train_mode(a5, m5, l5, il5, p5, ac5, sched5, dev, rank, world, cfg=cfg5, leg=(1, 2))
            r5["workload"] = ("configs[4]: CogVideoX1.5-5B SFT step, DROID 256x384x29f latents [B,8,16,32,48], p_t=2, RoPE, ofs, bf16 "
                              "params + grads, activation checkpointing (block-level recompute)")
            S5 = 226 + (8 // 2) * 16 * 24
            r5["achieved_tflops_attn_ffn"] = round(r5["clips_per_sec"] * 3.0 * flops_per_sample(cfg5, S5) / 1e12, 1)
            legs["train_5b_ckpt"] = r5
            del m5
        except Exception as e:
            legs["train_5b_ckpt"] = {"error": repr(e)[:300]}

    if rank == 0:
        S = 226 + 3000
        total_steps = world * B * args.steps
        value = total_steps / wall
        fl = flops_per_sample(cfg, S)
        # per-kernel live timings -> dominant kernel roofline
        kernels = []
        # the attention symbol orv_attention_fwd_bounded launches for this model (random-init qk-LayerNorm: bound ~11.8 log2 units)
        attn_sym = "attn_fwd_pp_kernel<true>" if os.environ.get("ORV_ATTN_PP", "1") != "0" and os.environ.get("ORV_ATTN_STATIC", "1") != "0" \
            else ("attn_fwd_pp_kernel<false>" if os.environ.get("ORV_ATTN_PP", "1") != "0" else "attn_fwd_v2_kernel")
        for key, ms in timeline.items():
            if key[0] == "gemm":
                M, N, K, epi = key[1:5]
                ap, cp = (key[5], key[6]) if len(key) > 5 else (0, 0)
                # kernel symbol as orv_gemm_bf16 dispatches it (cost-model tile choice, packed-operand flags), so it matches the rocprofv3 name
                sym = ops.gemm_kernel_name(M, N, K, epi, ap, cp)
                flop, name = 2.0 * M * N * K, f"{sym} M={M} N={N} K={K}" + (" packed A" if ap else "") + (" packed C" if cp else "")
            else:
                _, b, s, h = key
                flop, name = 4.0 * b * h * s * s * 64, f"{attn_sym} B={b} S={s} H={h}"
            avg = sum(ms) / len(ms)
            kernels.append({"kernel": name, "launches": len(ms), "avg_ms": round(avg, 4), "total_ms": round(sum(ms), 2),
                            "tflops": round(flop / avg / 1e9, 1)})
        kernels.sort(key=lambda k: -k["total_ms"])
        # roofline of the dominant KERNEL SYMBOL (all its shapes together), so that launches / average duration are the
        # same quantities rocprofv3 --stats prints for that name (profiles/r1_bench_kernel_stats_summary.txt)
        by_sym = {}
        for k in kernels:
            sym = k["kernel"].split(" M=")[0].split(" B=")[0]
            e = by_sym.setdefault(sym, {"kernel": sym, "launches": 0, "total_ms": 0.0, "flop": 0.0, "shapes": []})
            e["launches"] += k["launches"]
            e["total_ms"] += k["total_ms"]
            e["flop"] += k["tflops"] * 1e9 * k["avg_ms"] * k["launches"]
            e["shapes"].append(k["kernel"][len(sym):].strip())
        dom = max(by_sym.values(), key=lambda e: e["total_ms"]) if by_sym else None
        if dom:
            dom["avg_ms"] = round(dom["total_ms"] / dom["launches"], 4)
            dom["tflops"] = round(dom["flop"] / dom["total_ms"] / 1e9, 1)
        # PMC figures of the dominant symbol: measured live (three rocprofv3 --pmc passes of a two-step child run) when rocprofv3 is
        # on PATH, else read from the committed profile of this same command (tools/pmc_bench.sh)
        traffic = mfma_busy = clock = None
        pmc_source = None
        if dom and world == 1 and not args.no_pmc and args.layers == 30:
            pm = live_pmc(dom["kernel"])
            if pm is not None:
                traffic, mfma_busy, clock = pm.get("traffic"), pm.get("mfma_busy"), pm.get("clock_ghz")
                pmc_source = {"how": "live: rocprofv3 --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE) of a "
                                     "two-step eager child run after the timed region; FETCH x 2, WRITE calibrated on ln_mod_kernel", **pm}
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if dom and (traffic is None or mfma_busy is None) and os.path.exists(prof):
            try:
                ent = json.load(open(prof)).get(dom["kernel"])
                if ent is not None:
                    if traffic is None:
                        traffic = int(ent.get("total_bytes"))
                    if mfma_busy is None:
                        mfma_busy, clock = ent.get("mfma_busy"), ent.get("clock_ghz")
                    pmc_source = pmc_source or ("roofline.traffic / mfma_busy / clock_ghz_under_pmc are read from profiles/hbm_traffic.json "
                                                "(separate rocprofv3 --pmc passes of this command, tools/pmc_bench.sh), not re-measured in this run")
            except Exception:
                pass
        line = {
            "metric": "denoise-steps/sec", "value": round(value, 3), "unit": "steps/s (clips x denoise steps per second)",
            "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall / args.steps, 3), **step_stats,
            "step_stats": "median / p10 / p90 / min / max of the K per-step HIP-event intervals on the launch stream (rank 0)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "frames_per_sec": round(17.0 * world * B * args.steps / 50.0 / wall, 3),
            "achieved_tflops_attn_ffn": round(value * fl / 1e12, 1),
            "frac_mfma_peak_attn_ffn": round(value * fl / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
            "config": {"workload": "configs[1]: CogVideoX-2B singleview 320x480x17f (latents [B,5,32,40,60], S=3226), "
                                   "DDIM 50-step schedule, guidance 1.0, trajectory-conditioned",
                       "batch_per_gpu": B, "num_layers": args.layers, "parallelism": f"replica x{world} (no collective)",
                       "valid": args.layers == 30},
            "roofline": None if dom is None else {
                "bound": "mfma", "kernel": dom["kernel"], "shapes": dom["shapes"], "achieved": dom["tflops"],
                "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "mfma_busy": mfma_busy, "clock_ghz_under_pmc": clock, "launches": dom["launches"], "avg_ms": dom["avg_ms"]},
            "kernels": kernels[:8],
            "timed_path": ("default __call__ path: CogVideoXImageToVideoPipelineTraj.transformer_forward -> HIP-graph replay of the transformer forward "
                           "(automatic, bit-identical to eager)") if use_graph else "eager launches (enable_hip_graph(False))",
            "eager_ms_per_step": round(eager_ms, 3),
            "kernel_timeline": f"HIP events on the launch stream over a separate eager loop of {n_tl} steps after the timed region",
            "lib": {"path": os.path.relpath(LIB_PATH, ROOT), "orv_version": int(lib().orv_version())},
            "pmc_source": pmc_source,
            "attention_softmax": {**softmax_census, "note": "random-init qk-LayerNorm (gamma 1): every layer takes the fixed-shift kernel; "
                                  "a checkpoint layer whose bound exceeds the limit runs the online kernel (leg attn_online = all of them, "
                                  "leg attn_mixed = trained-like gains straddling the limit)"},
            "vae_decode": vae_leg,
            "b1": legs.get("b1"), "attn_online": legs.get("attn_online"), "attn_mixed": legs.get("attn_mixed"), "cond": legs.get("cond"), "train": legs.get("train"),
            "train_5b_ckpt": legs.get("train_5b_ckpt"),
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, (os.cpu_count() or 2) // 2)
            line["cpu_baseline"] = cpu_baseline(CFG_2B, args.cpu_baseline_layers, threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
