"""GPU: the MI355X VAE (orv_amd.vae.AutoencoderKLCogVideoX) against the CPU oracle (oracle/vae.py) on
random weights.  PARITY UNPINNED: the oracle restates diffusers' published AutoencoderKLCogVideoX (absent package, no
checkpoint, no reference fixture), so these tests pin HIP == oracle and the structural rules (frame counts, causality, the
first-frame special cases), not the oracle itself.  Tolerance: bf16 activations through ~20 convolutions vs an fp32 oracle on
bf16-rounded weights: rel-L2 <= 3e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae as ovae  # noqa: E402  (checker only)

BF = torch.bfloat16
TINY = dict(block_out_channels=(32, 64, 64, 128), layers_per_block=1)


def rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / ref.norm()).item()


def make(cfg, seed=0):
    from orv_amd.vae import AutoencoderKLCogVideoX
    torch.manual_seed(seed)
    ref = ovae.AutoencoderKLCogVideoX(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in ref.named_parameters():
        if p.ndim > 1:
            p.data.copy_(torch.randn(p.shape, generator=g) * (1.5 / (p[0].numel() ** 0.5)))
        elif "norm" in n and n.endswith("weight"):
            p.data.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
        else:
            p.data.copy_(0.1 * torch.randn(p.shape, generator=g))
        p.data.copy_(p.data.to(BF).float())
    m = AutoencoderKLCogVideoX(**cfg)
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref.eval(), m.to("cuda:0", BF).eval()


@pytest.mark.parametrize("frames", [1, 2, 3, 5])
def test_decode_matches_oracle(frames):
    ref, m = make(TINY)
    z = torch.randn(2, 16, frames, 4, 6, generator=torch.Generator().manual_seed(frames)).to(BF).float()
    with torch.no_grad():
        want = ref.decode(z)
    got = m.decode(z.to("cuda:0", BF)).sample
    t_out = 1 + 4 * (frames - 1) if frames % 2 == 1 else 4 * frames        # even clips have no "first frame apart" branch
    assert got.shape == want.shape == (2, 3, t_out, 32, 48) and got.dtype == BF
    assert rel_l2(got, want) <= 3e-2


@pytest.mark.parametrize("frames", [1, 5, 9, 17])
def test_encode_matches_oracle(frames):
    ref, m = make(TINY, seed=3)
    x = (torch.rand(2, 3, frames, 32, 48, generator=torch.Generator().manual_seed(frames)) * 2 - 1).to(BF).float()
    with torch.no_grad():
        want = ref.encode(x)
    dist = m.encode(x.to("cuda:0", BF)).latent_dist
    assert dist.mean.shape == want.mean.shape == (2, 16, 1 + (frames - 1) // 4, 4, 6)
    assert rel_l2(dist.parameters, want.parameters) <= 3e-2
    # sample(generator) draws on the CPU generator in the parameter dtype, like randn_tensor
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    s = dist.sample(g1)
    eps = torch.randn(dist.mean.shape, generator=g2, dtype=BF).to("cuda:0")
    assert torch.equal(s, dist.mean + dist.std * eps)


def test_decode_frame_batching_rules():
    """diffusers decodes in batches of 2 latent frames (first batch takes the remainder: 5 = 3 + 2) with every causal
    convolution's conv_cache carried over and GroupNorm computed per batch.  So: a change confined to the second batch's latent
    frames leaves the first batch's 9 output frames bit-identical (nothing looks ahead, statistics are per batch); a change in the
    first batch reaches the second through the conv caches; and runs are bit-reproducible (no atomics in the statistics)."""
    ref, m = make(TINY, seed=7)
    assert m.frame_batches(5, 2) == [(0, 3), (3, 5)] and m.frame_batches(17, 8) == [(0, 9), (9, 17)] and m.frame_batches(1, 2) == [(0, 1)]
    z = torch.randn(1, 16, 5, 4, 6, generator=torch.Generator().manual_seed(0)).to("cuda:0", BF)
    base = m.decode(z).sample
    assert base.shape == (1, 3, 17, 32, 48) and torch.equal(base, m.decode(z).sample)
    z2 = z.clone()
    z2[:, :, 3:] += 1.0
    out2 = m.decode(z2).sample
    assert torch.equal(out2[:, :, :9], base[:, :, :9]) and not torch.equal(out2[:, :, 9:], base[:, :, 9:])
    z3 = z.clone()
    z3[:, :, 2] -= 1.0                                    # last frame of the first batch: reaches the second batch via conv_cache
    out3 = m.decode(z3).sample
    assert not torch.equal(out3[:, :, 9:], base[:, :, 9:])
    # the first batch alone decodes to the same 9 frames
    assert torch.equal(m.decode(z[:, :, :3]).sample, base[:, :, :9])


def test_state_dict_roundtrip_and_surface(tmp_path):
    from orv_amd.vae import AutoencoderKLCogVideoX
    ref, m = make(TINY, seed=9)
    m.save_pretrained(str(tmp_path / "vae"))
    m2 = AutoencoderKLCogVideoX.from_pretrained(str(tmp_path), subfolder="vae", torch_dtype=BF).to("cuda:0")
    assert m2.config.block_out_channels == (32, 64, 64, 128) and m2.config.scaling_factor == 1.15258426
    z = torch.randn(1, 16, 2, 4, 6).to("cuda:0", BF)
    assert torch.equal(m2.decode(z).sample, m.decode(z).sample)
    m2.enable_slicing(), m2.enable_tiling()
    with pytest.raises(RuntimeError, match="MI355X only"):
        m2.decode(z.cpu())


def test_full_size_decode_shape_and_pipeline_handoff():
    """The real 2B VAE geometry (128/256/256/512, 3 layers per block, 215.6 M parameters) on one 17-frame 320x480 clip: output
    shape, finiteness, and the pipeline's decode_latents -> postprocess_video('pil') path."""
    from orv_amd.vae import AutoencoderKLCogVideoX
    from orv_amd.components import VideoProcessor
    torch.manual_seed(0)
    m = AutoencoderKLCogVideoX()
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 215.58) < 0.01
    for p in m.parameters():
        if p.ndim > 1:
            p.data.normal_(0, 1.0 / (p[0].numel() ** 0.5))
    m = m.to("cuda:0", BF).eval()
    z = torch.randn(1, 16, 5, 40, 60, device="cuda:0", dtype=BF)
    torch.cuda.synchronize()
    import time
    m.decode(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.decode(z).sample
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"full-size VAE decode [1,16,5,40,60] -> {tuple(out.shape)}: {dt * 1e3:.1f} ms, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert out.shape == (1, 3, 17, 320, 480) and torch.isfinite(out.float()).all()
    frames = VideoProcessor().postprocess_video(out, output_type="pil")
    assert len(frames[0]) == 17 and frames[0][0].size == (480, 320)


def test_pipeline_end_to_end_with_the_mi355x_vae(tmp_path):
    """RGB frame in -> PIL frames out through orv_amd only: vae.encode(reference frame).latent_dist.sample(generator) in
    prepare_latents (:1150-1167), the DPM loop, vae.decode in decode_latents (:1476-1479), postprocess_video.  The pipeline
    directory is written by save_pretrained (vae/ included) and read back by from_pretrained."""
    import numpy as np
    import PIL.Image
    from conftest import load_golden
    from orv_amd import schedulers
    from orv_amd.cogvideox_control import CogVideoXImageToVideoPipelineTraj, CogVideoXTransformer3DModelTraj
    from orv_amd.vae import AutoencoderKLCogVideoX
    cfg, _, ins, w, _ = load_golden("pipe_ddim")
    tr = CogVideoXTransformer3DModelTraj(**cfg)
    tr.load_state_dict(w)
    ref, vae = make(TINY, seed=11)
    sched = schedulers.CogVideoXDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                             beta_schedule="scaled_linear", prediction_type="v_prediction",
                                             rescale_betas_zero_snr=True, snr_shift_scale=3.0, timestep_spacing="trailing")
    CogVideoXImageToVideoPipelineTraj(transformer=tr, scheduler=sched, vae=vae).save_pretrained(str(tmp_path / "pipe"))
    pipe = CogVideoXImageToVideoPipelineTraj.from_pretrained(str(tmp_path / "pipe"), torch_dtype=BF)
    assert isinstance(pipe.vae, AutoencoderKLCogVideoX) and isinstance(pipe.scheduler, schedulers.CogVideoXDPMScheduler)
    pipe.to("cuda", dtype=BF)
    pipe.vae.enable_slicing(), pipe.vae.enable_tiling()
    pipe.transformer.action_embed.forced_mask = torch.zeros(1, dtype=torch.bool)
    image = PIL.Image.fromarray(np.random.default_rng(2).integers(0, 255, (64, 96, 3), dtype=np.uint8))
    args = dict(image=image, prompt_embeds=ins["prompt_embeds"][:1].to("cuda", BF), height=64, width=96, num_frames=9,
                num_inference_steps=3, guidance_scale=1.0, controls_or_guidances={"actions": ins["actions"][:1].to("cuda", BF)})
    video = pipe(**args, generator=torch.Generator().manual_seed(3), output_type="pil").frames[0]
    assert len(video) == 9 and video[0].size == (96, 64)
    again = pipe(**args, generator=torch.Generator().manual_seed(3), output_type="pil").frames[0]
    assert all(np.array_equal(np.array(a), np.array(b)) for a, b in zip(video, again))           # bit-reproducible end to end
    # the decoded video is what the oracle VAE makes of the pipeline's latents
    lat = pipe(**args, generator=torch.Generator().manual_seed(3), output_type="latent").frames
    with torch.no_grad():
        want = ref.decode((lat.float().cpu().permute(0, 2, 1, 3, 4) / 1.15258426))
    got = pipe(**args, generator=torch.Generator().manual_seed(3), output_type="pt").frames          # [B, F, C, H, W] in [0, 1]
    want01 = (want / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)
    assert rel_l2(got, want01) <= 3e-2


@pytest.mark.parametrize("B,T,H,W,Cin,Cout,kt,t_shift,residual", [
    (2, 3, 5, 7, 64, 128, 3, 0, False),      # tiles cross image rows, frames and batch elements; replicate-first-frame padding
    (1, 4, 9, 2, 128, 128, 3, 2, True),      # W = 2: every voxel is an x edge; conv_cache frames in front (t_shift); residual epilogue
    (1, 2, 20, 33, 64, 256, 1, 0, False),    # per-frame 3x3 (kt = 1), 256-wide tile, M = 1320 (ragged last tile)
    (3, 1, 16, 16, 192, 128, 3, 0, True),    # C = 3 channel blocks per tap, single frame
    (1, 1, 320, 322, 64, 128, 3, 0, True),   # M = 103040 >= 384 x 256: the 384-row strip tile of the 128-wide convolutions
    (1, 2, 128, 130, 64, 256, 3, 1, False),  # M = 33280: 130 tiles of 256 x 256 (small shapes above run the 128 x 128 tile)
    (1, 2, 128, 130, 64, 128, 3, 0, True),   # 256 x 128 tile
])
def test_conv_kernels_against_torch_conv3d(B, T, H, W, Cin, Cout, kt, t_shift, residual):
    """orv_conv_gemm_bf16 (strip kernel for the stride-1 3x3 taps) against torch.nn.functional.conv3d in fp32 on the same
    bf16-rounded operands: causal temporal context (copies of frame 0, or the t_shift leading frames), zero spatial padding,
    weight column = ((dt * 3 + dy) * 3 + dx) * Cin + ci.  Tolerance: bf16 output rounding + fp32 accumulation order."""
    from orv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + W)
    Ts = T + t_shift
    x = torch.randn(B, Ts, H, W, Cin, generator=g).to(BF)                       # channels-last source, leading context frames first
    w = (torch.randn(Cout, Cin, kt, 3, 3, generator=g) / (Cin * kt * 9) ** 0.5).to(BF)
    bias = torch.randn(Cout, generator=g).to(BF)
    res = torch.randn(B * T * H * W, Cout, generator=g).to(BF) if residual else None
    # reference: frames [t_shift - (kt-1) + t .. t_shift + t] with negative indices clamped to frame 0
    xf = x.float().permute(0, 4, 1, 2, 3)                                       # [B, C, Ts, H, W]
    pad_t = (kt - 1) - t_shift
    if pad_t > 0:
        xf = torch.cat([xf[:, :, :1]] * pad_t + [xf], dim=2)
    ref = torch.nn.functional.conv3d(xf, w.float(), bias.float(), padding=(0, 1, 1))   # [B, Cout, T, H, W]
    ref = ref.permute(0, 2, 3, 4, 1).reshape(B * T * H * W, Cout)
    if residual:
        ref = ref + res.float()
    wp = w.permute(0, 2, 3, 4, 1).reshape(Cout, kt * 9 * Cin).contiguous()      # column = tap * Cin + ci
    out = torch.full((B * T * H * W, Cout), float("nan"), dtype=BF, device=dev)
    ops.conv_gemm(x.to(dev), wp.to(dev), bias.to(dev), out, B, Ts, H, W, Cin, T, H, W, kt, 3, 3, 1, 1, 0, 0, t_shift, Cout,
                  R=None if res is None else res.to(dev), ldr=Cout)
    err = (out.float().cpu() - ref).abs()
    assert bool((err <= 1.6e-2 * ref.abs() + 2e-2).all()), err.max().item()


# ---- tiling (what the reference's entry points switch on: inference_control_to_video.py:98-99) -------------------------------
TILED = dict(TINY, sample_height=96, sample_width=160)     # tile 48 x 80 px = 6 x 10 latent, stride 5 x 8, blend 8 x 16 px


def test_blend_kernel_matches_the_formula():
    """orv_vae_blend against b[y] = a[-e + y] (1 - y / e) + b[y] (y / e) (diffusers blend_v / blend_h) in fp32, one bf16 rounding."""
    from orv_amd import ops
    g = torch.Generator().manual_seed(2)
    for horizontal, (ha, wa, hb, wb), e in [(False, (7, 9, 5, 9), 4), (True, (6, 11, 6, 4), 3), (False, (3, 5, 8, 5), 3)]:
        a = torch.randn(2, 3, ha, wa, 8, generator=g).to(BF)
        b = torch.randn(2, 3, hb, wb, 8, generator=g).to(BF)
        want = b.float().clone()
        for t in range(e):
            w = t / e
            if horizontal:
                want[:, :, :, t] = a.float()[:, :, :, wa - e + t] * (1 - w) + b.float()[:, :, :, t] * w
            else:
                want[:, :, t] = a.float()[:, :, ha - e + t] * (1 - w) + b.float()[:, :, t] * w
        got = ops.vae_blend(a.cuda(), b.cuda().clone(), e, horizontal).cpu()
        assert torch.allclose(got.float(), want.to(BF).float(), rtol=8e-3, atol=1e-6)
        untouched = got[:, :, :, e:] if horizontal else got[:, :, e:]
        assert torch.equal(untouched, b[:, :, :, e:] if horizontal else b[:, :, e:])


def test_tiled_decode_matches_tiled_oracle():
    """Latent 8 x 14 against a 6 x 10 tile: 4 tiles (6x10, 6x6, 3x10, 3x6), each decoded with its own GroupNorm statistics and
    conv caches, seams blended in place in raster order.  HIP tiled == oracle tiled, and tiled != untiled (it is arithmetic)."""
    ref, m = make(TILED, seed=5)
    z = torch.randn(2, 16, 3, 8, 14, generator=torch.Generator().manual_seed(9)).to(BF).float()
    ref.enable_tiling(); m.enable_tiling(); m.enable_slicing()
    with torch.no_grad():
        want = ref.decode(z)
        ref.disable_tiling()
        untiled = ref.decode(z)
    got = m.decode(z.to("cuda:0", BF)).sample
    assert got.shape == want.shape == untiled.shape == (2, 3, 9, 64, 112)
    assert rel_l2(got, want) <= 3e-2
    assert rel_l2(untiled, want) > 5e-2            # per-tile statistics: a different function
    m.disable_tiling()
    assert rel_l2(m.decode(z.to("cuda:0", BF)).sample, untiled) <= 3e-2


def test_tiles_on_side_streams_are_bit_identical_to_the_serial_loop(monkeypatch):
    """Round 4: the independent tiles of a tiled decode / encode run on their own HIP streams (forked behind the input, joined
    before the seam blends).  Same kernels on the same data: bit-identical to the serial tile loop (ORV_VAE_TILE_STREAMS=0), on a
    cold model (weights repacked before the fork) and on repeated calls."""
    ref, m = make(TILED, seed=6)
    m.enable_tiling()
    z = torch.randn(2, 16, 3, 8, 14, generator=torch.Generator().manual_seed(4)).to("cuda:0", BF)
    x = (torch.rand(1, 3, 5, 64, 112, generator=torch.Generator().manual_seed(5)) * 2 - 1).to("cuda:0", BF)
    par = [m.decode(z).sample, m.decode(z).sample]                      # first call: cold packed-weight cache
    enc = m.encode(x).latent_dist.parameters
    monkeypatch.setenv("ORV_VAE_TILE_STREAMS", "0")
    ser = m.decode(z).sample
    assert torch.equal(par[0], ser) and torch.equal(par[1], ser)
    assert torch.equal(enc, m.encode(x).latent_dist.parameters)
    torch.cuda.synchronize()


def test_tiled_encode_matches_tiled_oracle():
    ref, m = make(TILED, seed=6)
    x = (torch.rand(1, 3, 9, 64, 112, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(BF).float()
    ref.enable_tiling(); m.enable_tiling()
    with torch.no_grad():
        want = ref.encode(x)
    dist = m.encode(x.to("cuda:0", BF)).latent_dist
    assert dist.mean.shape == want.mean.shape == (1, 16, 3, 8, 14)
    assert rel_l2(dist.parameters, want.parameters) <= 3e-2


def test_tiling_leaves_small_inputs_on_the_untiled_path():
    """A latent that fits one latent tile (<= 6 x 10 here; <= 30 x 45 for the real config) is decoded untiled even with tiling
    enabled: bit-identical to the run with tiling disabled."""
    _, m = make(TILED, seed=7)
    z = torch.randn(1, 16, 3, 6, 10, generator=torch.Generator().manual_seed(1)).to("cuda:0", BF)
    plain = m.decode(z).sample
    m.enable_tiling()
    assert torch.equal(m.decode(z).sample, plain)
    real = type(m)()                               # default config: 480 x 720 -> tiles of 30 x 45 latent / 240 x 360 px
    real.enable_tiling()
    assert (real.tile_latent_min_height, real.tile_latent_min_width, real.tile_sample_min_height, real.tile_sample_min_width) == (30, 45, 240, 360)
