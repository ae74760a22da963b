"""CPU: hand-derived known answers for the building blocks of oracle/vae.py (the VAE oracle is PARITY UNPINNED - diffusers is
not reachable - so what CAN be pinned is that each block does what its docstring says, on inputs whose answer is computed by
hand below, independently of the block's own code path).  Each test states its derivation."""
import torch

from oracle import vae as ovae


def test_frame_batches_rule():
    # max(n // size, 1) batches, the first takes the remainder: 5 latent frames by 2 -> [0,3) [3,5); 17 sample frames by 8 ->
    # [0,9) [9,17); fewer frames than a batch -> one batch
    fb = ovae.AutoencoderKLCogVideoX.frame_batches
    assert fb(5, 2) == [(0, 3), (3, 5)]
    assert fb(13, 2) == [(0, 3), (3, 5), (5, 7), (7, 9), (9, 11), (11, 13)]
    assert fb(17, 8) == [(0, 9), (9, 17)]
    assert fb(1, 2) == [(0, 1)] and fb(1, 8) == [(0, 1)] and fb(8, 8) == [(0, 8)] and fb(2, 2) == [(0, 2)]


def test_causal_conv_impulse_response_and_first_frame_padding():
    # all-ones 3x3x3 kernel, no bias, input = 1 at (t=0, y=2, x=2) of a 4-frame 5x5 clip.  Temporal context of output frame t is
    # input frames {t-2, t-1, t} with frames < 0 replaced by COPIES OF FRAME 0: frame 0 is counted 3x at t=0, 2x at t=1, 1x at
    # t=2 and is out of reach at t=3; spatially the 3x3 window around (2,2) sees the impulse once.
    conv = ovae.CogVideoXCausalConv3d(1, 1, 3)
    with torch.no_grad():
        conv.conv.weight.fill_(1.0)
        conv.conv.bias.zero_()
    x = torch.zeros(1, 1, 4, 5, 5)
    x[0, 0, 0, 2, 2] = 1.0
    y = conv(x).detach()
    assert y.shape == x.shape
    for t, want in enumerate([3.0, 2.0, 1.0, 0.0]):
        blk = torch.zeros(5, 5)
        blk[1:4, 1:4] = want
        assert torch.equal(y[0, 0, t], blk), t
    # an impulse in the LAST frame never reaches an earlier one (causality)
    x = torch.zeros(1, 1, 4, 5, 5)
    x[0, 0, 3, 0, 0] = 1.0
    y = conv(x).detach()
    assert float(y[0, 0, :3].abs().sum()) == 0.0 and float(y[0, 0, 3, 0, 0]) == 1.0 and float(y[0, 0, 3, 2, 2]) == 0.0


def test_conv_cache_makes_chunked_convolution_equal_to_whole_clip():
    # the cache of a batch = its last two INPUT frames; the next batch uses them instead of copies of its own first frame, so
    # chunk-by-chunk == whole clip exactly (same additions in the same order per output voxel)
    torch.manual_seed(0)
    conv = ovae.CogVideoXCausalConv3d(3, 4, 3)
    x = torch.randn(2, 3, 7, 6, 5)
    whole = conv(x)
    outs, prev = [], None
    for a, b in [(0, 3), (3, 4), (4, 7)]:          # a one-frame batch in the middle: its cache is [last of batch 0, itself]
        cc = ovae.ConvCache(prev)
        outs.append(conv(x[:, :, a:b], cc))
        prev = cc.new
        assert torch.equal(cc.new[id(conv)], x[:, :, b - 2:b])
    assert torch.equal(torch.cat(outs, dim=2), whole)


def test_spatial_norm_first_frame_rule():
    # identity GroupNorm data (f == 0 -> norm(f) == beta == 0), conv_y weight 0 / bias 0, conv_b = identity on one channel:
    # out == resized zq.  f has 5 frames (odd, > 1): frame 0 comes from zq frame 0 ALONE, frames 1..4 from nearest resize of
    # zq frames 1..2 to 4 frames = [1, 1, 2, 2].  Spatially 2x2 -> 4x4 nearest = each value repeated 2x2.
    sn = ovae.CogVideoXSpatialNorm3D(2, 2, groups=1)
    with torch.no_grad():
        for c in (sn.conv_y, sn.conv_b):
            c.conv.weight.zero_()
            c.conv.bias.zero_()
        sn.conv_b.conv.weight[0, 0] = 1.0
        sn.conv_b.conv.weight[1, 1] = 1.0
    zq = torch.zeros(1, 2, 3, 2, 2)
    for t in range(3):
        zq[0, 0, t] = torch.tensor([[10.0 * t + 1, 10.0 * t + 2], [10.0 * t + 3, 10.0 * t + 4]])
    out = sn(torch.zeros(1, 2, 5, 4, 4), zq)
    for t, src in enumerate([0, 1, 1, 2, 2]):
        want = zq[0, 0, src].repeat_interleave(2, 0).repeat_interleave(2, 1)
        assert torch.equal(out[0, 0, t], want), t
    # even frame count: plain nearest over (T, H, W): 4 frames from 2 -> [0, 0, 1, 1]
    out = sn(torch.zeros(1, 2, 4, 4, 4), zq[:, :, :2])
    for t, src in enumerate([0, 0, 1, 1]):
        assert torch.equal(out[0, 0, t], zq[0, 0, src].repeat_interleave(2, 0).repeat_interleave(2, 1)), t


def _identity_conv2d(conv):
    with torch.no_grad():
        conv.weight.zero_()
        conv.bias.zero_()
        for c in range(conv.weight.shape[0]):
            conv.weight[c, c, 1, 1] = 1.0


def test_upsample_frame_counts_and_duplication():
    # compress_time: odd clip (T=3): frame 0 only resized spatially, frames 1..2 doubled -> [0, 1, 1, 2, 2] (1 + 2 (T-1) = 5);
    # even clip (T=2): all doubled -> [0, 0, 1, 1]; single frame stays single; without compress_time T is unchanged.
    up = ovae.CogVideoXUpsample3D(1, 1, compress_time=True)
    _identity_conv2d(up.conv)
    x = torch.arange(3.0).view(1, 1, 3, 1, 1).expand(1, 1, 3, 2, 2).clone()
    y = up(x)
    assert y.shape == (1, 1, 5, 4, 4) and [float(v) for v in y[0, 0, :, 1, 1]] == [0.0, 1.0, 1.0, 2.0, 2.0]
    y = up(x[:, :, :2])
    assert y.shape == (1, 1, 4, 4, 4) and [float(v) for v in y[0, 0, :, 0, 0]] == [0.0, 0.0, 1.0, 1.0]
    assert up(x[:, :, :1]).shape == (1, 1, 1, 4, 4)
    up2 = ovae.CogVideoXUpsample3D(1, 1, compress_time=False)
    _identity_conv2d(up2.conv)
    y = up2(x)
    assert y.shape == (1, 1, 3, 4, 4) and [float(v) for v in y[0, 0, :, 3, 3]] == [0.0, 1.0, 2.0]


def test_downsample_time_pooling_and_stride2_padding():
    # compress_time on 5 frames [0, 2, 4, 6, 8]: first kept, the rest averaged in pairs -> [0, 3, 7]; 4 frames [0, 2, 4, 6] -> [1, 5].
    # Spatial: pad (right, bottom) by one zero, 3x3 stride 2 no padding: 4x4 -> floor((5 - 3) / 2) + 1 = 2; with a centre-tap
    # identity kernel output (i, j) = input (2 i + 1, 2 j + 1).
    dn = ovae.CogVideoXDownsample3D(1, 1, compress_time=True)
    _identity_conv2d(dn.conv)
    ramp = torch.arange(16.0).view(4, 4)
    x = (torch.arange(5.0) * 2).view(1, 1, 5, 1, 1) + ramp.view(1, 1, 1, 4, 4) * 100
    y = dn(x)
    assert y.shape == (1, 1, 3, 2, 2)
    for t, tv in enumerate([0.0, 3.0, 7.0]):
        want = torch.tensor([[ramp[1, 1], ramp[1, 3]], [ramp[3, 1], ramp[3, 3]]]) * 100 + tv
        assert torch.equal(y[0, 0, t], want), t
    y = dn(x[:, :, :4])
    assert y.shape == (1, 1, 2, 2, 2) and [float(v) for v in (y[0, 0, :, 0, 0] - 500.0)] == [1.0, 5.0]


def test_decoder_and_encoder_frame_arithmetic():
    # temporal_compression_ratio 4 = two compress_time stages: T latent frames decode to 1 + 4 (T - 1) frames per BATCH rule:
    # batch of 3 latent frames (first batch of 5) -> 3 -> 5 -> 9 frames, batch of 2 -> 2 -> 4 -> 8 frames: 17 in total;
    # encode 17 frames: batch of 9 -> 5 -> 3, batch of 8 -> 4 -> 2: 5 latent frames; spatial factor 8.
    torch.manual_seed(0)
    vae = ovae.AutoencoderKLCogVideoX(block_out_channels=(8, 8, 8, 8), layers_per_block=1, norm_num_groups=2).eval()
    with torch.no_grad():
        for T, F_ in [(1, 1), (2, 8), (3, 9), (5, 17)]:       # an EVEN batch has no first-frame exception: 2 -> 4 -> 8
            assert vae.decode(torch.randn(1, 16, T, 2, 3)).shape == (1, 3, F_, 16, 24), T
        for F_, T in [(1, 1), (9, 3), (17, 5)]:
            d = vae.encode(torch.randn(1, 3, F_, 16, 24))
            assert d.mean.shape == (1, 16, T, 2, 3), F_


def test_diagonal_gaussian_known_answer():
    # parameters [mean | logvar] on dim 1; logvar clamped to [-30, 20]; std = exp(0.5 logvar).  mean 2, logvar ln 4 -> std 2:
    # sample = 2 + 2 eps; logvar 100 -> clamped 20 -> std e^10; mode == mean.
    import math
    p = torch.zeros(1, 4, 1, 1, 1)
    p[0, 0], p[0, 1] = 2.0, -1.0
    p[0, 2], p[0, 3] = math.log(4.0), 100.0
    d = ovae.DiagonalGaussianDistribution(p)
    assert torch.allclose(d.std.flatten(), torch.tensor([2.0, math.exp(10.0)]))
    assert torch.equal(d.mode().flatten(), torch.tensor([2.0, -1.0]))
    g = torch.Generator().manual_seed(3)
    eps = torch.randn(d.mean.shape, generator=torch.Generator().manual_seed(3))
    assert torch.allclose(d.sample(g), d.mean + d.std * eps)


def test_tile_geometry_and_blend_known_answers():
    """Tiling arithmetic of the 480 x 720 VAE config, by hand: sample tile 240 x 360, latent tile 30 x 45, strides
    int(30 * 5/6) = 25 and int(45 * 4/5) = 36, sample-space blend extents 40 / 72, crop 200 x 288: a 40 x 60 latent is the four
    tiles (30|15) x (45|24) and reassembles to 320 x 480.  Seam blend: constant tiles a = 1, b = 3 over an extent of 4 give
    1, 1.5, 2, 2.5 on b's first four rows (weights y / 4), the rest of b untouched; tiles are blended in place in raster order,
    so the tile BELOW sees the already-blended tile above."""
    import torch
    from oracle.vae import AutoencoderKLCogVideoX
    v = AutoencoderKLCogVideoX(block_out_channels=(8, 8, 8, 8), layers_per_block=1, norm_num_groups=4)
    v.enable_tiling()
    assert (v.tile_sample_min_height, v.tile_sample_min_width, v.tile_latent_min_height, v.tile_latent_min_width) == (240, 360, 30, 45)
    assert [int(30 * (1 - v.tile_overlap_factor_height)), int(45 * (1 - v.tile_overlap_factor_width))] == [25, 36]
    assert [int(240 * v.tile_overlap_factor_height), int(360 * v.tile_overlap_factor_width)] == [40, 72]
    seen = []

    def fake_decoder(z, cc):                      # 8 x nearest upsampling stands in for the decoder: records the tile sizes
        seen.append(tuple(z.shape[-2:]))
        return z[:, :3].repeat_interleave(8, -2).repeat_interleave(8, -1)
    z = torch.arange(40 * 60, dtype=torch.float32).reshape(1, 1, 1, 40, 60).expand(1, 16, 1, 40, 60).contiguous()
    out = v._tiled(fake_decoder, z, 2, 30, 45, 40, 72, 200, 288)
    assert seen == [(30, 45), (30, 24), (15, 45), (15, 24)] and out.shape == (1, 3, 1, 320, 480)
    # every tile shows the same underlying image, so blending tiles of it must reproduce it (to fp32 rounding of v (1 - w) + v w)
    assert torch.allclose(out, z[:, :3].repeat_interleave(8, -2).repeat_interleave(8, -1), rtol=1e-6, atol=0)
    a, b = torch.ones(1, 1, 1, 6, 2), torch.full((1, 1, 1, 5, 2), 3.0)
    r = AutoencoderKLCogVideoX.blend_v(a, b, 4)
    assert r is b and torch.equal(b[0, 0, 0, :, 0], torch.tensor([1.0, 1.5, 2.0, 2.5, 3.0]))
    a, b = torch.ones(1, 1, 1, 2, 3), torch.full((1, 1, 1, 2, 7), 5.0)
    AutoencoderKLCogVideoX.blend_h(a, b, 4)        # extent clipped to min(3, 7, 4) = 3: weights 0, 1/3, 2/3
    assert torch.allclose(b[0, 0, 0, 0], torch.tensor([1.0, 1 + 4 / 3, 1 + 8 / 3, 5.0, 5.0, 5.0, 5.0]))
