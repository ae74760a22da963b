#!/usr/bin/env python
"""Independent second derivation of the CogVideoX VAE decode -> tests/golden/vae_tiny_naive.safetensors.

No reference vector for the VAE can exist here (diffusers is absent, /root/reference holds no VAE input / output pair), so
oracle/vae.py and the HIP path (orv_amd/vae.py) were only ever compared with each other - both written by one hand.  This
script restates the published algorithm of ``AutoencoderKLCogVideoX.decode`` (diffusers >= 0.32, the VAE the reference
calls at /root/reference/orv/models/cogvideox_control.py:1476-1479 with tiling switched on at
/root/reference/orv/pipeline/inference_control_to_video.py:98-99) a SECOND time, as plain numpy with explicit loops over
convolution taps, frames, groups and tiles: no torch, no import from oracle/ or orv_amd/, no shared helper.  A transcription
error in either derivation (a tap offset, a padding side, the first-frame rule, a blend weight) shows up as a mismatch; an
error in the author's memory of diffusers common to both does not - parity stays "unpinned against the reference" and the
fixture says so.

Weights are not stored (6 M numbers): both sides generate them from the integer hash ``hash_uniform`` below (rounded to bf16 so
that the bf16 HIP model holds exactly the same values).  Stored: the config, the latents, the decoded clips (untiled and tiled).

    python tools/make_vae_naive.py        # ~1 min of numpy
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(block_out_channels=(32, 64, 64, 128), layers_per_block=1, latent_channels=16, out_channels=3, norm_num_groups=32,
           temporal_compression_ratio=4, sample_height=96, sample_width=160)


# ---------------------------------------------------------------- weights: a documented integer hash, rounded to bf16
def to_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def hash_uniform(n, salt):
    """n numbers in [-0.5, 0.5): u_i = ((i * 2654435761 + salt * 40503 + 12345) mod 2^32) / 2^32 - 0.5, i = 0 .. n-1, with a
    xorshift of the 32-bit word so neighbouring indices decorrelate."""
    i = np.arange(n, dtype=np.uint64)
    x = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    x ^= (x >> np.uint64(15))
    x = (x * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    x ^= (x >> np.uint64(13))
    return (x.astype(np.float64) / 4294967296.0 - 0.5).astype(np.float32)


def decoder_param_shapes(cfg):
    """(diffusers key under ``decoder.``, shape) in a fixed order."""
    boc, L, zc = list(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["latent_channels"]
    rev = boc[::-1]
    out = []

    def conv3(name, co, ci, k):
        out.append((name + ".conv.weight", (co, ci, k, k, k)))
        out.append((name + ".conv.bias", (co,)))

    def snorm(name, c):
        out.append((name + ".norm_layer.weight", (c,)))
        out.append((name + ".norm_layer.bias", (c,)))
        conv3(name + ".conv_y", c, zc, 1)
        conv3(name + ".conv_b", c, zc, 1)

    def resnet(name, ci, co):
        snorm(name + ".norm1", ci)
        conv3(name + ".conv1", co, ci, 3)
        snorm(name + ".norm2", co)
        conv3(name + ".conv2", co, co, 3)
        if ci != co:
            out.append((name + ".conv_shortcut.weight", (co, ci, 1, 1, 1)))
            out.append((name + ".conv_shortcut.bias", (co,)))

    conv3("conv_in", rev[0], zc, 3)
    for j in range(2):
        resnet(f"mid_block.resnets.{j}", rev[0], rev[0])
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            resnet(f"up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i != len(rev) - 1:
            out.append((f"up_blocks.{i}.upsamplers.0.conv.weight", (c, c, 3, 3)))
            out.append((f"up_blocks.{i}.upsamplers.0.conv.bias", (c,)))
        prev = c
    snorm("norm_out", rev[-1])
    conv3("conv_out", cfg["out_channels"], rev[-1], 3)
    return out


def make_weights(cfg):
    w = {}
    for salt, (name, shape) in enumerate(decoder_param_shapes(cfg)):
        n = int(np.prod(shape))
        u = hash_uniform(n, salt + 1)
        if len(shape) > 1:
            fan_in = int(np.prod(shape[1:]))
            v = u * (2.0 * 1.5 * np.sqrt(3.0) / np.sqrt(fan_in))          # uniform with std 1.5 / sqrt(fan_in)
        elif "norm_layer.weight" in name:
            v = 1.0 + 0.4 * u
        else:
            v = 0.3 * u
        w[name] = to_bf16(v).reshape(shape)
    return w


# ---------------------------------------------------------------- the decoder, tensor layout [C, T, H, W] (one clip)
def silu(x):
    return x / (1.0 + np.exp(-x))


def nearest_index(n_out, n_in):
    """torch ``F.interpolate(mode='nearest')``: source index floor(dst * n_in / n_out)."""
    return np.minimum((np.arange(n_out) * (n_in / n_out)).astype(np.int64), n_in - 1)


def resize_nearest(x, size):
    """x [C, T, H, W] -> [C, *size] by nearest neighbour in all three axes."""
    t, h, w = size
    return x[:, nearest_index(t, x.shape[1])][:, :, nearest_index(h, x.shape[2])][:, :, :, nearest_index(w, x.shape[3])]


class Caches:
    """conv_cache of the frame-batched decode: per causal convolution (by parameter name) the last two input frames of the
    previous batch."""

    def __init__(self, prev=None):
        self.prev, self.new = prev or {}, {}


def causal_conv3d(x, w, name, cc):
    W, b = w[name + ".conv.weight"].astype(np.float64), w[name + ".conv.bias"].astype(np.float64)
    co, ci, kt, kh, kw = W.shape
    if kt > 1:
        front = cc.prev[name] if name in cc.prev else np.repeat(x[:, :1], kt - 1, axis=1)     # first frame repeated, or the cache
        x = np.concatenate([front, x], axis=1)
        cc.new[name] = x[:, -(kt - 1):].copy()
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((ci, x.shape[1], x.shape[2] + 2 * ph, x.shape[3] + 2 * pw))
    xp[:, :, ph:ph + x.shape[2], pw:pw + x.shape[3]] = x
    T, H, Wd = x.shape[1] - (kt - 1), x.shape[2], x.shape[3]
    out = np.zeros((co, T, H, Wd)) + b[:, None, None, None]
    for dt in range(kt):                       # explicit taps
        for dy in range(kh):
            for dx in range(kw):
                out += np.tensordot(W[:, :, dt, dy, dx], xp[:, dt:dt + T, dy:dy + H, dx:dx + Wd], axes=(1, 0))
    return out


def conv2d_same(x, W, b):
    """per-frame 3 x 3 convolution, zero padding 1: x [C, T, H, W]."""
    W, b = W.astype(np.float64), b.astype(np.float64)
    co, ci, kh, kw = W.shape
    xp = np.zeros((ci, x.shape[1], x.shape[2] + 2, x.shape[3] + 2))
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((co,) + x.shape[1:]) + b[:, None, None, None]
    for dy in range(kh):
        for dx in range(kw):
            out += np.tensordot(W[:, :, dy, dx], xp[:, :, dy:dy + x.shape[2], dx:dx + x.shape[3]], axes=(1, 0))
    return out


def group_norm(x, gamma, beta, groups, eps=1e-6):
    c = x.shape[0]
    per = c // groups
    out = np.empty_like(x)
    for g in range(groups):                    # statistics over the channels of the group and ALL frames / pixels of this batch
        blk = x[g * per:(g + 1) * per]
        m = blk.mean()
        v = ((blk - m) ** 2).mean()
        out[g * per:(g + 1) * per] = (blk - m) / np.sqrt(v + eps)
    return out * gamma.astype(np.float64)[:, None, None, None] + beta.astype(np.float64)[:, None, None, None]


def spatial_norm(f, zq, w, name, groups, cc):
    T = f.shape[1]
    if T > 1 and T % 2 == 1:                   # odd clip: the first frame of the latent is resized on its own
        z = np.concatenate([resize_nearest(zq[:, :1], (1,) + f.shape[2:]), resize_nearest(zq[:, 1:], (T - 1,) + f.shape[2:])], axis=1)
    else:
        z = resize_nearest(zq, f.shape[1:])
    n = group_norm(f, w[name + ".norm_layer.weight"], w[name + ".norm_layer.bias"], groups)
    return n * causal_conv3d(z, w, name + ".conv_y", cc) + causal_conv3d(z, w, name + ".conv_b", cc)


def resnet(x, zq, w, name, groups, cc):
    h = causal_conv3d(silu(spatial_norm(x, zq, w, name + ".norm1", groups, cc)), w, name + ".conv1", cc)
    h = causal_conv3d(silu(spatial_norm(h, zq, w, name + ".norm2", groups, cc)), w, name + ".conv2", cc)
    if name + ".conv_shortcut.weight" in w:
        Ws = w[name + ".conv_shortcut.weight"].astype(np.float64)[:, :, 0, 0, 0]
        x = np.tensordot(Ws, x, axes=(1, 0)) + w[name + ".conv_shortcut.bias"].astype(np.float64)[:, None, None, None]
    return h + x


def upsample(x, w, name, compress_time):
    C, T, H, Wd = x.shape
    if compress_time and T > 1 and T % 2 == 1:     # first frame: space only; the rest: time and space doubled
        first = resize_nearest(x[:, :1], (1, 2 * H, 2 * Wd))
        rest = resize_nearest(x[:, 1:], (2 * (T - 1), 2 * H, 2 * Wd))
        x = np.concatenate([first, rest], axis=1)
    elif compress_time and T > 1:
        x = resize_nearest(x, (2 * T, 2 * H, 2 * Wd))
    else:                                          # one frame, or a block that keeps the frame count: space only
        x = resize_nearest(x, (T, 2 * H, 2 * Wd))
    return conv2d_same(x, w[name + ".conv.weight"], w[name + ".conv.bias"])


def decoder(z, w, cfg, cc):
    rev = list(cfg["block_out_channels"])[::-1]
    G, L = cfg["norm_num_groups"], cfg["layers_per_block"]
    levels = int(np.log2(cfg["temporal_compression_ratio"]))
    h = causal_conv3d(z, w, "conv_in", cc)
    for j in range(2):
        h = resnet(h, z, w, f"mid_block.resnets.{j}", G, cc)
    for i in range(len(rev)):
        for j in range(L + 1):
            h = resnet(h, z, w, f"up_blocks.{i}.resnets.{j}", G, cc)
        if i != len(rev) - 1:
            h = upsample(h, w, f"up_blocks.{i}.upsamplers.0", compress_time=i < levels)
    h = silu(spatial_norm(h, z, w, "norm_out", G, cc))
    return causal_conv3d(h, w, "conv_out", cc)


def frame_batches(n, size):
    nb, rem = max(n // size, 1), n % size
    return [(size * i + (0 if i == 0 else rem), min(size * (i + 1) + rem, n)) for i in range(nb)]


def decode_untiled(z, w, cfg):
    """batches of 2 latent frames (the first one takes the remainder), conv caches carried from batch to batch."""
    outs, prev = [], None
    for a, b in frame_batches(z.shape[1], 2):
        cc = Caches(prev)
        outs.append(decoder(z[:, a:b], w, cfg, cc))
        prev = cc.new
    return np.concatenate(outs, axis=1)


def decode_tiled(z, w, cfg):
    """enable_tiling(): tiles of (sample_height / 2, sample_width / 2) pixels = that / 8 latent rows / columns, overlap 1/6 and
    1/5; every tile decoded on its own, then blended in raster order with the finished tile above and to the left, cropped, joined."""
    down = 2 ** (len(cfg["block_out_channels"]) - 1)
    ts_h, ts_w = cfg["sample_height"] // 2, cfg["sample_width"] // 2
    tl_h, tl_w = int(ts_h / down), int(ts_w / down)
    ov_h, ov_w = 1 / 6, 1 / 5
    if z.shape[2] <= tl_h and z.shape[3] <= tl_w:
        return decode_untiled(z, w, cfg)
    st_h, st_w = int(tl_h * (1 - ov_h)), int(tl_w * (1 - ov_w))
    be_h, be_w = int(ts_h * ov_h), int(ts_w * ov_w)
    lim_h, lim_w = ts_h - be_h, ts_w - be_w
    rows = []
    for i in range(0, z.shape[2], st_h):
        rows.append([decode_untiled(z[:, :, i:i + tl_h, j:j + tl_w], w, cfg) for j in range(0, z.shape[3], st_w)])
    out_rows = []
    for i, row in enumerate(rows):
        out_row = []
        for j, tile in enumerate(row):
            if i > 0:                              # rows 0 .. e-1 of this tile fade in from the last e rows of the tile above
                a = rows[i - 1][j]
                e = min(a.shape[2], tile.shape[2], be_h)
                for y in range(e):
                    tile[:, :, y, :] = a[:, :, -e + y, :] * (1 - y / e) + tile[:, :, y, :] * (y / e)
            if j > 0:
                a = row[j - 1]
                e = min(a.shape[3], tile.shape[3], be_w)
                for x in range(e):
                    tile[:, :, :, x] = a[:, :, :, -e + x] * (1 - x / e) + tile[:, :, :, x] * (x / e)
            out_row.append(tile[:, :, :lim_h, :lim_w])
        out_rows.append(np.concatenate(out_row, axis=3))
    return np.concatenate(out_rows, axis=2)


def main():
    from safetensors.numpy import save_file
    w = make_weights(CFG)
    zc = CFG["latent_channels"]
    z_small = to_bf16(4.0 * hash_uniform(zc * 5 * 4 * 6, 1001)).reshape(zc, 5, 4, 6)        # 5 latent frames: batches 3 + 2
    z_even = to_bf16(4.0 * hash_uniform(zc * 2 * 4 * 6, 1002)).reshape(zc, 2, 4, 6)         # one batch, even: no first-frame rule
    z_tiled = to_bf16(4.0 * hash_uniform(zc * 3 * 8 * 12, 1003)).reshape(zc, 3, 8, 12)      # 8 x 12 latent > 6 x 10 tile: four tiles (6|3 rows x 10|4 columns)
    out = {
        "z_small": z_small, "sample_small": decode_untiled(z_small.astype(np.float64), w, CFG).astype(np.float32),
        "z_even": z_even, "sample_even": decode_untiled(z_even.astype(np.float64), w, CFG).astype(np.float32),
        "z_tiled": z_tiled, "sample_tiled": decode_tiled(z_tiled.astype(np.float64), w, CFG).astype(np.float32),
        "sample_tiled_untiled_path": decode_untiled(z_tiled.astype(np.float64), w, CFG).astype(np.float32),
        # spot values of the weight hash, so a test can hold its own re-implementation of it to the script's
        "weight_probe": np.concatenate([w["conv_in.conv.weight"].ravel()[:8], w["norm_out.norm_layer.weight"].ravel()[:4],
                                        w["conv_out.conv.bias"].ravel()]).astype(np.float32),
    }
    meta = {"config": json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in CFG.items()}),
            "param_shapes": json.dumps([[n, list(s)] for n, s in decoder_param_shapes(CFG)]),
            "generator": "tools/make_vae_naive.py (numpy, explicit loops; independent of oracle/vae.py and orv_amd/vae.py)",
            "parity": "second derivation by the same author: removes transcription errors, NOT pinned against diffusers"}
    path = os.path.join(ROOT, "tests", "golden", "vae_tiny_naive.safetensors")
    save_file({k: np.ascontiguousarray(v) for k, v in sorted(out.items())}, path, metadata=meta)
    # the metadata entries come out of safetensors' hash map in a per-process order: sort them in place (same bytes, same length) so that
    # a regeneration is byte-identical
    import struct
    with open(path, "r+b") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        hdr = json.loads(f.read(n))
        hdr["__metadata__"] = dict(sorted(hdr["__metadata__"].items()))
        raw = json.dumps(hdr, separators=(",", ":"), ensure_ascii=False).encode()
        assert len(raw) <= n
        f.seek(8)
        f.write(raw + b" " * (n - len(raw)))
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
