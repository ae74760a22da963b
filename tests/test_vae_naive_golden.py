"""The VAE decode held to an INDEPENDENT second derivation: tests/golden/vae_tiny_naive.safetensors is written by
tools/make_vae_naive.py - plain numpy with explicit loops over taps / frames / groups / tiles, sharing no code with oracle/vae.py
or orv_amd/vae.py.  CPU: the oracle reproduces it (untiled odd clip with two frame batches and conv caches, even clip, tiled
decode with seam blends).  GPU: the HIP path reproduces it.  Still NOT a pin against diffusers (both derivations come from the same
memory of its published algorithm; /root/reference holds no VAE vector): it removes transcription errors, and says so."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_tiny_naive.safetensors")


def _load():
    with safe_open(GOLD, framework="pt") as f:
        meta = f.metadata()
        t = {k: f.get_tensor(k) for k in f.keys()}
    cfg = json.loads(meta["config"])
    shapes = [(n, tuple(s)) for n, s in json.loads(meta["param_shapes"])]
    return cfg, shapes, t


def _to_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _hash_uniform(n, salt):
    # the fixture's weight rule (tools/make_vae_naive.py docstring), restated here: the probe values below hold it to the script's
    i = np.arange(n, dtype=np.uint64)
    x = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    x ^= (x >> np.uint64(15))
    x = (x * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    x ^= (x >> np.uint64(13))
    return (x.astype(np.float64) / 4294967296.0 - 0.5).astype(np.float32)


def _decoder_weights(shapes):
    w = {}
    for salt, (name, shape) in enumerate(shapes):
        u = _hash_uniform(int(np.prod(shape)), salt + 1)
        if len(shape) > 1:
            v = u * (2.0 * 1.5 * np.sqrt(3.0) / np.sqrt(int(np.prod(shape[1:]))))
        elif "norm_layer.weight" in name:
            v = 1.0 + 0.4 * u
        else:
            v = 0.3 * u
        w["decoder." + name] = torch.from_numpy(_to_bf16(v).reshape(shape).copy())
    return w


def _model_kwargs(cfg):
    return dict(block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                latent_channels=cfg["latent_channels"], out_channels=cfg["out_channels"], norm_num_groups=cfg["norm_num_groups"],
                temporal_compression_ratio=cfg["temporal_compression_ratio"], sample_height=cfg["sample_height"],
                sample_width=cfg["sample_width"])


def _rel_l2(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).norm() / ref.norm()).item()


def test_fixture_weight_rule_and_key_set_match_the_oracle_decoder():
    from oracle import vae as ovae
    cfg, shapes, t = _load()
    w = _decoder_weights(shapes)
    probe = torch.cat([w["decoder.conv_in.conv.weight"].reshape(-1)[:8], w["decoder.norm_out.norm_layer.weight"].reshape(-1)[:4],
                       w["decoder.conv_out.conv.bias"].reshape(-1)])
    assert torch.equal(probe, t["weight_probe"])
    ref = ovae.AutoencoderKLCogVideoX(**_model_kwargs(cfg))
    dec = {k: tuple(v.shape) for k, v in ref.state_dict().items() if k.startswith("decoder.")}
    assert dec == {"decoder." + n: s for n, s in shapes}          # the two derivations agree on every parameter name and shape


def test_oracle_decode_reproduces_the_naive_derivation():
    from oracle import vae as ovae
    cfg, shapes, t = _load()
    ref = ovae.AutoencoderKLCogVideoX(**_model_kwargs(cfg)).double().eval()
    missing, unexpected = ref.load_state_dict({k: v.double() for k, v in _decoder_weights(shapes).items()}, strict=False)
    assert not unexpected and all(k.startswith("encoder.") for k in missing)
    with torch.no_grad():
        for zk, sk in (("z_small", "sample_small"), ("z_even", "sample_even")):
            got = ref.decode(t[zk][None].double())[0]
            assert got.shape == t[sk].shape
            assert _rel_l2(got, t[sk]) <= 1e-5, (zk, _rel_l2(got, t[sk]))
        # the same latent through the untiled path and, with enable_tiling(), through four blended tiles
        z = t["z_tiled"][None].double()
        assert _rel_l2(ref.decode(z)[0], t["sample_tiled_untiled_path"]) <= 1e-5
        ref.enable_tiling()
        got = ref.decode(z)[0]
        assert got.shape == t["sample_tiled"].shape == (3, 9, 64, 96)
        assert _rel_l2(got, t["sample_tiled"]) <= 1e-5
        assert _rel_l2(t["sample_tiled"], t["sample_tiled_untiled_path"]) > 1e-2      # tiling is arithmetic, not a memory knob


@pytest.mark.gpu
def test_hip_decode_reproduces_the_naive_derivation():
    from orv_amd.vae import AutoencoderKLCogVideoX
    cfg, shapes, t = _load()
    m = AutoencoderKLCogVideoX(**_model_kwargs(cfg))
    missing, unexpected = m.load_state_dict(_decoder_weights(shapes), strict=False)
    assert not unexpected and all(k.startswith("encoder.") for k in missing)
    m = m.to("cuda:0", torch.bfloat16).eval()
    for zk, sk in (("z_small", "sample_small"), ("z_even", "sample_even")):
        got = m.decode(t[zk][None].to("cuda:0", torch.bfloat16)).sample[0]
        assert got.shape == t[sk].shape
        assert _rel_l2(got, t[sk]) <= 3e-2, (zk, _rel_l2(got, t[sk]))      # bf16 activations through ~20 convolutions (tests/test_gpu_vae.py bound)
    z = t["z_tiled"][None].to("cuda:0", torch.bfloat16)
    assert _rel_l2(m.decode(z).sample[0], t["sample_tiled_untiled_path"]) <= 3e-2
    m.enable_tiling()
    got = m.decode(z).sample[0]
    assert got.shape == t["sample_tiled"].shape
    assert _rel_l2(got, t["sample_tiled"]) <= 3e-2
