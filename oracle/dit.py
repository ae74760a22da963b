"""CPU oracle: functional fp32 restatement of ORV's 3-D DiT forward and its action conditioning.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It is a restatement, op for op and in the
reference's order, of

* ``CogVideoXTransformer3DModelTraj.forward``      /root/reference/orv/models/cogvideox_control.py:715-948
* ``CogVideoXLayerNormZero.forward``               :60-150
* ``AdaLayerNorm.forward``                         :155-197
* ``CogVideoXAttnProcessor2_0.__call__``           :202-270
* ``CogVideoXBlock.forward`` / ``MVBlock.forward`` :394-445 / :313-348
* ``ActionEmbed.forward`` / ``ActionRecon.forward`` /root/reference/orv/models/components.py:47-71 / :92-104

written against a flat ``state_dict`` (the checkpoint key names of SURVEY.md §8(b)) instead of an
``nn.Module`` tree, so that it shares no code with either the reference or the product.
The ORV-authored logic restated here IS pinned: ``tests/golden/*.safetensors`` were produced by
running the reference's own ``forward`` methods (oracle/ref_harness.py + oracle/gen_golden.py) and
``tests/test_oracle_golden.py`` checks this file against them.  The diffusers leaf arithmetic
underneath (oracle/leaf.py) is PARITY UNPINNED - see that file's header.
"""
from __future__ import annotations

import fnmatch
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import leaf

SD = Dict[str, torch.Tensor]

DEFAULT_CONFIG = dict(
    num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16, flip_sin_to_cos=True,
    freq_shift=0, time_embed_dim=512, ofs_embed_dim=None, text_embed_dim=4096, num_layers=30, dropout=0.0,
    attention_bias=True, sample_width=90, sample_height=60, sample_frames=49, patch_size=2, patch_size_t=None,
    temporal_compression_ratio=4, max_text_seq_length=226, activation_fn="gelu-approximate",
    timestep_activation_fn="silu", norm_elementwise_affine=True, norm_eps=1e-5, spatial_interpolation_scale=1.875,
    temporal_interpolation_scale=1.0, use_rotary_positional_embeddings=False,
    use_learned_positional_embeddings=False, patch_bias=True, loaded_pretrained_model_name_or_path=None,
    modulate_encoder_hidden_states=False, num_control_blocks=12, recon_action=False, visual_guidance=False,
    num_control_keys=2, multiview=False, max_n_view=3, from_t2v=False)


def _lin(sd: SD, name: str, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd: SD, name: str, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd.get(name + ".weight"), sd.get(name + ".bias"), eps)


def pad_actions(actions):
    """cogvideox_control.py:805-812: left-pad so that (F+1) % 4 == 0. Returns (padded, pad_frames)."""
    res = (actions.size(1) + 1) % 4
    if res == 0:
        return actions, 0
    pad = 4 - res
    return torch.cat([actions.new_zeros(actions.shape[0], pad, actions.shape[2]), actions], dim=1), pad


def action_embed(sd: SD, actions, is_mask=None, compress_ratio=4, patch_size_t=None, prefix="action_embed"):
    """components.py:47-71. ``actions`` already padded.  ``is_mask`` (bool[B]) replaces the RNG draw."""
    b, f, _ = actions.shape
    x = torch.cat([torch.zeros_like(actions[:, :1]), actions], dim=1)
    if compress_ratio > 1:
        x = x.reshape(b, (f + 1) // compress_ratio, -1)
    if (patch_size_t or 1) > 1:
        x = x.reshape(b, x.shape[1] // patch_size_t, -1)
    x = _lin(sd, prefix + ".mlp.3", F.gelu(_lin(sd, prefix + ".mlp.0", x), approximate="tanh"))
    if is_mask is not None and bool(is_mask.any()):
        x = x.clone()
        x[is_mask] = sd[prefix + ".mask_embed.weight"][None].to(x.dtype).expand(int(is_mask.sum()), x.shape[1], -1)
    return x


def action_recon(sd: SD, action_emb, compress_ratio=4, prefix="action_recon"):
    """components.py:92-104."""
    b, f, _ = action_emb.shape
    x = _lin(sd, prefix + ".mlp.2", F.gelu(_lin(sd, prefix + ".mlp.0", action_emb), approximate="tanh"))
    if compress_ratio > 1:
        x = x.reshape(b, f * compress_ratio, x.shape[-1] // compress_ratio)
    return x[:, 1:]


def time_embedding(sd: SD, cfg, timestep, dtype):
    """cogvideox_control.py:762-769 (+ diffusers Timesteps/TimestepEmbedding)."""
    inner = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    t_emb = leaf.get_timestep_embedding(timestep, inner, cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(dtype)
    return _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))


def ofs_embedding(sd: SD, cfg, ofs, dtype):
    """cogvideox_control.py:771-775."""
    e = leaf.get_timestep_embedding(ofs, cfg["ofs_embed_dim"], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(dtype)
    return _lin(sd, "ofs_embedding.linear_2", F.silu(_lin(sd, "ofs_embedding.linear_1", e)))


def sincos_table(cfg, t_lat, h, w, dtype=torch.float32):
    """Positional table added inside patch-embed (leaf.CogVideoXPatchEmbed); [1, Nt+T*h/p*w/p, D], text rows 0."""
    inner = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    p = cfg["patch_size"]
    pos = leaf.get_3d_sincos_pos_embed(inner, (w // p, h // p), t_lat, cfg["spatial_interpolation_scale"],
                                       cfg["temporal_interpolation_scale"]).flatten(0, 1)
    out = torch.zeros(1, cfg["max_text_seq_length"] + pos.shape[0], inner, dtype=torch.float32)
    out[:, cfg["max_text_seq_length"]:] = pos.to(torch.float32)
    return out.to(dtype)


def patchify(x, p, p_t=None):
    """[B,T,C,H,W] -> token features. p_t None: [B, T*h*w, C*p*p] with feature order (c, ph, pw) (= Conv2d
    weight.flatten(1) order); p_t: [B, T/pt*h*w, C*pt*p*p] with feature order (c, pt, ph, pw).  Index-exact."""
    b, t, c, hh, ww = x.shape
    h, w = hh // p, ww // p
    if p_t is None:
        y = x.reshape(b, t, c, h, p, w, p).permute(0, 1, 3, 5, 2, 4, 6)      # b t h w c ph pw
        return y.reshape(b, t * h * w, c * p * p)
    y = x.permute(0, 1, 3, 4, 2).reshape(b, t // p_t, p_t, h, p, w, p, c)
    return y.permute(0, 1, 3, 5, 7, 2, 4, 6).flatten(4, 7).flatten(1, 3)


def patch_embed(sd: SD, cfg, text, latents):
    """diffusers CogVideoXPatchEmbed.forward, called at cogvideox_control.py:788/833/842."""
    p, p_t = cfg["patch_size"], cfg["patch_size_t"]
    b, t, c, h, w = latents.shape
    txt = _lin(sd, "patch_embed.text_proj", text)
    tok = patchify(latents, p, p_t)
    wgt = sd["patch_embed.proj.weight"]
    vis = F.linear(tok, wgt.reshape(wgt.shape[0], -1), sd.get("patch_embed.proj.bias"))
    emb = torch.cat([txt, vis], dim=1)
    if not cfg["use_rotary_positional_embeddings"] or cfg["use_learned_positional_embeddings"]:
        if text.shape[1] != cfg["max_text_seq_length"]:
            raise ValueError("text length must equal max_text_seq_length when positional embeddings are added")
        emb = emb + sincos_table(cfg, t, h, w, emb.dtype)
    return emb


def unpatchify(x, b, t, h, w, p, p_t=None):
    """cogvideox_control.py:926-936. x [B, N, p*p*(pt)*C] -> [B,T,C,H,W]. Index-exact."""
    if p_t is None:
        out = x.reshape(b, t, h // p, w // p, -1, p, p)
        return out.permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    out = x.reshape(b, (t + p_t - 1) // p_t, h // p, w // p, -1, p_t, p, p)
    return out.permute(0, 1, 5, 4, 2, 6, 3, 7).flatten(6, 7).flatten(4, 5).flatten(1, 2)


def frame_index(n_vis_tokens, n_frames):
    """repeat_interleave map of :99-105: token -> frame = token // (n_vis_tokens // n_frames)."""
    return torch.arange(n_vis_tokens) // (n_vis_tokens // n_frames)


def layernorm_zero(sd: SD, prefix, cfg, vis, txt, temb, action_emb):
    """cogvideox_control.py:60-150. Returns (vis_mod, txt_mod, gate[B,Nv|1,D], enc_gate[B,1,D]|None)."""
    d, eps = vis.shape[-1], cfg["norm_eps"]
    w, bias = sd[prefix + ".linear.weight"], sd.get(prefix + ".linear.bias")
    nv, ne = _ln(sd, prefix + ".norm", vis, eps), _ln(sd, prefix + ".norm", txt, eps)
    if not cfg["modulate_encoder_hidden_states"]:
        if action_emb is None:
            shift, scale, gate = F.linear(F.silu(temb), w, bias).chunk(3, dim=-1)
            return nv * (1 + scale)[:, None] + shift[:, None], ne, gate[:, None], None
        shift, scale, gate = F.linear(F.silu(temb[:, None] + action_emb), w, bias).chunk(3, dim=-1)
        idx = frame_index(vis.shape[1], action_emb.shape[1])
        return nv * (1 + scale[:, idx]) + shift[:, idx], ne, gate[:, idx], None
    if action_emb is None:
        shift, scale, gate, es, esc, eg = F.linear(F.silu(temb), w, bias).chunk(6, dim=-1)
        return (nv * (1 + scale)[:, None] + shift[:, None], ne * (1 + esc)[:, None] + es[:, None],
                gate[:, None], eg[:, None])
    bv = None if bias is None else bias[: 3 * d]
    bt = None if bias is None else bias[3 * d:]
    shift, scale, gate = F.linear(F.silu(temb[:, None] + action_emb), w[: 3 * d], bv).chunk(3, dim=-1)
    es, esc, eg = F.linear(F.silu(temb), w[3 * d:], bt).chunk(3, dim=-1)
    idx = frame_index(vis.shape[1], action_emb.shape[1])
    return (nv * (1 + scale[:, idx]) + shift[:, idx], ne * (1 + esc)[:, None] + es[:, None], gate[:, idx],
            eg[:, None])


def ada_layernorm_out(sd: SD, cfg, x, temb, action_emb, prefix="norm_out"):
    """cogvideox_control.py:155-197 as configured for norm_out (chunk_dim=1): order is (shift, scale)."""
    eps = cfg["norm_eps"]
    if action_emb is None:
        shift, scale = _lin(sd, prefix + ".linear", F.silu(temb)).chunk(2, dim=1)
        return _ln(sd, prefix + ".norm", x, eps) * (1 + scale[:, None]) + shift[:, None]
    shift, scale = _lin(sd, prefix + ".linear", F.silu(temb[:, None] + action_emb)).chunk(2, dim=2)
    idx = frame_index(x.shape[1], action_emb.shape[1])
    return _ln(sd, prefix + ".norm", x, eps) * (1 + scale[:, idx]) + shift[:, idx]


def joint_attention(sd: SD, prefix, cfg, vis, txt, rope=None):
    """cogvideox_control.py:202-270 (text first, qk LayerNorm eps 1e-6, RoPE on video rows only)."""
    heads, hd = cfg["num_attention_heads"], cfg["attention_head_dim"]
    nt = 0 if txt is None else txt.shape[1]
    x = vis if txt is None else torch.cat([txt, vis], dim=1)
    b, s, _ = x.shape
    q = _lin(sd, prefix + ".to_q", x).view(b, s, heads, hd).transpose(1, 2)
    k = _lin(sd, prefix + ".to_k", x).view(b, s, heads, hd).transpose(1, 2)
    v = _lin(sd, prefix + ".to_v", x).view(b, s, heads, hd).transpose(1, 2)
    if prefix + ".norm_q.weight" in sd:
        q, k = _ln(sd, prefix + ".norm_q", q, 1e-6), _ln(sd, prefix + ".norm_k", k, 1e-6)
    if rope is not None:
        q = torch.cat([q[:, :, :nt], leaf.apply_rotary_emb(q[:, :, nt:], rope)], dim=2)
        k = torch.cat([k[:, :, :nt], leaf.apply_rotary_emb(k[:, :, nt:], rope)], dim=2)
    att = torch.softmax((q.float() @ k.float().transpose(-1, -2)) / math.sqrt(hd), dim=-1).to(v.dtype) @ v
    out = _lin(sd, prefix + ".to_out.0", att.transpose(1, 2).reshape(b, s, heads * hd))
    return out[:, nt:], (out[:, :nt] if nt else None)


def block(sd: SD, prefix, cfg, vis, txt, temb, rope, action_emb):
    """cogvideox_control.py:394-445."""
    mod = cfg["modulate_encoder_hidden_states"]
    nv, ne, g, eg = layernorm_zero(sd, prefix + ".norm1", cfg, vis, txt, temb, action_emb)
    av, ae = joint_attention(sd, prefix + ".attn1", cfg, nv, ne if mod else None, rope)
    vis = vis + g * av
    if mod:
        txt = txt + eg * ae
    nv, ne, g, eg = layernorm_zero(sd, prefix + ".norm2", cfg, vis, txt, temb, action_emb)

    def ff(x):
        return _lin(sd, prefix + ".ff.net.2", F.gelu(_lin(sd, prefix + ".ff.net.0.proj", x), approximate="tanh"))

    if not mod:
        return vis + g * ff(nv), txt
    nt = txt.shape[1]
    o = ff(torch.cat([ne, nv], dim=1))
    return vis + g * o[:, nt:], txt + eg * o[:, :nt]


def mv_block(sd: SD, prefix, cfg, vis, txt, temb, n_view, n_frame):
    """cogvideox_control.py:313-348 (image_rotary_emb_view is never supplied by any caller)."""
    nv, ne, g, _ = layernorm_zero(sd, prefix + ".norm1", cfg, vis, txt, temb, None)
    bv, n, d = nv.shape
    b, s = bv // n_view, n // n_frame
    x = nv.reshape(b, n_view, n_frame, s, d).permute(0, 2, 1, 3, 4).reshape(b * n_frame, n_view * s, d)
    e = None
    if cfg["modulate_encoder_hidden_states"]:
        e = ne.reshape(b, n_view * ne.shape[1], d).repeat_interleave(n_frame, dim=0)
    a, _ = joint_attention(sd, prefix + ".attn1", cfg, x, e, None)
    a = _lin(sd, prefix + ".proj_out", a)
    a = a.reshape(b, n_frame, n_view, s, d).permute(0, 2, 1, 3, 4).reshape(bv, n, d)
    return vis + g * a


def dit_forward(sd: SD, cfg: dict, hidden_states, encoder_hidden_states, timestep, actions=None, depths=None,
                labels=None, is_mask=None, image_rotary_emb=None, ofs=None, num_views=1, training=False,
                return_intermediates=False):
    """cogvideox_control.py:715-948. Returns (sample, action_emb, actions_recon[, intermediates])."""
    cfg = {**DEFAULT_CONFIG, **cfg}
    inter = {}
    if num_views > 1:                                                               # :756-758
        bb, vf = hidden_states.shape[:2]
        hidden_states = hidden_states.reshape(bb * num_views, vf // num_views, *hidden_states.shape[2:])
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_views, dim=0)
    b, t, _, h, w = hidden_states.shape
    dtype = hidden_states.dtype
    temb = time_embedding(sd, cfg, timestep, dtype)                                 # :762-769
    if cfg["ofs_embed_dim"]:
        temb = temb + ofs_embedding(sd, cfg, ofs, dtype)                            # :771-775
    if num_views > 1:
        temb = temb.repeat_interleave(num_views, dim=0)
    emb = patch_embed(sd, cfg, encoder_hidden_states, hidden_states)                # :788
    nt = encoder_hidden_states.shape[1]
    txt, vis = emb[:, :nt], emb[:, nt:]
    inter["embed"] = emb
    if num_views > 1:                                                               # :797-800
        s = vis.shape[1] // t
        pv = sd_pos_embedding_v(cfg, num_views).to(dtype)
        x = vis.reshape(b // num_views, num_views, t, s, -1).permute(0, 2, 1, 3, 4).reshape(-1, num_views * s, vis.shape[-1])
        x = x + pv
        vis = x.reshape(b // num_views, t, num_views, s, -1).permute(0, 2, 1, 3, 4).reshape(b, t * s, -1)

    action_emb = recon = None
    if actions is not None:                                                         # :805-825
        padded, pad = pad_actions(actions)
        action_emb = action_embed(sd, padded, is_mask, 4, cfg["patch_size_t"])
        if num_views > 1:
            action_emb = action_emb.repeat_interleave(num_views, dim=0)
        if training and cfg["recon_action"]:
            recon = action_recon(sd, action_emb)
            if pad > 0:
                recon = recon[:, pad:]
        inter["action_emb"] = action_emb

    ctrl = []
    if cfg["visual_guidance"]:                                                      # :828-858
        for c in (depths, labels):
            if c is not None:
                if num_views > 1:
                    c = c.reshape(c.shape[0] * num_views, c.shape[1] // num_views, *c.shape[2:])
                ctrl.append(patch_embed(sd, cfg, encoder_hidden_states, c)[:, nt:])
    if ctrl:
        assert len(ctrl) == cfg["num_control_keys"]
        ctrl = torch.cat(ctrl, dim=-1)
        vis = vis + _lin(sd, "initial_combine_linear", vis.repeat(1, 1, cfg["num_control_keys"]) + ctrl)
        inter["fused"] = vis

    for i in range(cfg["num_layers"]):                                              # :861-907
        if cfg["multiview"]:
            vis = mv_block(sd, f"mv_blocks.{i}", cfg, vis, txt, temb, num_views, t)
        vis, txt = block(sd, f"transformer_blocks.{i}", cfg, vis, txt, temb, image_rotary_emb, action_emb)
        inter[f"block{i}"] = torch.cat([txt, vis], dim=1)

    eps = cfg["norm_eps"]
    if fnmatch.fnmatch(str(cfg["loaded_pretrained_model_name_or_path"]), "*CogVideoX*-5b*"):   # :909-916
        vis = _ln(sd, "norm_final", torch.cat([txt, vis], dim=1), eps)[:, nt:]
    else:
        vis = _ln(sd, "norm_final", vis, eps)
    vis = ada_layernorm_out(sd, cfg, vis, temb, action_emb)                         # :919
    vis = _lin(sd, "proj_out", vis)                                                 # :920
    out = unpatchify(vis, b, t, h, w, cfg["patch_size"], cfg["patch_size_t"])       # :926-936
    if num_views > 1:
        out = out.reshape(b // num_views, num_views * out.shape[1], *out.shape[2:])
    if return_intermediates:
        return out, action_emb, recon, inter
    return out, action_emb, recon


def sd_pos_embedding_v(cfg, n_view):
    """cogvideox_control.py:650-688: per-view sincos table [1, n_view*h/p*w/p, D] (non-persistent buffer)."""
    inner = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    p = cfg["patch_size"]
    pos = leaf.get_3d_sincos_pos_embed(inner, (cfg["sample_width"] // p, cfg["sample_height"] // p),
                                       cfg["max_n_view"], cfg["spatial_interpolation_scale"], 1.0)
    return pos[:n_view].flatten(0, 1)[None].to(torch.float32)


def compute_action_loss(x, x_recon, loss_weight, mask=None):
    """cogvideox_control.py:690-713."""
    if mask is None:
        mask = torch.ones((x.size(0),), dtype=torch.bool)
    rot = 1 - torch.cos(x_recon[mask][..., 3:6] - x[mask][..., 3:6]).mean()
    grip_pred = torch.sigmoid(x_recon[..., -1])
    pos = F.smooth_l1_loss(x_recon[mask][..., :3], x[mask][..., :3])
    grip = F.smooth_l1_loss(grip_pred[mask], x[mask][..., -1])
    return rot * loss_weight["rot_loss"], pos * loss_weight["pos_loss"], grip * loss_weight["grip_loss"]
