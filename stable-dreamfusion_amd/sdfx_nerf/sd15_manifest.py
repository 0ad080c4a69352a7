"""The PUBLISHED Stable-Diffusion-1.5 parameter layout — key -> shape of `UNet2DConditionModel` and of the `AutoencoderKL` encoder as
diffusers serialises them — derived here from the published configuration, and the map between it and this package's restatement
(sdfx_nerf/sd15_arch.py: `UNetSD15`, `VAEEncoderSD15`).

Why: diffusers, transformers and the hub weights are absent from this image (guidance/sd_utils.py:37-65 pulls
`runwayml/stable-diffusion-v1-5`), so the frozen prior the bench times is a restatement with random weights. This file pins that
restatement to something other than itself: the layout below is generated from the published config values only
(unet/config.json: block_out_channels [320, 640, 1280, 1280], layers_per_block 2, attention_head_dim 8 (= 8 heads), cross_attention_dim
768, norm_num_groups 32, in / out channels 4, down blocks CrossAttnDownBlock2D x 3 + DownBlock2D, conv projections in the
transformers; vae/config.json: block_out_channels [128, 256, 512, 512], layers_per_block 2, latent_channels 4), its element totals
are checked against the published parameter counts — 859 520 964 for the UNet, 34 163 592 for the VAE encoder (+ 72 for
`quant_conv`) — and tests/test_sd15_manifest.py checks that the restatement's `state_dict()` maps onto it one-to-one, shape for shape.
`load_published` then loads real weights when a user has them (safetensors / torch state dicts in diffusers' naming, old or new
attention names).
"""
from __future__ import annotations

from collections import OrderedDict

UNET_PARAMS_PUBLISHED = 859_520_964        # runwayml/stable-diffusion-v1-5 unet (the "860 M" of the model card)
VAE_ENCODER_PARAMS_PUBLISHED = 34_163_592  # AutoencoderKL.encoder of the same repository (the full VAE: 83 653 863)
VAE_QUANT_CONV_PARAMS = 72

UNET_CONFIG = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, heads=8,
                   cross_attention_dim=768, cross_attn_down=(True, True, True, False))
VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2)


def _conv(d, name, cout, cin, k):
    d[name + ".weight"] = (cout, cin, k, k)
    d[name + ".bias"] = (cout,)


def _lin(d, name, cout, cin, bias=True):
    d[name + ".weight"] = (cout, cin)
    if bias:
        d[name + ".bias"] = (cout,)


def _norm(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)


def _resnet(d, name, cin, cout, temb):
    _norm(d, name + ".norm1", cin)
    _conv(d, name + ".conv1", cout, cin, 3)
    if temb:
        _lin(d, name + ".time_emb_proj", cout, temb)
    _norm(d, name + ".norm2", cout)
    _conv(d, name + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, name + ".conv_shortcut", cout, cin, 1)


def _transformer(d, name, c, ctx):
    _norm(d, name + ".norm", c)
    _conv(d, name + ".proj_in", c, c, 1)
    t = name + ".transformer_blocks.0"
    _norm(d, t + ".norm1", c)
    for n in ("to_q", "to_k", "to_v"):
        _lin(d, f"{t}.attn1.{n}", c, c, bias=False)
    _lin(d, t + ".attn1.to_out.0", c, c)
    _norm(d, t + ".norm2", c)
    _lin(d, t + ".attn2.to_q", c, c, bias=False)
    _lin(d, t + ".attn2.to_k", c, ctx, bias=False)
    _lin(d, t + ".attn2.to_v", c, ctx, bias=False)
    _lin(d, t + ".attn2.to_out.0", c, c)
    _norm(d, t + ".norm3", c)
    _lin(d, t + ".ff.net.0.proj", 8 * c, c)     # GEGLU: value and gate halves
    _lin(d, t + ".ff.net.2", c, 4 * c)
    _conv(d, name + ".proj_out", c, c, 1)


def unet_manifest(cfg=UNET_CONFIG):
    """key -> shape of diffusers' UNet2DConditionModel for `cfg` (default: the published SD-1.5 configuration)."""
    d = OrderedDict()
    ch, L, ctx = cfg["block_out_channels"], cfg["layers_per_block"], cfg["cross_attention_dim"]
    temb = ch[0] * 4
    _conv(d, "conv_in", ch[0], cfg["in_channels"], 3)
    _lin(d, "time_embedding.linear_1", temb, ch[0])
    _lin(d, "time_embedding.linear_2", temb, temb)
    skips, c = [ch[0]], ch[0]
    for i, co in enumerate(ch):
        for j in range(L):
            _resnet(d, f"down_blocks.{i}.resnets.{j}", c, co, temb)
            if cfg["cross_attn_down"][i]:
                _transformer(d, f"down_blocks.{i}.attentions.{j}", co, ctx)
            c = co
            skips.append(c)
        if i < len(ch) - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", c, c, 3)
            skips.append(c)
    _resnet(d, "mid_block.resnets.0", c, c, temb)
    _transformer(d, "mid_block.attentions.0", c, ctx)
    _resnet(d, "mid_block.resnets.1", c, c, temb)
    for b, (i, co) in enumerate(reversed(list(enumerate(ch)))):
        for j in range(L + 1):
            _resnet(d, f"up_blocks.{b}.resnets.{j}", c + skips.pop(), co, temb)
            if cfg["cross_attn_down"][i]:
                _transformer(d, f"up_blocks.{b}.attentions.{j}", co, ctx)
            c = co
        if i > 0:
            _conv(d, f"up_blocks.{b}.upsamplers.0.conv", c, c, 3)
    _norm(d, "conv_norm_out", c)
    _conv(d, "conv_out", cfg["out_channels"], c, 3)
    return d


def vae_encoder_manifest(cfg=VAE_CONFIG):
    """key -> shape of the encoder half of diffusers' AutoencoderKL (+ `quant_conv`), current attention names (`to_q` ...)."""
    d = OrderedDict()
    ch, L, z = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    _conv(d, "encoder.conv_in", ch[0], cfg["in_channels"], 3)
    c = ch[0]
    for i, co in enumerate(ch):
        for j in range(L):
            _resnet(d, f"encoder.down_blocks.{i}.resnets.{j}", c, co, 0)
            c = co
        if i < len(ch) - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    _resnet(d, "encoder.mid_block.resnets.0", c, c, 0)
    a = "encoder.mid_block.attentions.0"
    _norm(d, a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, f"{a}.{n}", c, c)
    _resnet(d, "encoder.mid_block.resnets.1", c, c, 0)
    _norm(d, "encoder.conv_norm_out", c)
    _conv(d, "encoder.conv_out", 2 * z, c, 3)
    _conv(d, "quant_conv", 2 * z, 2 * z, 1)
    return d


def numel(manifest, prefix=""):
    n = 0
    for k, shape in manifest.items():
        if k.startswith(prefix):
            e = 1
            for s in shape:
                e *= s
            n += e
    return n


# ---- this package's names -> the published names ---------------------------------------------------------------------------------
_RES = {"norm1": "norm1", "conv1": "conv1", "temb": "time_emb_proj", "norm2": "norm2", "conv2": "conv2", "skip": "conv_shortcut"}
_TRF = {"norm": "norm", "proj_in": "proj_in", "proj_out": "proj_out", "n1": "transformer_blocks.0.norm1", "n2": "transformer_blocks.0.norm2",
        "n3": "transformer_blocks.0.norm3", "ff_in": "transformer_blocks.0.ff.net.0.proj", "ff_out": "transformer_blocks.0.ff.net.2"}
_ATT = {"q": "to_q", "k": "to_k", "v": "to_v", "o": "to_out.0"}


def _res_key(rest):
    part, leaf = rest.split(".", 1)
    return _RES[part] + "." + leaf


def _trf_key(rest):
    part, leaf = rest.split(".", 1)
    if part in ("attn1", "attn2"):
        proj, leaf2 = leaf.split(".", 1)
        return f"transformer_blocks.0.{part}.{_ATT[proj]}.{leaf2}"
    return _TRF[part] + "." + leaf


def unet_key(ours: str, layers_per_block=2, levels=4) -> str:
    """`UNetSD15.state_dict()` key -> the published key."""
    p = ours.split(".")
    if p[0] == "time":
        return f"time_embedding.linear_{1 if p[1] == '0' else 2}.{p[2]}"
    if p[0] in ("conv_in", "conv_out"):
        return ours
    if p[0] == "norm_out":
        return "conv_norm_out." + p[1]
    if p[0] == "down":
        idx, slot, rest = int(p[1]), p[2], ".".join(p[3:])
        per = layers_per_block + 1                      # resnets + the downsampler of a level (the last level has none)
        lvl, j = (idx // per, idx % per) if idx < per * (levels - 1) else (levels - 1, idx - per * (levels - 1))
        if slot == "0" and j == layers_per_block and lvl < levels - 1:
            return f"down_blocks.{lvl}.downsamplers.0.conv.{rest}"
        return f"down_blocks.{lvl}.resnets.{j}.{_res_key(rest)}" if slot == "0" else f"down_blocks.{lvl}.attentions.{j}.{_trf_key(rest)}"
    if p[0] == "mid":
        rest = ".".join(p[2:])
        if p[1] == "1":
            return "mid_block.attentions.0." + _trf_key(rest)
        return f"mid_block.resnets.{0 if p[1] == '0' else 1}." + _res_key(rest)
    if p[0] == "up":
        idx, slot, rest = int(p[1]), p[2], ".".join(p[3:])
        b, j = idx // (layers_per_block + 1), idx % (layers_per_block + 1)
        if slot == "0":
            return f"up_blocks.{b}.resnets.{j}.{_res_key(rest)}"
        if slot == "1":
            return f"up_blocks.{b}.attentions.{j}.{_trf_key(rest)}"
        return f"up_blocks.{b}.upsamplers.0.conv.{rest}"
    raise KeyError(ours)


def vae_key(ours: str, layers_per_block=2, levels=4) -> str:
    """`VAEEncoderSD15.state_dict()` key -> the published key."""
    p = ours.split(".")
    if p[0] == "conv_in":
        return "encoder." + ours
    if p[0] == "blocks":
        idx, rest = int(p[1]), ".".join(p[2:])
        per = layers_per_block + 1
        lvl, j = (idx // per, idx % per) if idx < per * (levels - 1) else (levels - 1, idx - per * (levels - 1))
        if j == layers_per_block and lvl < levels - 1:
            return f"encoder.down_blocks.{lvl}.downsamplers.0.conv.{rest}"
        return f"encoder.down_blocks.{lvl}.resnets.{j}.{_res_key(rest)}"
    if p[0] in ("mid1", "mid2"):
        return f"encoder.mid_block.resnets.{0 if p[0] == 'mid1' else 1}.{_res_key('.'.join(p[1:]))}"
    if p[0] == "mid_norm":
        return "encoder.mid_block.attentions.0.group_norm." + p[1]
    if p[0] == "mid_attn":
        return f"encoder.mid_block.attentions.0.{_ATT[p[1]]}.{p[2]}"
    if p[0] == "norm_out":
        return "encoder.conv_norm_out." + p[1]
    if p[0] == "conv_out":
        return "encoder." + ours
    if p[0] == "quant":
        return "quant_conv." + p[1]
    raise KeyError(ours)


_OLD_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}   # diffusers < 0.18 names


def load_published(module, state_dict, kind: str, strict: bool = True):
    """Load a published state dict (diffusers naming; `kind` = "unet" for UNet2DConditionModel, "vae" for AutoencoderKL — decoder keys
    are ignored) into this package's `UNetSD15` / `VAEEncoderSD15`. Tensors are matched by the manifest's names and shapes; the old
    VAE attention names (`query` / `key` / `value` / `proj_attn`, some stored as [C, C, 1, 1]) are accepted. Returns the keys used."""
    import torch
    key_of = unet_key if kind == "unet" else vae_key
    manifest = unet_manifest() if kind == "unet" else vae_encoder_manifest()
    src = {}
    for k, v in state_dict.items():
        for old, new in _OLD_VAE_ATTN.items():
            k = k.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")
        src[k] = v
    own = module.state_dict()
    out, used = {}, []
    for ours, t in own.items():
        pub = key_of(ours)
        if pub not in src:
            if strict:
                raise KeyError(f"published state dict has no `{pub}` (for `{ours}`)")
            continue
        v = src[pub]
        if tuple(v.shape) != tuple(t.shape) and v.numel() == t.numel():
            v = v.reshape(t.shape)                       # [C, C, 1, 1] <-> [C, C]
        if tuple(v.shape) != manifest[pub] and tuple(v.shape) != tuple(t.shape):
            raise ValueError(f"`{pub}`: shape {tuple(v.shape)}, the published layout has {manifest[pub]}")
        out[ours] = v.to(dtype=t.dtype)
        used.append(pub)
    module.load_state_dict(out, strict=strict)
    return used
