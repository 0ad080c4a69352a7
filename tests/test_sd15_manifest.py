"""The frozen prior's restatement (sdfx_nerf/sd15_arch.py) against the PUBLISHED SD-1.5 layout (sdfx_nerf/sd15_manifest.py).

diffusers and the hub weights are absent, so nothing here can compare activations. What can be pinned without them:
  * the manifest itself, generated from the published config values, has the published parameter counts (859 520 964 for the
    UNet, 34 163 592 for the VAE encoder) and the published key names (spot-checked against names every SD-1.5 checkpoint holds);
  * the restatement's state_dict maps onto the manifest ONE-TO-ONE, shape for shape — so real weights would load, and the network
    the bench times has the published network's layers, widths and parameter count;
  * `load_published` round-trips a synthetic checkpoint written in the published naming (old and new VAE attention names).
"""
import importlib

import pytest
import torch

importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import sd15_manifest as M  # noqa: E402


def test_manifest_has_the_published_parameter_counts_and_names():
    u, v = M.unet_manifest(), M.vae_encoder_manifest()
    assert M.numel(u) == M.UNET_PARAMS_PUBLISHED == 859_520_964
    assert M.numel(v, "encoder.") == M.VAE_ENCODER_PARAMS_PUBLISHED == 34_163_592
    assert M.numel(v, "quant_conv.") == M.VAE_QUANT_CONV_PARAMS
    assert len(u) == 686                                  # tensors in diffusers' SD-1.5 UNet state dict
    # names and shapes every published SD-1.5 UNet checkpoint holds
    for k, shape in {"conv_in.weight": (320, 4, 3, 3), "time_embedding.linear_1.weight": (1280, 320),
                     "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight": (320, 768),
                     "down_blocks.1.resnets.0.conv_shortcut.weight": (640, 320, 1, 1),
                     "down_blocks.2.attentions.1.transformer_blocks.0.ff.net.0.proj.weight": (10240, 1280),
                     "down_blocks.3.resnets.1.time_emb_proj.weight": (1280, 1280),
                     "mid_block.attentions.0.proj_in.weight": (1280, 1280, 1, 1),
                     "up_blocks.0.resnets.2.conv1.weight": (1280, 2560, 3, 3), "up_blocks.1.resnets.2.conv1.weight": (1280, 1920, 3, 3),
                     "up_blocks.2.resnets.2.conv1.weight": (640, 960, 3, 3), "up_blocks.3.resnets.0.conv1.weight": (320, 960, 3, 3),
                     "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.2.weight": (320, 1280),
                     "up_blocks.2.upsamplers.0.conv.weight": (640, 640, 3, 3), "conv_norm_out.weight": (320,),
                     "conv_out.weight": (4, 320, 3, 3)}.items():
        assert u[k] == shape, k
    assert "down_blocks.3.attentions.0.norm.weight" not in u and "up_blocks.0.attentions.0.norm.weight" not in u   # DownBlock2D / UpBlock2D
    assert "up_blocks.3.upsamplers.0.conv.weight" not in u and "down_blocks.3.downsamplers.0.conv.weight" not in u
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.bias" not in u                                # projections without bias
    for k, shape in {"encoder.conv_in.weight": (128, 3, 3, 3), "encoder.down_blocks.1.resnets.0.conv_shortcut.weight": (256, 128, 1, 1),
                     "encoder.mid_block.attentions.0.to_q.bias": (512,), "encoder.conv_out.weight": (8, 512, 3, 3),
                     "quant_conv.weight": (8, 8, 1, 1)}.items():
        assert v[k] == shape, k


@pytest.fixture(scope="module")
def nets():
    from sdfx_nerf import sd15_arch as A
    with torch.device("meta"):                            # shapes only: no 3.4 GB of random numbers
        return A.UNetSD15(), A.VAEEncoderSD15()


def test_restatement_maps_onto_the_published_layout_one_to_one(nets):
    for net, key_of, manifest in ((nets[0], M.unet_key, M.unet_manifest()), (nets[1], M.vae_key, M.vae_encoder_manifest())):
        sd = net.state_dict()
        mapped = {key_of(k): tuple(t.shape) for k, t in sd.items()}
        assert len(mapped) == len(sd), "two of this package's keys map to one published key"
        missing, extra = sorted(set(manifest) - set(mapped)), sorted(set(mapped) - set(manifest))
        assert not missing and not extra, (missing[:5], extra[:5])
        wrong = {k: (mapped[k], manifest[k]) for k in manifest if mapped[k] != manifest[k]}
        assert not wrong, dict(list(wrong.items())[:5])
    assert sum(t.numel() for t in nets[0].state_dict().values()) == M.UNET_PARAMS_PUBLISHED
    assert sum(t.numel() for t in nets[1].state_dict().values()) == M.VAE_ENCODER_PARAMS_PUBLISHED + M.VAE_QUANT_CONV_PARAMS


def test_load_published_round_trips_a_checkpoint_in_the_published_naming():
    from sdfx_nerf import sd15_arch as A
    torch.manual_seed(0)
    vae = A.VAEEncoderSD15()
    g = torch.Generator().manual_seed(1)
    ckpt = {k: torch.randn(shape, generator=g) for k, shape in M.vae_encoder_manifest().items()}
    ckpt["decoder.conv_in.weight"] = torch.zeros(1)       # a full AutoencoderKL file also holds the decoder: ignored
    # an old-style file: query / key / value / proj_attn, stored as 1 x 1 convolutions
    old = {}
    for k, t in ckpt.items():
        for new, name in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f".attentions.0.{new}." in k:
                k = k.replace(f".attentions.0.{new}.", f".attentions.0.{name}.")
                t = t.reshape(*t.shape, 1, 1) if t.dim() == 2 else t
        old[k] = t
    for file in (ckpt, old):
        used = M.load_published(vae, file, "vae")
        assert len(used) == len(M.vae_encoder_manifest())
        for ours, t in vae.state_dict().items():
            assert torch.equal(t, ckpt[M.vae_key(ours)].reshape(t.shape)), ours
    x = torch.randn(1, 3, 32, 32)
    assert vae.encode_sample(x).shape == (1, 4, 4, 4) and torch.isfinite(vae.encode_sample(x)).all()
    with pytest.raises(KeyError):
        M.load_published(vae, {k: v for k, v in ckpt.items() if k != "encoder.conv_in.bias"}, "vae")
    # a small UNet of the same topology: every key of its layout is found and loaded
    cfg = dict(M.UNET_CONFIG, block_out_channels=(32, 64, 64, 64), cross_attention_dim=48)
    unet = A.UNetSD15(base=32, mult=(1, 2, 2, 2), ctx_dim=48, heads=4)
    man = M.unet_manifest(cfg)
    assert {M.unet_key(k) for k in unet.state_dict()} == set(man)
    assert all(tuple(t.shape) == man[M.unet_key(k)] for k, t in unet.state_dict().items())
