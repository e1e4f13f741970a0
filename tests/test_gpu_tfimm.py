"""GPU parity of the tfimm ViT / ConvNeXt paths against the fp32 CPU oracle."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops_ref as R  # noqa: E402
from oracle import tfimm_ref as ref  # noqa: E402
from tests.test_gpu_resnet_rs import _images  # noqa: E402
from tests.test_gpu_ops import check, dev, h  # noqa: E402


@pytest.mark.parametrize("B,N,heads", [(2, 197, 3), (3, 50, 6), (1, 224, 12), (2, 16, 1)])
def test_mhsa(B, N, heads, report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops
    D = heads * 64
    g = torch.Generator().manual_seed(N + heads)
    qkv = h(torch.randn(B, N, 3 * D, generator=g))
    q, k, v = qkv.reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    attn = torch.softmax((64 ** -0.5) * (q @ k.transpose(-1, -2)), dim=-1)
    ref_o = (attn @ v).permute(0, 2, 1, 3).reshape(B, N, D)
    got = ops.mhsa(dev(qkv), heads, 64 ** -0.5)
    torch.cuda.synchronize()
    check(report, f"mhsa B{B} N{N} heads{heads}", got, ref_o, tol=3e-3)


def test_vit_tokens(report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops
    g = torch.Generator().manual_seed(5)
    pt = h(torch.randn(3, 196, 192, generator=g))
    cls = h(torch.randn(192, generator=g))
    pos = h(torch.randn(197, 192, generator=g))
    ref_t = torch.cat([cls.expand(3, 1, 192), pt], dim=1) + pos
    got = ops.vit_tokens(dev(pt), dev(cls), dev(pos))
    check(report, "vit_tokens", got, ref_t)


def _model_check(report, tag, z, z_ref, ca, cb):
    for i, (a, b) in enumerate(zip(ca, cb)):
        b = b.float().cpu().reshape(a.shape)
        report(f"[{tag}] stage {i} ref_rms {a.pow(2).mean().sqrt().item():.3f} "
               f"rel_rms_err {((a - b) ** 2).mean().sqrt().item() / a.pow(2).mean().sqrt().item():.3e}")
    ze = (z.cpu() - z_ref).abs().max().item()
    report(f"[{tag}] logit max_abs_err={ze:.3e} logit mean={z_ref.mean().item():.3f} std={z_ref.std().item():.3f}")
    return ze


@pytest.mark.parametrize("name", ["vit_tiny_patch16_224", "vit_small_patch16_224", "vit_base_patch16_224"])
def test_vit(name, report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops, tfimm_models as tm
    cfg = tm.VIT_CONFIGS[name]
    p = tm.vit_synth_params(cfg, seed=1010)
    x = _images(2 if "base" in name else 4, 224).to(torch.float16).to(torch.float32)
    ca, cb = [], []
    with torch.no_grad():
        ref.vit_forward_tokens(p, x, name, collect=ca)
        z_ref = ref.vit_logits(p, x, name)
    m = tm.ViT(p, cfg)
    xd = ops.to_device_nhwc8(x)
    m.features(xd, collect=cb)
    z = m.logits(xd)
    torch.cuda.synchronize()
    ze = _model_check(report, name, z, z_ref, ca[::4], cb[::4])
    assert ze < 3e-3 * max(1.0, z_ref.abs().max().item())


@pytest.mark.parametrize("name", ["convnext_tiny_in22k", "convnext_small_in22k", "convnext_base_in22k", "convnext_large_in22ft1k"])
def test_convnext(name, report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops, tfimm_models as tm
    cfg = tm.CONVNEXT_CONFIGS[name]
    p = tm.convnext_synth_params(cfg, seed=1000)
    x = _images(2, 200).to(torch.float16).to(torch.float32)
    ca, cb = [], []
    with torch.no_grad():
        ref.convnext_features(p, x, name, collect=ca)
        z_ref = ref.convnext_logits(p, x, name)
    m = tm.ConvNeXt(p, cfg)
    xd = ops.to_device_nhwc8(x)
    m.features(xd, collect=cb)
    z = m.logits(xd)
    torch.cuda.synchronize()
    ze = _model_check(report, name, z, z_ref, ca, cb)
    assert ze < 3e-3 * max(1.0, z_ref.abs().max().item())


def test_vit_interpolate_input(report):
    """ViTConfig.interpolate_input (vit.py:58,425-433; layers/transformers.py:13-47): a 224-pixel ViT scored at 192 x 192 (12 x 12 patch
    grid, 145 tokens) with its position embeddings resampled - fast and strict precision against the oracle graph."""
    import dataclasses
    import vipcup_amd  # noqa: F401
    from oracle import tfimm_ref
    from vipcup_amd import ops, tfimm_models as tm
    key = "vit_tiny_patch16_224"
    cfg = dataclasses.replace(tm.VIT_CONFIGS[key], interpolate_input=True)
    p = tm.vit_synth_params(cfg, 1008)
    g = torch.Generator().manual_seed(9)
    x = torch.rand((3, 192, 192, 3), generator=g)
    with torch.no_grad():
        z = tfimm_ref.vit_logits(p, x, key, interpolate_input=True)
    for mode, tol in (("fast", 5e-3), ("strict", 2e-5)):
        with ops.precision(mode):
            m = tm.ViT(p, cfg)
        got = m.logits(ops.to_device_nhwc8(x, dtype=ops.act_dtype(mode))).cpu()
        d = (got - z).abs().max().item()
        report(f"[tfimm] {key} at 192x192 with interpolate_input, {mode}: max|dz| {d:.2e} (logit std {z.std().item():.3f})")
        assert d <= tol
    with pytest.raises(ValueError):      # the default configuration refuses another input size, as the reference's fixed pos_embed add does
        tm.ViT(p, tm.VIT_CONFIGS[key]).logits(ops.to_device_nhwc8(x))
