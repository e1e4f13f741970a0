"""GPU parity of the four kecam members (ResNest50, EfficientNetV2T, EfficientNetV1B4, ECA_NFNetL0)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kecam_ref as ref  # noqa: E402
from tests.test_gpu_resnet_rs import _images  # noqa: E402
from tests.test_gpu_ops import check, dev, h  # noqa: E402


def test_radix_combine(report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops
    g = torch.Generator().manual_seed(9)
    x = h(torch.randn(2, 5, 6, 128, generator=g))
    s = h(torch.rand(2, 128, generator=g))
    refv = (x * s[:, None, None, :]).reshape(2, 5, 6, 2, 64).sum(3)
    check(report, "radix_combine", ops.radix_combine(dev(x), dev(s), 2), refv)
    from tests.test_gpu_ops import split_gate
    s32 = torch.rand(2, 128, generator=g)                      # weights that are not fp16 numbers, as two planes
    ref2 = (x * s32[:, None, None, :]).reshape(2, 5, 6, 2, 64).sum(3)
    got2 = ops.radix_combine(dev(x), split_gate(s32), 2)
    check(report, "radix_combine split", got2, ref2)
    assert torch.equal(got2.cpu().float(), h(ref2)) or (got2.cpu().float() - h(ref2)).abs().max() <= 2e-3   # correctly rounded up to fp32 sum order


@pytest.mark.parametrize("key", ["resnest50", "efficientnet_v2t", "efficientnet_v1b4", "eca_nfnet_l0"])
def test_kecam_member(key, report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops, zoo
    spec = zoo.MEMBERS[key]
    p = spec.synth(spec.seed)
    x = _images(3, spec.input_hw).to(torch.float16).to(torch.float32)
    ca, cb = [], []
    with torch.no_grad():
        ref.features(key, p, x, collect=ca)
        z_ref = ref.predict_logits(key, p, x)
    m = spec.ctor(p)
    xd = ops.to_device_nhwc8(x)
    m.features(xd, collect=cb)
    z = m.logits(xd).cpu()
    torch.cuda.synchronize()
    assert len(ca) == len(cb)
    worst = 0.0
    for i, (a, b) in enumerate(zip(ca, cb)):
        b = b.float().cpu()
        assert a.shape == b.shape, (i, a.shape, b.shape)
        rel = ((a - b) ** 2).mean().sqrt().item() / a.pow(2).mean().sqrt().item()
        worst = max(worst, rel)
        report(f"[{key}] stage {i} shape {tuple(a.shape)} ref_rms {a.pow(2).mean().sqrt().item():.3f} rel_rms_err {rel:.3e}")
    ze = (z - z_ref).abs().max().item()
    report(f"[{key}] logit max_abs_err={ze:.3e} logit mean={z_ref.mean().item():.3f} std={z_ref.std().item():.3f}")
    assert worst < 6e-3
    assert ze < 6e-3 * max(1.0, z_ref.abs().max().item())   # raw synthetic head; 38-block MBConv net: see DESIGN.md §4


def _compare(report, tag, ca, cb, z, z_ref):
    assert len(ca) == len(cb)
    worst = 0.0
    for i, (a, b) in enumerate(zip(ca, cb)):
        b = b.float().cpu()
        assert a.shape == b.shape, (i, a.shape, b.shape)
        worst = max(worst, ((a - b) ** 2).mean().sqrt().item() / a.pow(2).mean().sqrt().item())
    ze = (z - z_ref).abs().max().item()
    report(f"[{tag}] worst stage rel_rms_err {worst:.3e} | logit max_abs_err={ze:.3e} logit mean={z_ref.mean().item():.3f}")
    assert worst < 6e-3
    assert ze < 6e-3 * max(1.0, z_ref.abs().max().item())


def test_kecam_legacy_configs_reduced_depth(report):
    """The larger members of the reference's earlier ensembles (main.py:43-56) that are re-configurations of graphs built
    here - ResNest200 (stem 128), ECA_NFNetL2 (features x2), EfficientNetV2M (7 stages, TF-SAME, SE from stage 3 on) - at
    reduced depth: every config-specific path once, without the activation growth that seeded weights give 60+ blocks."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import kecam_models as km, ops
    x = _images(2, 200).to(torch.float16).to(torch.float32)
    xd = ops.to_device_nhwc8(x)

    cfg = dict(km.RESNEST200, num_blocks=(1, 2, 2, 1))
    p = km.resnest_synth_params(1021, cfg=cfg)
    ca, cb = [], []
    with torch.no_grad():
        f = ref.resnest_features(p, x, num_blocks=cfg["num_blocks"], stem_width=cfg["stem_width"], collect=ca)
        z_ref = ref.R.dense(ref.R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])
    m = km.ResNest(p, cfg=cfg)
    m.features(xd, collect=cb)
    _compare(report, "ResNest200 d1221", ca, cb, m.logits(xd).cpu(), z_ref)

    cfg = dict(km.RESNET200D, num_blocks=(1, 2, 2, 1))
    p = km.resnest_synth_params(1027, cfg=cfg)
    ca, cb = [], []
    with torch.no_grad():
        f = ref.resnest_features(p, x, num_blocks=cfg["num_blocks"], attn=None, collect=ca)
        z_ref = ref.R.dense(ref.R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])
    m = km.ResNest(p, cfg=cfg)
    m.features(xd, collect=cb)
    _compare(report, "ResNet200D d1221", ca, cb, m.logits(xd).cpu(), z_ref)

    cfg = dict(km.NFNET_L2, num_blocks=(1, 2, 2, 1))
    p = km.nfnet_synth_params(1025, cfg=cfg)
    ca, cb = [], []
    with torch.no_grad():
        f = ref.nfnet_features(p, x, num_blocks=cfg["num_blocks"], num_features_factor=cfg["num_features_factor"], collect=ca)
        z_ref = ref.R.dense(ref.R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])
    m = km.NormFreeNet(p, cfg=cfg)
    m.features(xd, collect=cb)
    _compare(report, "ECA_NFNetL2 d1221", ca, cb, m.logits(xd).cpu(), z_ref)

    name = "EfficientNetV2M_d"
    small = dict(km.EFFNET["EfficientNetV2M"], depthes=[1, 2, 2, 2, 2, 2, 1])
    km.EFFNET[name] = ref.EFFNET[name] = small
    try:
        p = km.effnet_synth_params(name, 1023)
        ca, cb = [], []
        with torch.no_grad():
            z_ref = ref.predict_logits(name, p, x)
            ref.features(name, p, x, collect=ca)
        m = km.EfficientNet(p, name)
        m.features(xd, collect=cb)
        _compare(report, "EfficientNetV2M d1222221", ca, cb, m.logits(xd).cpu(), z_ref)
    finally:
        del km.EFFNET[name], ref.EFFNET[name]


def test_efficientnet_v2l_full_depth(report):
    """EfficientNetV2L (efficientnet_v2.py:313-325; a member of the earlier ensembles, main.py:43-56) at FULL depth (79 blocks), 2 images.
    With the seeded weights as generated the 6x-expanded hidden tensors of the late stages leave fp16 range (trunk rms 6e3 at the end; a
    trained checkpoint does not do that), so the residual branches' projection kernels are halved here - in the ONE dict both the oracle
    and the product read - which keeps the whole 79-block graph comparable."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops, zoo
    spec = zoo.MEMBERS["efficientnet_v2l"]
    p = spec.synth(spec.seed)
    for k in list(p):
        if k.endswith("MB_pw_conv/kernel") or k.endswith("fu_conv/kernel"):
            p[k] = p[k] * 0.5
    x = _images(2, 200).to(torch.float16).to(torch.float32)
    ca, cb = [], []
    with torch.no_grad():
        z_ref = ref.predict_logits("EfficientNetV2L", p, x)
        ref.features("EfficientNetV2L", p, x, collect=ca)
    m = spec.ctor(p)
    xd = ops.to_device_nhwc8(x)
    m.features(xd, collect=cb)
    z = m.logits(xd).cpu()
    torch.cuda.synchronize()
    assert torch.isfinite(z).all()
    _compare(report, "EfficientNetV2L full", ca, cb, z, z_ref)


def test_hornet_reduced_depth(report):
    """HorNetBase (hornet.py:196-198) at depths (1,1,2,1): recursive gated convolution with 2..5 orders, depthwise 7x7 on
    the 2C - C/2^(n-1) gate channels, channel-slice products, folded layer scales, LN -> Dense head."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import hornet, ops
    from oracle import hornet_ref
    cfg = dict(hornet.CONFIGS["hornet_base"], num_blocks=(1, 1, 2, 1))
    p = hornet.synth_params(cfg, 1028)
    x = _images(2, 200).to(torch.float16).to(torch.float32)
    ca, cb = [], []
    with torch.no_grad():
        hornet_ref.forward_features(p, x, cfg, collect=ca)
        z_ref = hornet_ref.forward_logits(p, x, cfg)
    m = hornet.HorNet(p, **cfg)
    xd = ops.to_device_nhwc8(x)
    m.features(xd, collect=cb)
    _compare(report, "HorNetBase d1121", ca, cb, m.logits(xd).cpu(), z_ref)
