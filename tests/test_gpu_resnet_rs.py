"""GPU parity of the ResNet-RS path (BASELINE config 2) against the fp32 CPU oracle."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import resnet_rs_ref as ref  # noqa: E402


def _images(n, size=200):
    from PIL import Image
    from tools.make_synth import synth_jpeg
    out = []
    for i in range(n):
        im = Image.open(io.BytesIO(synth_jpeg(i))).convert("RGB").resize((size, size))
        out.append(torch.from_numpy(np.asarray(im).copy()).float() / 255.0)
    return torch.stack(out)


def _run(depth_args, size, n, report, tag, seed):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops, resnet_rs
    x = _images(n, size)
    x16 = x.to(torch.float16).to(torch.float32)  # both sides see the same fp16-rounded pixels
    p = resnet_rs.synth_params(50, seed=seed, block_args=depth_args)
    with torch.no_grad():
        f_ref = ref.forward_features(p, x16, block_args=depth_args)
        z_ref = ref.forward_logits(p, x16, block_args=depth_args)
    m = resnet_rs.ResNetRS(p, depth=50, block_args=depth_args)
    xd = ops.to_device_nhwc8(x16)
    f = m.features(xd).float().cpu()
    z = m.logits(xd).cpu()
    torch.cuda.synchronize()
    fe = (f - f_ref).abs().max().item() / (f_ref.abs().max().item() + 1e-9)
    frms = ((f - f_ref) ** 2).mean().sqrt().item() / (f_ref.pow(2).mean().sqrt().item() + 1e-9)
    ze = (z - z_ref).abs().max().item()
    report(f"[resnet_rs {tag}] feat rel_max_err={fe:.3e} rel_rms_err={frms:.3e} | logit max_abs_err={ze:.3e} "
           f"logit mean={z_ref.mean().item():.3f} std={z_ref.std().item():.3f} feat_rms={f_ref.pow(2).mean().sqrt().item():.3f}")
    return fe, frms, ze, z_ref


def test_resnet_rs_tiny(report):
    """BLOCK_ARGS (1,1,1,1) at 64x64 — every layer type once."""
    fe, frms, ze, z_ref = _run([(64, 1), (128, 1), (256, 1), (512, 1)], 64, 4, report, "tiny", 1006)
    assert frms < 5e-3 and fe < 3e-2


def test_resnet_rs101_full(report):
    """ResNet-RS-101 (block_args.py:8-13; the 200-layer variant of the earlier ensembles is the same code path)."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import resnet_rs
    fe, frms, ze, z_ref = _run(resnet_rs.BLOCK_ARGS[101], 200, 3, report, "rs101", 1016)
    assert frms < 4e-3
    assert ze < 3e-3 * max(1.0, z_ref.abs().max().item())


def test_resnet_rs200_full(report):
    """ResNet-RS-200 (block_args.py: 3, 24, 36, 3 bottlenecks), a member of the earlier ensembles (main.py:43-56): 2 images, full depth."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import resnet_rs
    fe, frms, ze, z_ref = _run(resnet_rs.BLOCK_ARGS[200], 200, 2, report, "rs200", 1026)
    assert frms < 6e-3
    assert ze < 4e-3 * max(1.0, z_ref.abs().max().item())


def test_resnet_rs50_full(report):
    """Full ResNet-RS-50 at 200x200 on 8 synthetic images: logits vs the fp32 oracle."""
    fe, frms, ze, z_ref = _run(None, 200, 8, report, "rs50", 1006)
    assert frms < 3e-3
    # raw (uncalibrated) synthetic head: compare relative to the logit scale
    assert ze < 2e-3 * max(1.0, z_ref.abs().max().item())
