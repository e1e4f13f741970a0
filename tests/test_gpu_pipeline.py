"""GPU parity of the input pipeline: JPEG bytes -> uint8 RGB (bit-exact vs libjpeg-turbo / the oracle) ->
bicubic resize + /255 -> fp16 (vs the oracle restatement of tf.image.resize), plus the TTA ops."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from oracle import jpeg_ref  # noqa: E402
from oracle import ops_ref as R  # noqa: E402
from tests.test_oracle_jpeg import GOLD, _progressive, _rgb_coded, _variants  # noqa: E402
from tools.make_synth import synth_jpeg  # noqa: E402


def _pil(b):
    return np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))


def _raws():
    raws = [synth_jpeg(i) for i in range(12)] + [synth_jpeg(49)] + list(_variants().values()) + list(_progressive().values()) + list(_rgb_coded().values())
    for n in ("dog_cat", "cat", "dog"):
        raws.append(open(os.path.join(GOLD, f"ref_{n}.jpg"), "rb").read())
    return raws


def test_decode_bit_exact(report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    raws = _raws()
    batch = pipeline.decode_jpegs(raws)
    torch.cuda.synchronize()
    rgb = batch.rgb.cpu().numpy()
    worst = 0
    for i, raw in enumerate(raws):
        h, w = batch.sizes_host[i]
        ref = _pil(raw)
        assert ref.shape == (h, w, 3)
        d = np.abs(rgb[i, :h, :w].astype(int) - ref.astype(int)).max()
        worst = max(worst, d)
        assert d == 0, f"image {i}: max diff {d}"
        if i < 3:
            assert np.array_equal(rgb[i, :h, :w], jpeg_ref.decode_rgb(raw))
    report(f"[pipeline] decode_jpegs: {len(raws)} images bit-exact vs libjpeg-turbo (max diff {worst})")


@pytest.mark.parametrize("out_hw", [(200, 200), (224, 224)])
def test_resize_normalize(out_hw, report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    raws = [synth_jpeg(i) for i in (0, 1, 2, 49)] + [open(os.path.join(GOLD, "ref_cat.jpg"), "rb").read()]
    batch = pipeline.decode_jpegs(raws)
    got = batch.resized(*out_hw).float().cpu()
    torch.cuda.synchronize()
    assert got.shape == (len(raws), out_hw[0], out_hw[1], 8)
    assert got[..., 3:].abs().max().item() == 0.0
    worst = 0.0
    for i, raw in enumerate(raws):
        ref = R.decode_resize_normalize(_pil(raw), *out_hw)
        ref16 = ref.to(torch.float16).float()
        err = (got[i, ..., :3] - ref16).abs().max().item()
        worst = max(worst, err)
        # identical fp32 arithmetic -> identical fp16 values (allow one fp16 ulp at |x| <= ~1.1: 9.8e-4)
        assert err <= 1e-3, (i, err)
        if _pil(raw).shape[:2] == out_hw:   # scale 1: the resize is the identity
            assert torch.equal(got[i, ..., :3], (torch.from_numpy(_pil(raw).copy()).float() / 255.0).to(torch.float16).float())
    report(f"[pipeline] resize {out_hw}: max |hip - oracle| = {worst:.3e} (fp16 values)")


def test_tta_ops(report):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    g = torch.Generator().manual_seed(0)
    x = torch.rand((4, 10, 12, 8), generator=g).to(torch.float16)
    x[..., 3:] = 0
    got = pipeline.apply_augment(x.cuda(), [1, 0, 1, 0], [0, 1, 1, 0], [0, 0, 1, 1]).float().cpu()
    xf = x.float()
    ref = [xf[0].flip(1), xf[1].flip(0), xf[2].flip(0).flip(1), xf[3]]
    for i in (2, 3):  # tf.image.rgb_to_grayscale + grayscale_to_rgb (augment.py:142-146)
        gch = (0.2989 * ref[i][..., 0] + 0.5870 * ref[i][..., 1] + 0.1140 * ref[i][..., 2])
        ref[i] = torch.cat([gch[..., None].expand(-1, -1, 3), ref[i][..., 3:]], -1)
    for i in range(4):
        assert (got[i] - ref[i]).abs().max().item() <= 1e-3, i
    report("[pipeline] tta flips/gray ok")


def test_tiny_gray_and_odd_images_through_decode_and_resize(report):
    """1x1, 3x5, 17x13 colour and grayscale files in one batch with a 200x200 one: decode bit-exact, resize vs the oracle."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    from tools.make_synth import synth_pixels
    px = synth_pixels(11)
    raws = []
    for hw, mode, kw in (((1, 1), "RGB", dict(quality=90)), ((3, 5), "RGB", dict(quality=75, subsampling=2)),
                         ((17, 13), "L", dict(quality=80)), ((9, 31), "RGB", dict(quality=60, subsampling=1, progressive=True)),
                         # chroma planes <= 2 samples wide: libjpeg-turbo replicates instead of fancy-upsampling (jdsample.c)
                         ((20, 4), "RGB", dict(quality=95, subsampling=1)), ((16, 3), "RGB", dict(quality=77, subsampling=2)),
                         ((9, 2), "RGB", dict(quality=30, subsampling=2)), ((8, 1), "RGB", dict(quality=95, subsampling=2)),
                         ((200, 200), "RGB", dict(quality=85))):
        b = io.BytesIO()
        Image.fromarray(px[:hw[0], :hw[1]]).convert(mode).save(b, format="JPEG", **kw)
        raws.append(b.getvalue())
    batch = pipeline.decode_jpegs(raws)
    torch.cuda.synchronize()
    rgb = batch.rgb.cpu().numpy()
    for i, raw in enumerate(raws):
        h, w = batch.sizes_host[i]
        assert np.array_equal(rgb[i, :h, :w], _pil(raw)), i
    got = batch.resized(200, 200).float().cpu()
    worst = 0.0
    for i, raw in enumerate(raws):
        ref16 = R.decode_resize_normalize(_pil(raw), 200, 200).to(torch.float16).float()
        worst = max(worst, (got[i, ..., :3] - ref16).abs().max().item())
    report(f"[pipeline] tiny/gray/odd images: decode bit-exact, resize max |hip - oracle| = {worst:.3e}")
    assert worst <= 1e-3
