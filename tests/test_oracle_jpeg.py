"""CPU: (1) pin the JPEG oracle to libjpeg-turbo (Pillow) on the reference's own fixture images and on the
synthetic set; (2) check the product's HOST entropy decoder (C++ in libvipcup_hip.so, no GPU needed)
against the oracle's coefficients."""
import io
import json
import os

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_ref
from tools.make_synth import synth_jpeg, synth_pixels

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _variants():
    """extra encodings of one synthetic image: 4:2:2, grayscale, restart markers, odd sizes"""
    px = synth_pixels(7)
    out = {}
    for name, kw in {"s422": dict(subsampling=1), "q30_420": dict(quality=30, subsampling=2),
                     "q100_444": dict(quality=100, subsampling=0)}.items():
        b = io.BytesIO()
        Image.fromarray(px).save(b, format="JPEG", **{"quality": 85, **kw})
        out[name] = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(px).convert("L").save(b, format="JPEG", quality=80)
    out["gray"] = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(px[:173, :131]).save(b, format="JPEG", quality=75, subsampling=2)
    out["odd_173x131_420"] = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(px[:57, :200]).save(b, format="JPEG", quality=75, subsampling=1, restart_marker_blocks=3)
    out["restart_422"] = b.getvalue()
    return out


def _rgb_coded():
    """files whose three components ARE R, G, B (Adobe APP14, transform 0) - libjpeg applies no colour conversion"""
    px = synth_pixels(9)
    out = {}
    for name, hw, q in (("rgb_64", (64, 64), 90), ("rgb_33x21", (33, 21), 60), ("rgb_5x3", (5, 3), 75)):
        b = io.BytesIO()
        Image.fromarray(px[:hw[0], :hw[1]]).save(b, format="JPEG", quality=q, keep_rgb=True)
        out[name] = b.getvalue()
        assert b"Adobe" in out[name]
    return out


def _progressive():
    """progressive (SOF2) encodings: libjpeg's default scan script has DC first/refine, AC first/refine scans,
    interleaved DC scans and per-component AC scans"""
    px = synth_pixels(5)
    out = {}
    for name, kw in {"p420": dict(quality=80, subsampling=2), "p444_q92": dict(quality=92, subsampling=0),
                     "p422_q60": dict(quality=60, subsampling=1), "p420_q20": dict(quality=20, subsampling=2)}.items():
        b = io.BytesIO()
        Image.fromarray(px).save(b, format="JPEG", progressive=True, **kw)
        out[name] = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(px[:173, :131]).save(b, format="JPEG", quality=75, subsampling=2, progressive=True)
    out["p_odd_173x131"] = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(px).convert("L").save(b, format="JPEG", quality=80, progressive=True)
    out["p_gray"] = b.getvalue()
    b = io.BytesIO()
    Image.fromarray(px[:90, :200]).save(b, format="JPEG", quality=75, subsampling=2, progressive=True,
                                        restart_marker_blocks=2)
    out["p_restart"] = b.getvalue()
    for name, raw in out.items():
        assert b"\xff\xc2" in raw, name
    return out


def _pil(b):
    return np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))


@pytest.mark.parametrize("name", ["dog_cat", "cat", "dog"])
def test_oracle_matches_pillow_on_reference_fixtures(name):
    """the three JPEGs embedded in the reference (kecam test_images.py:6-15)"""
    raw = open(os.path.join(GOLD, f"ref_{name}.jpg"), "rb").read()
    stats = json.load(open(os.path.join(GOLD, "ref_jpeg_pillow_stats.json")))[name]
    got = jpeg_ref.decode_rgb(raw)
    assert list(got.shape) == stats["shape"]
    assert int(got.astype(np.int64).sum()) == stats["sum"]          # committed Pillow statistic
    assert [int(got[r].astype(np.int64).sum()) for r in (0, 100, 255, 511)] == stats["crc_rows"]
    assert np.array_equal(got, _pil(raw))                            # and the live library


def test_oracle_matches_pillow_on_synthetic_set():
    for i in list(range(8)) + [49]:
        raw = synth_jpeg(i)
        assert np.array_equal(jpeg_ref.decode_rgb(raw), _pil(raw)), i
    for name, raw in _variants().items():
        assert np.array_equal(jpeg_ref.decode_rgb(raw), _pil(raw)), name


def test_oracle_matches_pillow_on_progressive_streams():
    for name, raw in _progressive().items():
        assert np.array_equal(jpeg_ref.decode_rgb(raw), _pil(raw)), name


def test_rgb_coded_files_skip_the_colour_conversion():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    raws = _rgb_coded()
    for name, raw in raws.items():
        assert np.array_equal(jpeg_ref.decode_rgb(raw), _pil(raw)), name
    desc, _ = pipeline.entropy_decode(list(raws.values()) + [synth_jpeg(0)])
    assert [d.rgb_coded for d in desc] == [1, 1, 1, 0]


def test_host_entropy_decoder_matches_oracle():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    raws = [synth_jpeg(i) for i in (0, 1, 49)] + list(_variants().values()) + list(_progressive().values())
    raws.append(open(os.path.join(GOLD, "ref_cat.jpg"), "rb").read())
    desc, coef = pipeline.entropy_decode(raws, threads=3)
    for i, raw in enumerate(raws):
        P = jpeg_ref.parse(raw)
        ref_coefs, _ = jpeg_ref.entropy_decode(P)
        d = desc[i]
        assert (d.height, d.width, d.ncomp) == (P["h"], P["w"], len(P["comp"]))
        for c, rc in enumerate(ref_coefs):
            n = rc.size
            got = coef[d.coef_off[c]:d.coef_off[c] + n].reshape(rc.shape)
            assert (d.blocks_h[c], d.blocks_w[c]) == rc.shape[:2]
            assert np.array_equal(got.astype(np.int64), rc), (i, c)
            assert np.array_equal(np.array(d.qt[c][:], dtype=np.int64), P["qt"][P["comp"][c]["tq"]])


def test_unsupported_streams_are_rejected():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import _abi, pipeline
    raw = bytearray(synth_jpeg(3))
    i = raw.find(b"\xff\xc0")
    raw[i + 1] = 0xC9                                     # arithmetic-coded frame: not a Huffman stream
    with pytest.raises(_abi.VipError):
        pipeline.entropy_decode([bytes(raw)])
    with pytest.raises(_abi.VipError):
        pipeline.entropy_decode([b"not a jpeg at all"])
    raw = bytearray(synth_jpeg(3))
    i = raw.find(b"\xff\xc0")
    raw[i + 5:i + 9] = b"\xff\xff\xff\xff"             # the frame header claims 65535 x 65535
    with pytest.raises(_abi.VipError, match="VIP_MAX_JPEG_PIXELS"):
        pipeline.entropy_decode([bytes(raw)])
    raw = bytearray(synth_jpeg(3))
    i = raw.find(b"\xff\xda")
    raw[i + 6] = 0x70                                     # SOS selects DC table 7
    with pytest.raises(_abi.VipError):
        pipeline.entropy_decode([bytes(raw)])


def test_host_decoder_matches_oracle_and_pillow_on_random_small_images():
    """40 random encodings (size 1..48, quality 5..100, 4:4:4 / 4:2:2 / 4:2:0 / gray, baseline / progressive / optimised
    tables / restart intervals): host coefficients == oracle coefficients, oracle pixels == libjpeg-turbo pixels."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    rng = np.random.default_rng(20221)
    raws = []
    for i in range(40):
        h, w = int(rng.integers(1, 49)), int(rng.integers(1, 49))
        smooth = rng.integers(0, 256, size=(max(1, h // 6) + 1, max(1, w // 6) + 1, 3))
        img = np.kron(smooth, np.ones((6, 6, 1)))[:h, :w] + rng.normal(0, 10, (h, w, 3))
        im = Image.fromarray(np.clip(img, 0, 255).astype(np.uint8))
        kw = dict(quality=int(rng.integers(5, 101)))
        mode = int(rng.integers(0, 4))
        if mode == 3:
            im = im.convert("L")
        else:
            kw["subsampling"] = mode
        if rng.random() < 0.4:
            kw["progressive"] = True
        if rng.random() < 0.3:
            kw["optimize"] = True
        if rng.random() < 0.3:
            kw["restart_marker_blocks"] = int(rng.integers(1, 5))
        b = io.BytesIO()
        im.save(b, format="JPEG", **kw)
        raws.append(b.getvalue())
    desc, coef = pipeline.entropy_decode(raws, threads=2)
    for i, raw in enumerate(raws):
        P = jpeg_ref.parse(raw)
        ref_coefs, _ = jpeg_ref.entropy_decode(P)
        d = desc[i]
        for c, rc in enumerate(ref_coefs):
            got = coef[d.coef_off[c]:d.coef_off[c] + rc.size].reshape(rc.shape)
            assert np.array_equal(got.astype(np.int64), rc), (i, c)
        assert np.array_equal(jpeg_ref.decode_rgb(raw), _pil(raw)), i
