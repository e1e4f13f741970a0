"""Shared helpers of the GPU parity tests: the oracle side (Pillow/libjpeg-turbo decode -> oracle resize -> oracle graph) and the
tolerance table.  Test infrastructure (imports oracle/)."""
import importlib
import io

import numpy as np
import torch
from PIL import Image

from oracle import ops_ref as R

# BASELINE.json north_star: |z_hip - z_ref| <= 1e-3 on the sigmoid logit - every member's and the ensemble score's.
TOL_NORTH_STAR = 1e-3
# Two precision modes, two sets of bounds:
#
# * STRICT (fp32 storage, f32 MFMA; tests/test_gpu_strict.py, the strict variants below): TOL_NORTH_STAR is asserted as it stands, on every
#   member's calibrated logit, on logit(ensemble mean), on the synthetic set and on the photographs, absolute.  Measured: <= 5e-5.
#
# * FAST (fp16 storage, the throughput path bench.py's `value` is quoted on): the member logits sit at the fp16-STORAGE floor of each
#   graph (tests/diag_gpu_vs_emul.py: a CPU emulation with fp32 arithmetic and only the operator outputs rounded to fp16 reproduces the
#   GPU's error for every member), amplified by the calibrated synthetic heads (tests/gen_synth_heads.py: |w||f| = 12 ... 78).  The two
#   members that meet the north star are held to it; the other five are NOT within 1e-3 in this mode - their ceilings below are their
#   measured floors (max over <= 128 images), stated so that a regression shows, not to claim the tolerance.
MEMBER_CEILING = {
    "eca_nfnet_l0": TOL_NORTH_STAR, "convnext_tiny_in22k": TOL_NORTH_STAR,        # measured 7.0e-4 / 8.1e-4 on 64 images
    "resnet_rs50": 2.5e-3, "resnest50": 4.0e-3, "gcvit_tiny": 7.0e-3, "efficientnet_v2t": 1.0e-2, "efficientnet_v1b4": 1.8e-2,
    "vit_tiny_patch16_224": 4.0e-3, "vit_small_patch16_224": 4.0e-3,
}
# fast mode, photographs (off the bias-calibration distribution; logits reach +-20): ABSOLUTE |dz| ceilings, measured
# (profiles/r02_parity_gpu.log): the relative form max(1, |z|/3) used to hide these
MEMBER_CEILING_PHOTO_ABS = {
    "eca_nfnet_l0": 3.0e-3, "convnext_tiny_in22k": 2.0e-3, "resnet_rs50": 6.0e-3, "resnest50": 8.0e-3, "gcvit_tiny": 1.2e-2,
    "efficientnet_v2t": 3.5e-2, "efficientnet_v1b4": 3.0e-2, "vit_tiny_patch16_224": 8.0e-3, "vit_small_patch16_224": 8.0e-3,
}
TOL_ENSEMBLE_PROB = 1e-3   # fast mode: the ensemble-mean PROBABILITY main.py thresholds at 0.487 (uncorrelated member errors average down)
# fast mode: logit(ensemble mean) - the north-star axis; d logit = dp / (p (1 - p)) ~ 4 dp near the threshold.  Measured 1.4e-3 (7 / 8
# members), 4.8e-3 (config 4's four): fast mode does NOT meet 1e-3 on this axis either; strict mode does.
FAST_ENSEMBLE_LOGIT_CEILING = {"ensemble": 2.5e-3, "ensemble8": 2.5e-3, "ensemble4": 6.0e-3}

_CACHE = {}


def decode_pixels(raws):
    return [np.asarray(Image.open(io.BytesIO(r)).convert("RGB")) for r in raws]


def e2e_image_ids(n: int):
    """seeds of the synthetic JPEG set shared by the CLI test and the workload tests (one oracle pass per member and session):
    n - 1 images of 200x200 and one 256x192 (the resize branch), no duplicates"""
    return list(range(100, 100 + n - 1)) + [149 if n <= 50 else 49]


def oracle_logits(key: str, set_name: str, raws) -> np.ndarray:
    """fp32 oracle logits of member `key` on the JPEG byte strings `raws` (cached per (member, set_name))."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import zoo
    ck = (key, set_name, len(raws))
    if ck not in _CACHE:
        spec = zoo.MEMBERS[key]
        pix = decode_pixels(raws)
        x = torch.stack([R.decode_resize_normalize(p, spec.input_hw, spec.input_hw) for p in pix])
        ref = importlib.import_module(f"oracle.{spec.oracle}")
        params = zoo.build_params(key)
        out = []
        with torch.no_grad():
            for i in range(0, len(raws), 16):
                out.append(ref.predict_logits(key, params, x[i:i + 16]).numpy()[:, 0])
        _CACHE[ck] = np.concatenate(out)
    return _CACHE[ck]


_MODELS = {}


def gpu_member(key: str, precision: str = "fast"):
    """(spec, model) built once per session and precision mode (fast: construction includes the bias-calibration pass)"""
    from vipcup_amd import zoo
    if (key, precision) not in _MODELS:
        _MODELS[(key, precision)] = zoo.build_member(key, precision=precision)
    return _MODELS[(key, precision)]


def logit(p):
    p = np.clip(p, 1e-7, 1 - 1e-7)
    return np.log(p / (1 - p))


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def real_photo_tiles():
    """12 JPEG byte strings that are NOT from tools/make_synth: 200x200 tiles of the three photographs the reference embeds
    (tests/golden/ref_*.jpg, 512x512), re-encoded at quality 75 / 90, 4:2:0 / 4:4:4."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raws = []
    k = 0
    for name in ("ref_cat.jpg", "ref_dog.jpg", "ref_dog_cat.jpg"):
        img = Image.open(os.path.join(here, name)).convert("RGB")
        for (x0, y0) in ((20, 30), (290, 40), (60, 300), (300, 290)):
            buf = io.BytesIO()
            img.crop((x0, y0, x0 + 200, y0 + 200)).save(buf, format="JPEG", quality=(75, 90)[k % 2], subsampling=(2, 0)[(k // 2) % 2])
            raws.append(buf.getvalue())
            k += 1
    return raws
