"""Diagnostic (not a pytest file): which fp16 roundings of a member's HIP graph matter for its calibrated logit.
    python tests/diag_sources2.py <member> [n_images] [tagset1 tagset2 ...]      (tagset = comma-joined emul_ops tags, '+' = none)
Rows: exact weights (w32) vs fp16 weights (w16, bias-corrected), fp32 activations (a32) vs rounded (a16), and a16 with the
listed operator classes exempt from rounding.  Error = emulated logit - oracle logit over n synthetic images."""
import importlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import vipcup_amd  # noqa
from vipcup_amd import zoo, gcvit, resnet_rs, tfimm_models as tm, kecam_models as km, ops
from oracle import ops_ref as R
from tests import emul_ops
from tools.make_synth import synth_jpeg

key = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
spec = zoo.MEMBERS[key]
pix = [np.asarray(Image.open(io.BytesIO(synth_jpeg(100 + i))).convert("RGB")) for i in range(n)]
x = torch.stack([R.decode_resize_normalize(p, spec.input_hw, spec.input_hw) for p in pix]).half().float()
params = zoo.build_params(key)
ref = importlib.import_module(f"oracle.{spec.oracle}")
CTORS = {"gcvit_tiny": lambda p: gcvit.GCViTTiny(p, device="cpu"),
         "efficientnet_v1b4": lambda p: km.EfficientNet(p, "EfficientNetV1B4", device="cpu"),
         "efficientnet_v2t": lambda p: km.EfficientNet(p, "EfficientNetV2T", device="cpu"),
         "resnest50": lambda p: km.ResNest(p, device="cpu"),
         "eca_nfnet_l0": lambda p: km.NormFreeNet(p, device="cpu"),
         "resnet_rs50": lambda p: resnet_rs.ResNetRS50(p, device="cpu"),
         "convnext_tiny_in22k": lambda p: tm.ConvNeXt(p, tm.CONVNEXT_CONFIGS["convnext_tiny_in22k"], device="cpu"),
         "vit_small_patch16_224": lambda p: tm.ViT(p, tm.VIT_CONFIGS["vit_small_patch16_224"], device="cpu"),
         "vit_tiny_patch16_224": lambda p: tm.ViT(p, tm.VIT_CONFIGS["vit_tiny_patch16_224"], device="cpu")}



def stt(d):
    return f"rms {np.sqrt((d**2).mean()):.2e} mean {d.mean():+.2e} std {d.std():.2e} max {np.abs(d).max():.2e}"
from vipcup_amd import pipeline
with torch.no_grad():
    z = ref.predict_logits(key, params, x)[:, 0]
    x8 = emul_ops.to_device_nhwc8(x)
    calraw = [np.asarray(Image.open(io.BytesIO(synth_jpeg(9000 + i))).convert("RGB")) for i in range(16)]
    cal = emul_ops.to_device_nhwc8(torch.stack([R.decode_resize_normalize(p, spec.input_hw, spec.input_hw) for p in calraw]).half().float())
    for off in ("0", "1"):
        os.environ["VIP_OFFSET_CALIBRATION"] = off
        ops.KEEP_ROUNDING_ERROR = True
        with emul_ops.patched(round_act=True):
            m = zoo.calibrate(CTORS[key](params), cal)
            ze = m.logits(x8)[:, 0].float()
        print(key, "offset calib", off, getattr(m, "offset_calibration", None), stt((ze - z).numpy()), flush=True)
