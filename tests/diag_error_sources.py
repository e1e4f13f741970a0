"""Diagnostic (not a pytest file): separates the fp16-weight and fp16-activation contributions to the logit
error of each member, on the CPU, with the calibrated heads."""
import importlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import vipcup_amd
from vipcup_amd import zoo
from oracle import ops_ref as R
from tests import emul_ops
from tools.make_synth import synth_jpeg

members = sys.argv[1:] or zoo.ENSEMBLE
idx = list(range(100, 108))
pix = [np.asarray(Image.open(io.BytesIO(synth_jpeg(i))).convert("RGB")) for i in idx]
for key in members:
    spec = zoo.MEMBERS[key]
    x = torch.stack([R.decode_resize_normalize(p, spec.input_hw, spec.input_hw) for p in pix]).half().float()
    params = zoo.build_params(key)
    ref = importlib.import_module(f"oracle.{spec.oracle}")
    with torch.no_grad():
        z = ref.predict_logits(key, params, x)[:, 0]
        x8 = emul_ops.to_device_nhwc8(x)
        with emul_ops.patched(round_act=False):
            zw = spec.ctor({k: v for k, v in params.items()}).logits(x8)[:, 0] if False else None
        res = {}
        for ra in (False, True):
            with emul_ops.patched(round_act=ra):
                import inspect
                kw = {"device": "cpu"}
                m = spec.ctor.__call__(params) if False else None
            res[ra] = None
    print(key)
