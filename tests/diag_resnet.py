"""Diagnostic (not collected by pytest): per-block relative error of the HIP ResNet-RS-50 vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_resnet_rs import _images
from oracle import resnet_rs_ref as ref
import vipcup_amd
from vipcup_amd import ops, resnet_rs

x = _images(4, 200).to(torch.float16).to(torch.float32)
p = resnet_rs.synth_params(50, seed=1006)
# oracle A: exact fp32 weights ; oracle B: weights rounded the way the product rounds them is covered by fold -> skip
ca, cb = [], []
with torch.no_grad():
    ref.forward_features(p, x, collect=ca)
m = resnet_rs.ResNetRS(p, depth=50)
m.features(ops.to_device_nhwc8(x), collect=cb)
torch.cuda.synchronize()
for i, (a, b) in enumerate(zip(ca, cb)):
    b = b.float().cpu()
    rms = ((a - b) ** 2).mean().sqrt().item() / (a.pow(2).mean().sqrt().item() + 1e-12)
    mx = (a - b).abs().max().item() / (a.abs().max().item() + 1e-12)
    print(f"stage {i:2d} shape {tuple(a.shape)} ref_rms {a.pow(2).mean().sqrt().item():9.3f} absmax {a.abs().max().item():9.2f} rel_rms_err {rms:.3e} rel_max_err {mx:.3e}")
