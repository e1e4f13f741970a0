"""Fused MLP (vip_mlp_fused_f16) vs two Dense launches on the ConvNeXt-T stage-0 and GCViT level-0 token shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
for M, C, hid in [(2509056, 96, 384), (802816, 64, 256), (614656, 192, 768)]:
    x = torch.randn((M, C), dtype=torch.float16, device="cuda")
    fc1 = ops.make_dense_weight(torch.randn(C, hid) * 0.1, torch.zeros(hid))
    fc2 = ops.make_dense_weight(torch.randn(hid, C) * 0.05, torch.zeros(C))
    def two():
        return ops.dense(ops.dense(x, fc1, act="gelu"), fc2, residual=x)
    def fused():
        return ops.mlp(x, fc1, fc2, act="gelu", residual=x)
    a, b = two(), fused()
    torch.cuda.synchronize()
    print(f"M={M} C={C} hid={hid}: max|fused - two| = {(a.float() - b.float()).abs().max().item():.3e}")
    for name, fn in (("two dense", two), ("fused", fused)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"   {name:10s} {ms*1e3:8.1f} us  {4.0*M*C*hid/ms/1e9:7.1f} TF  io {6.0*M*C/ms/1e6:7.0f} GB/s")
