"""LayerNorm micro-benchmark on the ensemble's token shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
for M, C in [(2509056, 96), (614656, 192), (147456, 384), (802816, 64), (200704, 128), (50176, 256), (12544, 512)]:
    x = torch.randn((M, C), dtype=torch.float16, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.layernorm(x, g, b, 1e-6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.layernorm(x, g, b, 1e-6)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M={M:8d} C={C:4d} {ms*1e3:8.1f} us {4.0*M*C/ms/1e6:7.0f} GB/s")
