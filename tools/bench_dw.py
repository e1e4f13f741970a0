"""Micro-benchmark of the depthwise conv kernel on the ConvNeXt-T / EfficientNet / GCViT layer shapes (B=256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
CASES = [("cnx.s0", 99, 96, 7, 1, None), ("cnx.s1", 49, 192, 7, 1, None), ("cnx.s2", 24, 384, 7, 1, None),
         ("cnx.s3", 12, 768, 7, 1, None), ("gcv.l0", 112, 64, 3, 1, "gelu"), ("eff.56", 56, 192, 3, 1, "silu"),
         ("eff.28", 28, 336, 5, 1, "silu"), ("eff.14a", 14, 960, 5, 1, "silu"), ("eff.14b", 14, 672, 3, 1, "silu"),
         ("eff.13", 13, 768, 3, 1, "silu"), ("eff.7a", 7, 1248, 3, 1, "silu"), ("eff.7b", 7, 1632, 5, 1, "silu"),
         ("eff.s2", 112, 144, 3, 2, "silu")]
B = int(os.environ.get("B", "256"))
for name, H, C, k, s, act in CASES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, H, H, C), generator=g).to("cuda", torch.float16)
    w = ops.make_dw_weight(torch.randn((k, k, C, 1), generator=g) / k)
    b = torch.zeros(C, device="cuda")
    p = k // 2
    y = ops.dwconv2d(x, w, b, k, s, (p, p, p, p), act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.dwconv2d(x, w, b, k, s, (p, p, p, p), act=act)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = 2.0 * (x.numel() + y.numel())
    fl = 2.0 * y.numel() * k * k
    print(f"{name:10s} H{H:3d} C{C:4d} k{k} s{s} {ms*1e3:8.1f} us {by/ms/1e6:7.1f} GB/s {fl/ms/1e9:6.2f} TFLOP/s", flush=True)
