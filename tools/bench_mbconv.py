"""Fused expand + depthwise (vip_mbconv_expand_dw_f16) vs the two launches, on EfficientNet-B4 / V2-T block shapes (B = 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
CASES = [("b4.s2.0", 100, 24, 144, 3, 2), ("b4.s2.1", 50, 32, 192, 3, 1), ("b4.s3.0", 50, 32, 192, 5, 2), ("b4.s3.1", 25, 56, 336, 5, 1),
         ("b4.s4.0", 25, 56, 336, 3, 2), ("b4.s4.1", 13, 112, 672, 3, 1), ("v2t.s4.0", 28, 48, 192, 3, 2), ("v2t.s4.1", 14, 104, 416, 3, 1),
         ("v2t.s5.0", 14, 104, 624, 3, 1), ("v2t.s5.1", 14, 128, 768, 3, 1), ("v2t.s6.0", 14, 128, 768, 3, 2)]
B = int(os.environ.get("B", "256"))


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, H, Cin, Ce, k, s in CASES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, H, H, Cin), generator=g).to("cuda", torch.float16)
    cw = ops.make_conv_weight(torch.randn(1, 1, Cin, Ce, generator=g) / Cin ** 0.5, torch.zeros(Ce), hilo=H >= 25)
    wd = ops.make_dw_weight(torch.randn((k, k, Ce, 1), generator=g) / k)
    bd = torch.zeros(Ce, device="cuda")
    p = k // 2
    pad = (p, p, p, p) if s == 1 else ((k - 1) // 2 if H % 2 else (k - 2) // 2, k // 2, (k - 1) // 2 if H % 2 else (k - 2) // 2, k // 2)
    os.environ["VIP_MBCONV_FUSED"] = "1"
    ms1 = timeit(lambda: ops.mbconv_expand_dw(x, cw, wd, bd, k, s, pad, act="silu"))
    del os.environ["VIP_MBCONV_FUSED"]
    ms2 = timeit(lambda: ops.mbconv_expand_dw(x, cw, wd, bd, k, s, pad, act="silu"))
    Ho = (H + pad[0] + pad[1] - k) // s + 1
    nb = 2.0 * (x.numel() + B * Ho * Ho * Ce)
    print(f"{name:9s} {H:3d}x{H:<3d} Cin{Cin:4d} Ce{Ce:4d} k{k} s{s}  fused {ms1*1e3:7.1f} us ({nb/ms1/1e6:5.0f} GB/s)   two launches {ms2*1e3:7.1f} us   x{ms2/ms1:.2f}", flush=True)
