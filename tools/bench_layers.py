"""Per-layer micro-benchmark of the conv/GEMM kernel on the ResNet-RS-50 @200x200 B=256 layer shapes
(and any extra shapes given as B,H,W,Cin,Cout,k,stride,groups).  Prints TFLOP/s and algorithmic GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import ops  # noqa: E402

RS50 = [
    # name, H, Cin, Cout, k, stride
    ("stem1", 200, 8, 32, 3, 2), ("stem2", 100, 32, 32, 3, 1), ("stem3", 100, 32, 64, 3, 1), ("stem4", 100, 64, 64, 3, 2),
    ("c2.c1a", 50, 64, 64, 1, 1), ("c2.c2", 50, 64, 64, 3, 1), ("c2.c3", 50, 64, 256, 1, 1), ("c2.c1", 50, 256, 64, 1, 1),
    ("c3.c1a", 50, 256, 128, 1, 1), ("c3.c2s", 50, 128, 128, 3, 2), ("c3.c3", 25, 128, 512, 1, 1), ("c3.c1", 25, 512, 128, 1, 1),
    ("c3.c2", 25, 128, 128, 3, 1), ("c3.proj", 25, 256, 512, 1, 1),
    ("c4.c1a", 25, 512, 256, 1, 1), ("c4.c2s", 25, 256, 256, 3, 2), ("c4.c3", 13, 256, 1024, 1, 1), ("c4.c1", 13, 1024, 256, 1, 1),
    ("c4.c2", 13, 256, 256, 3, 1), ("c4.proj", 13, 512, 1024, 1, 1),
    ("c5.c1a", 13, 1024, 512, 1, 1), ("c5.c2s", 13, 512, 512, 3, 2), ("c5.c3", 7, 512, 2048, 1, 1), ("c5.c1", 7, 2048, 512, 1, 1),
    ("c5.c2", 7, 512, 512, 3, 1), ("c5.proj", 7, 1024, 2048, 1, 1),
]


def bench(name, B, H, Cin, Cout, k, s, groups=1, iters=20):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, H, H, Cin), generator=g).to("cuda", torch.float16)
    w = torch.randn((k, k, Cin // groups, Cout), generator=g) * 0.05
    cw = ops.make_conv_weight(w, torch.zeros(Cout), groups=groups)
    p = k // 2
    pad = (p, p, p, p)
    y = ops.conv2d(x, cw, stride=s, pad=pad, act="relu")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, cw, stride=s, pad=pad, act="relu", out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = y.shape[0] * y.shape[1] * y.shape[2]
    fl = 2.0 * M * Cout * k * k * Cin / groups
    by = 2.0 * (x.numel() + y.numel() + cw.w.numel())
    print(f"{name:9s} H{H:4d} Cin{Cin:5d} Cout{Cout:5d} k{k} s{s} g{groups} M={M:8d} {ms:8.4f} ms {fl / ms / 1e9:8.1f} TFLOP/s "
          f"{by / ms / 1e6:8.1f} GB/s", flush=True)
    return ms, fl


if __name__ == "__main__":
    B = int(os.environ.get("B", "256"))
    tot_ms = tot_fl = 0.0
    for (n, H, ci, co, k, s) in RS50:
        ms, fl = bench(n, B, H, ci, co, k, s)
        tot_ms += ms
        tot_fl += fl
    print(f"sum of distinct layers: {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
