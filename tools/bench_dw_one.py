"""One depthwise shape, a few launches (for rocprofv3 --pmc passes): python tools/bench_dw_one.py H C k [B] [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
H, C, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
n = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = torch.Generator().manual_seed(0)
x = torch.randn((B, H, H, C), generator=g).to("cuda", torch.float16)
w = ops.make_dw_weight(torch.randn((k, k, C, 1), generator=g) / k)
b = torch.zeros(C, device="cuda")
p = k // 2
for _ in range(n):
    ops.dwconv2d(x, w, b, k, 1, (p, p, p, p))
torch.cuda.synchronize()
