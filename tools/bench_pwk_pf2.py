"""A/B of the direct pointwise kernel against its two-chunks-ahead variant (VIP_PWK_PF2=1, read per call) on the pwk_direct shapes of
the ensemble step (profiles/r02_conv_dense_shapes.log), with a parity check of the variant against the default kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import ops  # noqa: E402

SHAPES = [  # M, N, K, act, residual
    (50432, 1152, 384, None, False), (50176, 768, 256, "gelu", False), (50432, 384, 384, None, True), (50176, 256, 256, None, True),
    (43264, 1024, 256, None, True), (50176, 768, 256, None, False), (160000, 128, 512, "relu", False), (50176, 512, 256, None, False),
    (200704, 128, 384, None, True), (43264, 1024, 256, None, False), (50176, 960, 160, "silu", False), (12544, 1248, 208, "silu", False),
    (43264, 768, 128, "silu", False), (12544, 1632, 272, "silu", False), (43264, 512, 192, "relu", False), (160000, 256, 320, None, False),
]
tot = [0.0, 0.0]
for M, N, K, act, use_res in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((M, K), generator=g, dtype=torch.float16).cuda()
    cw = ops.make_dense_weight(torch.randn((K, N), generator=g) * 0.05, torch.randn(N, generator=g) * 0.1)
    res = torch.randn((M, N), generator=g, dtype=torch.float16).cuda() if use_res else None
    out, ms = [], []
    for pf2 in ("0", "1"):
        os.environ["VIP_PWK_PF2"] = pf2
        y = ops.dense(x, cw, act=act, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.dense(x, cw, act=act, residual=res)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 20)
        out.append(y)
    d = (out[0].float() - out[1].float()).abs().max().item()
    by = 2.0 * (M * K + M * N * (2 if use_res else 1))
    tot[0] += ms[0]
    tot[1] += ms[1]
    print(f"M={M:7d} N={N:5d} K={K:4d} act={str(act):5s} res={int(use_res)}  default {ms[0] * 1e3:7.1f} us ({by / ms[0] / 1e6:5.0f} GB/s)  "
          f"pf2 {ms[1] * 1e3:7.1f} us ({by / ms[1] / 1e6:5.0f} GB/s)  x{ms[0] / ms[1]:.2f}  max|d| {d:.1e}", flush=True)
print(f"sum: default {tot[0] * 1e3:.0f} us, pf2 {tot[1] * 1e3:.0f} us")
