"""Micro-benchmark of vip_window_attn_fwd_f16 at the four GCViT-Tiny levels (B=256, 224x224 input).
Reports time, algorithmic GB/s (q,k,v read + out write, fp16) and TFLOP/s (4*N^2*hd per window-head)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import ops  # noqa: E402

LEVELS = [("L0", 56, 64, 2, 7), ("L1", 28, 128, 4, 7), ("L2", 14, 256, 8, 14), ("L3", 7, 512, 16, 7)]


def main():
    B = int(os.environ.get("B", "256"))
    iters = 20
    for name, H, C, heads, ws in LEVELS:
        for glob in (False, True):
            g = torch.Generator().manual_seed(1)
            nq = 2 if glob else 3
            qkv = torch.randn((B, H, H, nq * C), generator=g).to("cuda", torch.float16)
            qg = torch.randn((B, ws * ws, C), generator=g).to("cuda", torch.float16) if glob else None
            table = (torch.randn(((2 * ws - 1) ** 2, heads), generator=g) * 0.5).cuda()
            ops.window_attention(qkv, qg, table, heads, ws, 32 ** -0.5)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.window_attention(qkv, qg, table, heads, ws, 32 ** -0.5)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            nwh = B * (H // ws) ** 2 * heads
            N = ws * ws
            by = nwh * 4.0 * N * 32 * 2
            fl = nwh * 4.0 * N * N * 32
            print(f"{name} ws{ws:2d} heads{heads:2d} global={int(glob)} items={nwh:7d} {ms * 1e3:8.1f} us  {by / ms / 1e6:7.1f} GB/s "
                  f"{fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
