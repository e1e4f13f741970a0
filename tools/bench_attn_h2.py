"""Timing of the packed-strict attention cores on the ensemble's shapes (B = 256): GCViT windows (ws 7 / 14, hd 32) and ViT-S MHSA (197
tokens, hd 64).  Algorithmic bytes = qkv read + out write at 4 bytes per element.
    python tools/bench_attn_h2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
_w = torch.randn(4096, 4096, device="cuda")
for _ in range(200):
    _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
g = torch.Generator().manual_seed(0)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, H, C, heads, ws in (("gcv.l0", 56, 64, 2, 7), ("gcv.l1", 28, 128, 4, 7), ("gcv.l2", 14, 256, 8, 14), ("gcv.l3", 7, 512, 16, 7)):
    qkv = ops.pack_h2(torch.randn((256, H, H, 3 * C), generator=g).cuda())
    tab = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).cuda()
    ms = timed(lambda: ops.window_attention(qkv, None, tab, heads, ws, 32 ** -0.5))
    by = 4.0 * 256 * H * H * 4 * C
    print(f"{name} ws{ws:2d} C{C:3d}: {ms*1e3:7.1f} us {by/ms/1e6:6.0f} GB/s", flush=True)
qkv = ops.pack_h2(torch.randn((256, 197, 3 * 384), generator=g).cuda())
ms = timed(lambda: ops.mhsa(qkv, 6, 0.125))
print(f"vit-s mhsa 197 x 384: {ms*1e3:7.1f} us {4.0*256*197*4*384/ms/1e6:6.0f} GB/s")
ops.h2_check("bench_attn_h2")
