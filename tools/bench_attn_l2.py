"""L2-only (ws=14) window attention loop for PMC profiling."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
B, H, C, heads, ws = 256, 14, 256, 8, 14
g = torch.Generator().manual_seed(1)
qkv = torch.randn((B, H, H, 3 * C), generator=g).to("cuda", torch.float16)
table = (torch.randn(((2 * ws - 1) ** 2, heads), generator=g) * 0.5).cuda()
for _ in range(int(os.environ.get("ITERS", "5"))):
    ops.window_attention(qkv, None, table, heads, ws, 32 ** -0.5)
torch.cuda.synchronize()
