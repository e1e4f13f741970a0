"""A/B of the pwk_direct tile height on the layers whose 256-pixel grid under-fills the chip (run twice: default and VIP_PWK_FILL=0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
CASES = [(12544, 208, 1248, True), (12544, 272, 1632, True), (43264, 128, 768, True), (50176, 160, 960, True), (50176, 256, 256, False),
         (12544, 1248, 208, False), (12544, 512, 128, False), (43264, 1024, 256, False), (12544, 2048, 512, False), (3136, 512, 2048, False)]
for M, N, K, gated in CASES:
    g = torch.Generator().manual_seed(0)
    B = 256
    hw = M // B
    x = torch.randn((B, 1, hw, K), generator=g).to("cuda", torch.float16)
    cw = ops.make_conv_weight(torch.randn(1, 1, K, N, generator=g) / K ** 0.5, torch.zeros(N))
    gate = torch.rand((B, 2, K), generator=g).to("cuda", torch.float16) if gated else None
    res = torch.randn((B, 1, hw, N), generator=g).to("cuda", torch.float16) if gated else None
    f = lambda: ops.conv2d(x, cw, residual=res, gate=gate, act=None if gated else "silu")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={B*hw:6d} N={N:5d} K={K:5d} gated={int(gated)} {ms*1e3:7.1f} us {2.0*B*hw*N*K/ms/1e9:6.1f} TF", flush=True)
