"""The fused GCViT level-0 attention block (vip_gcvit_attn_block_f16) against the four launches it replaces, at the ensemble's shape
(B = 256, 56 x 56, C = 64, 2 heads, 7 x 7 windows), local and global query; and the whole member either way.
    python tools/bench_gcvit_block.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops, zoo
ops._GCVIT_BLOCK14 = True      # time the opt-in ws 14 form as well


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, C, heads, ws in ((256, 56, 64, 2, 7), (256, 28, 128, 4, 7), (256, 14, 256, 8, 14)):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(B, H, H, C, generator=g) * 1.5).half().cuda()
    ln = (torch.ones(C).cuda(), torch.zeros(C).cuda(), 1e-5)
    table = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).cuda()
    cp = ops.make_dense_weight(torch.randn(C, C, generator=g) / 8, torch.zeros(C))
    for global_q in (False, True):
        nq = 2 if global_q else 3
        cq = ops.make_dense_weight(torch.randn(C, nq * C, generator=g) / 8, torch.zeros(nq * C))
        qg = torch.randn(B, ws * ws, C, generator=g).half().cuda() if global_q else None
        out = {}
        for fused in (True, False):
            ops._GCVIT_BLOCK_FUSED = fused
            out[fused] = timeit(lambda: ops.gcvit_attn_block(x, qg, ln, cq, cp, table, heads, ws, 32 ** -0.5))
        N = ws * ws
        nwin = B * (H // ws) ** 2
        flops = nwin * ((2 + 2 * nq) * N * C * C + 4.0 * N * N * C)       # qkv + proj GEMMs + the core (8 N C^2 + 4 N^2 C with nq = 3)
        byts = nwin * 4.0 * N * C                                          # x in, y out (fp16)
        print(f"C={C} {H}x{H} global_q={global_q}: fused {out[True]:7.1f} us ({byts / out[True] / 1e3:5.0f} GB/s algorithmic, "
              f"{flops / out[True] / 1e6:5.0f} TFLOP/s)   four launches {out[False]:7.1f} us", flush=True)
ops._GCVIT_BLOCK_FUSED = True
ops._GCVIT_BLOCK14 = False     # the member as shipped
spec, model = zoo.build_member("gcvit_tiny")
xi = torch.rand(256, 224, 224, 8).half().cuda()
for fused in (True, False, True, False):
    ops._GCVIT_BLOCK_FUSED = fused
    print(f"gcvit_tiny B=256 fused={fused}: {timeit(lambda: model.predict(xi), 5) / 1e3:.2f} ms", flush=True)
