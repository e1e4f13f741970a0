"""Micro-benchmark of the HBM-bound helper kernels on the ensemble's shapes (B = 256): LayerNorm, scale*x+residual (+act, two outputs),
global pool, SE gate.  GB/s = algorithmic bytes (inputs + outputs once) / time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = int(os.environ.get("B", "256"))
g = torch.Generator().manual_seed(0)
print("LayerNorm")
for name, rows, C in [("cnx.stem", B * 99 * 99, 96), ("cnx.s1.down", B * 49 * 49, 192), ("cnx.s2", B * 24 * 24, 384), ("cnx.s3", B * 12 * 12, 768),
                      ("gcv.l0", B * 56 * 56, 64), ("gcv.l1", B * 28 * 28, 128), ("gcv.l2", B * 14 * 14, 256), ("gcv.l3", B * 7 * 7, 512),
                      ("vit-s", B * 197, 384)]:
    x = torch.randn((rows, C), generator=g).to("cuda", torch.float16)
    gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ms = timeit(lambda: ops.layernorm(x, gm, bt, 1e-6))
    print(f"  {name:12s} rows {rows:8d} C {C:4d} {ms*1e3:8.1f} us {4.0*x.numel()/ms/1e6:7.0f} GB/s", flush=True)
print("scale_add_act (x * gate + residual)")
for name, H, C, act2 in [("nfnet.s1", 56, 256, "silu"), ("nfnet.s2", 28, 512, "silu"), ("nfnet.s3", 14, 1536, "silu"), ("nfnet.s4", 7, 1536, None),
                         ("rs50.s1", 40, 256, None), ("rs50.s2", 20, 512, None), ("rs50.s3", 10, 1024, None)]:
    x = torch.randn((B, H, H, C), generator=g).to("cuda", torch.float16)
    r = torch.randn((B, H, H, C), generator=g).to("cuda", torch.float16)
    s = torch.rand((B, 2, C), generator=g).to("cuda", torch.float16)
    ms = timeit(lambda: ops.scale_add_act(x, s, r, "relu" if act2 is None else None, act2=act2))
    nb = 2.0 * x.numel() * (3 if act2 is None else 4)
    print(f"  {name:12s} {H:3d}x{H:<3d} C {C:4d} act2={act2} {ms*1e3:8.1f} us {nb/ms/1e6:7.0f} GB/s", flush=True)
print("global_avgpool / se_gate")
for name, H, C, Cr in [("v1b4.56", 56, 192, 8), ("v1b4.28", 28, 336, 16), ("v1b4.14", 14, 960, 40), ("v1b4.7", 7, 1632, 72), ("v2t.14", 14, 624, 32)]:
    x = torch.randn((B, H, H, C), generator=g).to("cuda", torch.float16)
    fc1 = ops.make_dense_weight(torch.randn(C, Cr, generator=g) / C ** 0.5, torch.zeros(Cr))
    fc2 = ops.make_dense_weight(torch.randn(Cr, C, generator=g) / Cr ** 0.5, torch.zeros(C))
    ms = timeit(lambda: ops.se_gate(x, fc1, fc2, "silu", "sigmoid"))
    ms2 = timeit(lambda: ops.global_avgpool(x))
    print(f"  {name:12s} {H:3d}x{H:<3d} C {C:4d} se_gate {ms*1e3:7.1f} us {2.0*x.numel()/ms/1e6:6.0f} GB/s   gap {ms2*1e3:7.1f} us {2.0*x.numel()/ms2/1e6:6.0f} GB/s", flush=True)
