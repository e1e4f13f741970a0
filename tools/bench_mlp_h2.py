"""Micro-benchmark of the strict (packed) MLP on the ensemble's narrow-token shapes: one fused launch (csrc/mlp_h2.hip) against LayerNorm +
two GEMM launches (VIP_MLP_H2_FUSED=0 semantics via ops.unfused()).  TF = logical flops (x3 MFMA work on the packed storage).
    python tools/bench_mlp_h2.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
CASES = [("cnx.s0", 256 * 99 * 99, 96, 384), ("gcv.l0", 256 * 56 * 56, 64, 256), ("gcv.l1", 256 * 28 * 28, 128, 512),
         ("gcv.l0x3", 256 * 56 * 56, 64, 192), ("gcv.l1x3", 256 * 28 * 28, 128, 384)]
_w = torch.randn(4096, 4096, device="cuda")
for _ in range(200):
    _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
for name, M, C, Hd in CASES:
    g = torch.Generator().manual_seed(0)
    x = ops.pack_h2(torch.randn((M, C), generator=g).cuda())
    with ops.precision("strict"):
        fc1 = ops.make_dense_weight(torch.randn(C, Hd, generator=g) / math.sqrt(C), torch.randn(Hd, generator=g) * 0.1)
        fc2 = ops.make_dense_weight(torch.randn(Hd, C, generator=g) / math.sqrt(Hd), torch.randn(C, generator=g) * 0.1)
    ln = (torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 1e-5)
    res = {}
    for mode in ("fused", "three"):
        def run():
            if mode == "fused":
                return ops.mlp(x, fc1, fc2, act="gelu", residual=x, ln=ln)
            with ops.unfused():
                return ops.mlp(x, fc1, fc2, act="gelu", residual=x, ln=ln)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 5
    fl = 4.0 * M * C * Hd
    print(f"{name:9s} M={M:8d} C={C:3d} hidden={Hd:4d}  fused {res['fused']*1e3:8.1f} us {fl/res['fused']/1e9:6.1f} TF   three launches {res['three']*1e3:8.1f} us", flush=True)
ops.h2_check("bench_mlp_h2")
