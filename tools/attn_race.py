"""Window attention under concurrency: tools/bisect_determinism.py shows vip_window_attn_fwd_f16 returning different results for the
same inputs when another stream keeps the chip busy.  This isolates the op: fixed inputs, `--iters` launches on stream A while stream
B runs (a) nothing, (b) big copies, (c) GEMMs of this library, (d) LDS-heavy kernels of this library; every result is compared
bit for bit with a reference launch and the mismatching elements are mapped back to (image, window, head, token, channel).

    python tools/attn_race.py [--ws 14] [--B 64] [--heads 8] [--iters 200]
"""
import argparse
import collections
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ws", type=int, default=14)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--nw", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--glob", action="store_true")
    a = ap.parse_args()
    ws, B, heads = a.ws, a.B, a.heads
    C = heads * 32
    Hp = Wp = a.nw * ws
    g = torch.Generator().manual_seed(1)
    nq = 2 if a.glob else 3
    qkv = torch.randn(B, Hp, Wp, nq * C, generator=g).to(torch.float16).cuda()
    qg = torch.randn(B, ws * ws, C, generator=g).to(torch.float16).cuda() if a.glob else None
    table = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).cuda()
    scale = 32 ** -0.5
    ref = ops.window_attention(qkv, qg, table, heads, ws, scale).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    junk = torch.empty((1 << 28,), dtype=torch.uint8, device="cuda")
    xg = torch.randn(50176, 256, generator=g).to(torch.float16).cuda()
    cw = ops.make_dense_weight(torch.randn(256, 768, generator=g) / 16, torch.zeros(768))
    xd = torch.randn(64, 56, 56, 96, generator=g).to(torch.float16).cuda()
    wd = ops.make_dw_weight(torch.randn(7, 7, 96, 1, generator=g) / 7)
    xm = torch.randn(64 * 56 * 56, 96, generator=g).to(torch.float16).cuda()
    f1 = ops.make_dense_weight(torch.randn(96, 384, generator=g) / 10, torch.zeros(384))
    f2 = ops.make_dense_weight(torch.randn(384, 96, generator=g) / 20, torch.zeros(96))

    def noise(kind):
        if kind == "copies":
            junk[: 1 << 27].copy_(junk[1 << 27:])
        elif kind == "gemm":
            for _ in range(4):
                ops.dense(xg, cw, act="gelu")
        elif kind == "dwconv7":
            for _ in range(2):
                ops.dwconv2d(xd, wd, None, 7, 1, (3, 3, 3, 3))
        elif kind == "mlp_fused":
            for _ in range(2):
                ops.mlp(xm, f1, f2, act="gelu", residual=xm)
        elif kind == "attn_other":
            for _ in range(4):
                ops.window_attention(qkv, qg, table, heads, ws, scale)

    for kind in ("none", "copies", "gemm", "dwconv7", "mlp_fused", "attn_other"):
        bad_runs = 0
        where = collections.Counter()
        worst = 0.0
        for it in range(a.iters):
            if kind != "none":
                with torch.cuda.stream(sb):
                    noise(kind)
            with torch.cuda.stream(sa):
                out = ops.window_attention(qkv, qg, table, heads, ws, scale)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad_runs += 1
                d = (out.float() - ref.float()).abs()
                worst = max(worst, float(d.max()))
                idx = torch.nonzero(d > 0)
                for b, y, x, c in idx[:: max(1, len(idx) // 2000)].tolist():
                    tok = (y % ws) * ws + (x % ws)
                    where[(b, (y // ws, x // ws), c // 32, tok // 16, (c % 32) // 8)] += 1
        print(f"[attn_race] ws {ws} B {B} heads {heads} global={a.glob} | other stream: {kind:10s}: {bad_runs} of {a.iters} launches differ, worst |d| {worst:.3e}")
        if where:
            items = collections.Counter((k[0], k[1], k[2]) for k in where.elements())
            tiles = collections.Counter(k[3] for k in where.elements())
            lanes = collections.Counter(k[4] for k in where.elements())
            print(f"    distinct (image, window, head) items hit: {len(items)}; by 16-query tile: {dict(sorted(tiles.items()))}; "
                  f"by 8-channel group: {dict(sorted(lanes.items()))}")
            print(f"    most hit items: {items.most_common(5)}")


if __name__ == "__main__":
    main()
