"""GPU parity: every HIP operator (through the C ABI) against the fp32 CPU oracle on seeded inputs.

Inputs are rounded to fp16 first, so the only differences are fp32 accumulation order and the final
fp16 rounding of the output: tolerance = 2e-3 relative to the output scale (fp16 has 2^-11 = 4.9e-4
relative precision).
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops_ref as R  # noqa: E402
from oracle import gcvit_ref  # noqa: E402


def _ops():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops
    return ops


def _experiments():
    """the experimental kernels (VIP_BUILD_EXPERIMENTS=1 python vip-cup-2022_amd/build.py --force) are not in the default library"""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import _abi
    return bool(_abi.lib().vip_experiments_built())


needs_experiments = pytest.mark.skipif("not __import__('torch').cuda.is_available() or not _experiments()",
                                       reason="experimental kernel: build with VIP_BUILD_EXPERIMENTS=1")


def h(t):
    """fp16-rounded fp32 copy (CPU) of t"""
    return t.to(torch.float16).to(torch.float32)


def dev(t):
    return t.to(torch.float16).cuda().contiguous()


def check(report, name, got, ref, tol=2e-3):
    got = got.float().cpu()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    rms = ((got - ref) ** 2).mean().sqrt().item()
    report(f"[ops] {name}: max_abs_err={err:.3e} rms={rms:.3e} ref_absmax={scale:.3e} rel={err / scale:.3e}")
    assert torch.isfinite(got).all(), name
    assert err <= tol * scale, f"{name}: err {err} > {tol}*{scale}"


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad(t,b,l,r), groups, act, residual
    (2, 9, 9, 8, 32, 3, 2, (1, 1, 1, 1), 1, "relu", False),      # stem-like: Cin=8, K=72 (partial k-tile)
    (2, 12, 10, 32, 64, 3, 1, (1, 1, 1, 1), 1, "relu", False),   # 3x3 same
    (3, 7, 7, 64, 256, 1, 1, (0, 0, 0, 0), 1, None, True),       # 1x1 + residual
    (2, 13, 13, 128, 128, 3, 2, (1, 1, 1, 1), 1, "silu", False),  # stride 2 odd size
    (1, 20, 20, 24, 40, 3, 1, (1, 1, 1, 1), 1, "gelu", False),   # channels not multiples of 32/64
    (2, 8, 8, 128, 128, 3, 1, (1, 1, 1, 1), 2, "relu", False),   # grouped (NFNet / ResNeSt style)
    (2, 10, 10, 16, 200, 4, 2, (0, 0, 0, 0), 1, None, False),    # 4x4/2 VALID patchify, Cout not /64
    (2, 6, 6, 512, 72, 1, 1, (0, 0, 0, 0), 1, "sigmoid", False),  # deep K, narrow N
    (1, 33, 31, 64, 64, 3, 1, (1, 1, 1, 1), 1, "relu", True),    # M tail (1023 pixels)
    (2, 9, 9, 64, 64, 5, 1, (2, 2, 2, 2), 1, None, False),       # 5x5
    (2, 9, 9, 96, 96, 2, 2, (0, 0, 0, 0), 1, None, False),       # 2x2/2 downsample (ConvNeXt)
    (2, 9, 9, 32, 32, 3, 2, (0, 1, 0, 1), 1, "silu", False),     # TF SAME asymmetric pad (EffNetV1)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c[:9]))
def test_conv2d(case, report):
    ops = _ops()
    B, H, W, Cin, Cout, k, s, pad, groups, act, use_res = case
    g = torch.Generator().manual_seed(hash(case[:9]) % (2 ** 31))
    x = h(torch.randn(B, H, W, Cin, generator=g))
    w = h(torch.randn(k, k, Cin // groups, Cout, generator=g) / math.sqrt(k * k * Cin / groups))
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = R.conv2d(x, w, bias, s, pad, groups)
    ref = R.act(ref, act)
    res = None
    if use_res:
        res = h(torch.randn(*ref.shape, generator=g))
        ref = ref + res
    cw = ops.make_conv_weight(w, bias, groups=groups)
    got = ops.conv2d(dev(x), cw, stride=s, pad=pad, act=act, residual=None if res is None else dev(res))
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    check(report, f"conv2d {case}", got, ref)


# the weights-stationary streaming kernel (1x1 stride 1, K <= 256, M >= 65536): every k-step template, ragged
# K (not a multiple of 32), ragged N (not a multiple of 64), an M tail, N chunked over blockIdx.y, residual
PW_CASES = [
    # M(=B*H*W as B,H,W), Cin, Cout, act, residual
    ((2, 181, 182), 24, 144, "silu", False),
    ((1, 257, 256), 64, 256, "relu", True),
    ((4, 128, 129), 96, 384, "gelu", False),
    ((2, 200, 170), 128, 56, None, True),
    ((1, 300, 221), 160, 960, "silu", False),
    ((1, 256, 257), 256, 768, "gelu", True),
    ((3, 150, 150), 40, 8, "sigmoid", False),
]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}")
def test_conv2d_pointwise_stream(case, report):
    ops = _ops()
    (B, H, W), Cin, Cout, act, use_res = case
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = h(torch.randn(B, H, W, Cin, generator=g))
    w = h(torch.randn(1, 1, Cin, Cout, generator=g) / math.sqrt(Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = R.act(R.conv2d(x, w, bias, 1, (0, 0, 0, 0), 1), act)
    res = None
    if use_res:
        res = h(torch.randn(*ref.shape, generator=g))
        ref = ref + res
    cw = ops.make_conv_weight(w, bias)
    got = ops.conv2d(dev(x), cw, act=act, residual=None if res is None else dev(res))
    torch.cuda.synchronize()
    check(report, f"conv2d pointwise-stream {case}", got, ref)


def split_gate(g32):
    """fp32 [B, C] -> the split fp16 gate [B, 2, C] (cuda) the kernels consume"""
    hi = g32.to(torch.float16)
    lo = (g32 - hi.float()).to(torch.float16)
    return torch.stack([hi, lo], 1).cuda().contiguous()


# squeeze-excite gate (split: ~22 bits) folded into the pointwise conv's activation load vs scale_add_act then conv
@pytest.mark.parametrize("B,H,W,Cin,Cout,use_res", [(3, 14, 14, 672, 112, True), (2, 57, 56, 144, 32, False),
                                                    (2, 7, 7, 1632, 272, True), (4, 9, 9, 200, 72, False)])
def test_conv2d_gated(B, H, W, Cin, Cout, use_res, report):
    ops = _ops()
    g = torch.Generator().manual_seed(Cin + Cout)
    x = h(torch.randn(B, H, W, Cin, generator=g))
    gate = torch.rand(B, Cin, generator=g)                   # fp32: the gate is NOT representable in fp16
    w = h(torch.randn(1, 1, Cin, Cout, generator=g) / math.sqrt(Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    res = h(torch.randn(B, H, W, Cout, generator=g)) if use_res else None
    gd = split_gate(gate)
    geff = (gd[:, 0].float() + gd[:, 1].float()).cpu()       # what the two planes carry: the gate to ~2^-22
    assert (geff - gate).abs().max().item() < 1e-6
    xs = h(x * geff[:, None, None, :])
    ref = R.conv2d(xs, w, bias, 1, (0, 0, 0, 0), 1)
    if use_res:
        ref = ref + res
    cw = ops.make_conv_weight(w, bias)
    xd = dev(x)
    rd = None if res is None else dev(res)
    got = ops.conv2d(xd, cw, residual=rd, gate=gd)
    xg = ops.scale_add_act(xd, gd, None, None)
    two = ops.conv2d(xg, cw, residual=rd)
    torch.cuda.synchronize()
    assert torch.equal(xg.cpu().float(), xs), "scale_add_act with a split gate must be the correctly rounded product"
    check(report, f"conv2d gated {B}x{H}x{W}x{Cin}->{Cout}", got, ref)
    # the kernel forms fma(x, hi, x * lo) in fp16; scale_add_act rounds x * (hi + lo) from fp32: the two differ only where
    # the inner rounding of x * lo (2^-22 of the product) tips a final rounding - rare, and one ulp of one operand
    d = (got.float() - two.float()).abs().max().item()
    report(f"[ops] conv2d gated vs scale-then-conv: max diff {d:.3e}")
    assert d <= 2e-3 * ref.abs().max().item()
    # and with a gate that IS fp16-representable (lo plane zero) the two paths are bit-identical
    g16 = torch.stack([gd[:, 0], torch.zeros_like(gd[:, 0])], 1).contiguous()
    assert torch.equal(ops.conv2d(xd, cw, residual=rd, gate=g16),
                       ops.conv2d(ops.scale_add_act(xd, g16, None, None), cw, residual=rd))


def test_conv2d_im2col_pointwise_kernel_all_cases():
    """The im2col staging of the pointwise kernel is dispatched for stems and for layers with >= 32 K pixels by default; VIP_PWK_CONV=1 routes every
    eligible convolution through it (the switch is read once per process, hence the subprocess)."""
    import os, subprocess, sys
    env = dict(os.environ, VIP_PWK_CONV="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k",
                        "test_conv2d and not im2col and not two_term", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_conv2d_act_post_and_channel_slices(report):
    """act applied after the residual; input/outputs addressed as channel slices of wider tensors."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    x = h(torch.randn(2, 8, 8, 64, generator=g))
    w = h(torch.randn(3, 3, 32, 48, generator=g) / math.sqrt(9 * 32))
    res = h(torch.randn(2, 8, 8, 48, generator=g))
    ref = R.act(R.conv2d(x[..., 32:], w, None, 1, (1, 1, 1, 1)) + res, "relu")
    cw = ops.make_conv_weight(w, None)
    out = torch.zeros(2, 8, 8, 96, dtype=torch.float16, device="cuda")
    ops.conv2d(dev(x), cw, pad=(1, 1, 1, 1), act_post="relu", residual=dev(res), out=out, cin_off=32, cout_off=48)
    torch.cuda.synchronize()
    check(report, "conv2d slices", out[..., 48:], ref)
    assert out[..., :48].abs().max().item() == 0.0


@pytest.mark.parametrize("M,K,N", [(256, 64, 16), (1000, 256, 768), (197 * 3, 192, 576), (5, 2048, 8), (4096, 768, 3072),
                                   (300, 200, 136), (12544, 1248, 208), (777, 72, 40), (2500, 1536, 384)])
def test_dense(M, K, N, report):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K + N)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(K, N, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    res = h(torch.randn(M, N, generator=g))
    ref = R.act(R.dense(x, w, b), "gelu") + res
    got = ops.dense(dev(x), ops.make_dense_weight(w, b), act="gelu", residual=dev(res))
    torch.cuda.synchronize()
    check(report, f"dense {M}x{K}x{N}", got, ref)


# at most 256 rows (squeeze-excite / ECA layers, M = batch): the barrier-free rows kernel; ragged M, N, K tails
@pytest.mark.parametrize("act", ["relu", "sigmoid", "silu", None])
@pytest.mark.parametrize("M,K,N", [(256, 2048, 512), (256, 72, 1632), (200, 1536, 1536), (16, 1248, 56), (1, 64, 8), (256, 40, 104)])
def test_dense_few_rows(M, K, N, act, report):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 3 + K + N)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(K, N, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    ref = R.act(R.dense(x, w, b), act)
    got = ops.dense(dev(x), ops.make_dense_weight(w, b), act=act)
    torch.cuda.synchronize()
    check(report, f"dense few-rows {act} {M}x{K}x{N}", got, ref)


# epilogue variants of the pointwise kernels (activation only / residual / residual + post-ReLU) over the general-K
# kernel's edge cases: K tail (K % 64 != 0), odd chunk counts, N <= 64 and N not a multiple of 64/128, M tail
@pytest.mark.parametrize("mode", ["gelu", "silu", "relu", "sigmoid", "none", "res", "res_relu"])
@pytest.mark.parametrize("M,K,N", [(300, 200, 136), (12544, 1248, 208), (777, 72, 40), (2500, 1536, 384), (4096, 768, 3072),
                                   (1000, 320, 64)])
def test_dense_pointwise_modes(M, K, N, mode, report):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K + N)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(K, N, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    cw = ops.make_dense_weight(w, b)
    if mode in ("res", "res_relu"):
        res = h(torch.randn(M, N, generator=g))
        ref = R.dense(x, w, b) + res
        post = "relu" if mode == "res_relu" else None
        ref = R.act(ref, post)
        got = ops.dense(dev(x), cw, act_post=post, residual=dev(res))
    else:
        act = None if mode == "none" else mode
        ref = R.act(R.dense(x, w, b), act)
        got = ops.dense(dev(x), cw, act=act)
    torch.cuda.synchronize()
    check(report, f"dense-pointwise {mode} {M}x{K}x{N}", got, ref)


# the LDS-DMA deep-K kernel (csrc/gemm8p.hpp: 256 x 256 x 64 tiles, K % 64 == 0, N % 256 == 0, >= 128 tiles): ragged M, 2-12 channel
# tiles, short and long K loops, every epilogue family; the dispatcher must really pick it
@pytest.mark.parametrize("mode", ["gelu", "res", "none"])
@pytest.mark.parametrize("M,K,N", [(33017, 384, 512), (20000, 1536, 768), (16640, 768, 3072), (65536 + 5, 448, 256), (8500, 3072, 1024)])
def test_dense_gemm8p(M, K, N, mode, report, monkeypatch):
    monkeypatch.setenv("VIP_G8P_MINK", "256")          # the product default is 1024 (three-stream step); read per call
    ops = _ops()
    from vipcup_amd import _abi
    g = torch.Generator().manual_seed(M + K + N)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(K, N, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    cw = ops.make_dense_weight(w, b)
    d = _abi.ConvDesc(B=M, H=1, W=1, Cin=K, Cout=N, kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Ho=1, Wo=1, groups=1, ldx=K, cin_off=0, ldy=N,
                      cout_off=0, ldr=N if mode == "res" else 0, res_off=0, ldw=cw.ldw, act_pre=3 if mode == "gelu" else 0, act_post=0)
    assert ops.conv_kernel_name(d, mode == "res") == "gemm8p_kernel"
    if mode == "res":
        res = h(torch.randn(M, N, generator=g))
        ref = R.dense(x, w, b) + res
        got = ops.dense(dev(x), cw, residual=dev(res))
    else:
        act = None if mode == "none" else mode
        ref = R.act(R.dense(x, w, b), act)
        got = ops.dense(dev(x), cw, act=act)
    torch.cuda.synchronize()
    check(report, f"dense-gemm8p {mode} {M}x{K}x{N}", got, ref)
    got2 = ops.dense(dev(x), cw, act=None if mode != "gelu" else "gelu", residual=dev(res) if mode == "res" else None)
    torch.cuda.synchronize()
    assert torch.equal(got, got2), "two launches on the same operands must agree bit for bit (no race in the DMA ring)"


# the activation-resident short-K / wide-N kernel (pwx_kernel in csrc/conv_igemm.hip: K <= 256, N >= 256, M >= 16384): 2 / 3 / 4 k chunks and a
# K tail, a half-empty last channel tile, ragged M, every epilogue family; the dispatcher must really pick it, and it must agree bit for bit
# with the kernel it replaces (VIP_PWX=0 is read once per process, so the comparison is against the fp32 oracle and a repeat launch)
@pytest.mark.parametrize("mode", ["gelu", "res", "res_relu", "none"])
@pytest.mark.parametrize("M,K,N", [(16384 + 37, 256, 768), (20000, 192, 320), (16500, 128, 256), (17000, 200, 1024), (16384, 72, 384)])
def test_dense_pwx(M, K, N, mode, report, monkeypatch):
    monkeypatch.setenv("VIP_PWX", "1")          # an experiment, off by default; read per call
    ops = _ops()
    from vipcup_amd import _abi
    g = torch.Generator().manual_seed(M + K + N)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(K, N, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    cw = ops.make_dense_weight(w, b)
    has_res = mode in ("res", "res_relu")
    d = _abi.ConvDesc(B=M, H=1, W=1, Cin=K, Cout=N, kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Ho=1, Wo=1, groups=1, ldx=K, cin_off=0, ldy=N,
                      cout_off=0, ldr=N if has_res else 0, res_off=0, ldw=cw.ldw, act_pre=3 if mode == "gelu" else 0,
                      act_post=1 if mode == "res_relu" else 0)
    assert ops.conv_kernel_name(d, has_res) == "pwx_kernel"
    if has_res:
        res = h(torch.randn(M, N, generator=g))
        post = "relu" if mode == "res_relu" else None
        ref = R.act(R.dense(x, w, b) + res, post)
        run = lambda: ops.dense(dev(x), cw, act_post=post, residual=dev(res))  # noqa: E731
    else:
        act = None if mode == "none" else mode
        ref = R.act(R.dense(x, w, b), act)
        run = lambda: ops.dense(dev(x), cw, act=act)  # noqa: E731
    got = run()
    torch.cuda.synchronize()
    check(report, f"dense-pwx {mode} {M}x{K}x{N}", got, ref)
    assert torch.equal(got, run()), "two launches on the same operands must agree bit for bit"


# fused MLP (hidden tensor in registers): LDS-resident (C 64/96) and streamed (C 192) weights, M tails, with/without residual, vs two fp32 denses
@pytest.mark.parametrize("use_ln", [False, True])
@pytest.mark.parametrize("M,C,hid,use_res", [(8192, 96, 384, True), (20011, 96, 384, False), (9000, 64, 256, True),
                                             (8192 + 513, 64, 192, True), (9001, 192, 768, True),
                                             (10000, 192, 384, True)])
def test_mlp_fused(M, C, hid, use_res, use_ln, report):
    ops = _ops()
    from vipcup_amd import _abi
    assert _abi.lib().vip_mlp_fused_supported(M, C, hid, 3)
    g = torch.Generator().manual_seed(M + C + hid)
    x = h(torch.randn(M, C, generator=g) * 1.5 + 0.3)
    w1 = h(torch.randn(C, hid, generator=g) / math.sqrt(C))
    b1 = torch.randn(hid, generator=g) * 0.1
    w2 = h(torch.randn(hid, C, generator=g) / math.sqrt(hid))
    b2 = torch.randn(C, generator=g) * 0.1
    res = h(torch.randn(M, C, generator=g)) if use_res else None
    ln = None
    xin = x
    if use_ln:
        lg, lb = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.1
        ln = (lg.cuda(), lb.cuda(), 1e-6)
        xin = R.layernorm(x, lg, lb, 1e-6)
    ref = R.dense(R.act(R.dense(xin, w1, b1), "gelu"), w2, b2)
    if use_res:
        ref = ref + res
    got = ops.mlp(dev(x), ops.make_dense_weight(w1, b1), ops.make_dense_weight(w2, b2), act="gelu",
                  residual=None if res is None else dev(res), ln=ln)
    torch.cuda.synchronize()
    # the normalised input and the hidden activations are rounded to fp16 (as on the unfused path of the product);
    # against the fp32 oracle that is one extra fp16 rounding inside each dot product
    check(report, f"mlp_fused M{M} C{C} hid{hid} res{use_res} ln{use_ln}", got, ref, tol=4e-3)


# squeeze-excite gate in one launch vs pool -> dense -> dense of the oracle: wide/narrow C, Cr padded to 8, HW not a
# multiple of the pixel-group count, more channels than threads
@pytest.mark.parametrize("B,H,W,C,Cr,Co,act1", [(5, 7, 7, 1632, 72, 1632, "silu"), (3, 56, 56, 192, 8, 192, "silu"),
                                                (4, 13, 13, 64, 16, 64, "gelu"), (2, 25, 25, 512, 128, 1024, "relu"),
                                                (2, 7, 7, 4352, 184, 4352, "silu")])
def test_se_gate(B, H, W, C, Cr, Co, act1, report):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 7 + C)
    x = h(torch.randn(B, H, W, C, generator=g) + 0.3)
    w1 = h(torch.randn(C, Cr, generator=g) / math.sqrt(C) * 3)
    b1 = torch.randn(Cr, generator=g) * 0.1
    w2 = h(torch.randn(Cr, Co, generator=g) / math.sqrt(Cr) * 2)
    b2 = torch.randn(Co, generator=g) * 0.1
    ref = R.act(R.dense(R.act(R.dense(x.mean(dim=(1, 2)), w1, b1), act1), w2, b2), "sigmoid")      # fp32 throughout
    fc1, fc2 = ops.make_dense_weight(w1, b1), ops.make_dense_weight(w2, b2)
    got = ops.se_gate(dev(x), fc1, fc2, act1, "sigmoid")           # host picks fused / pool + 2 GEMMs by weight size
    torch.cuda.synchronize()
    assert got.shape == (B, 2, Co)
    fused = C * fc1.cout + fc1.cout * Co <= 256 * 1024
    gsum = got[:, 0].float() + got[:, 1].float()
    # fused: fp32 pooled/hidden vectors; wide path: pooled/hidden vectors as hi + lo fp16 planes -> the split gate is exact to ~1e-6
    check(report, f"se_gate split B{B} {H}x{W} C{C} Cr{Cr} fused={fused}", gsum, ref, tol=2e-5 if fused else 4e-5)
    assert (got[:, 1].float().abs() <= got[:, 0].float().abs() * 2.0 ** -11 + 1e-7).all(), "lo plane exceeds half an ulp of hi"
    plain = ops.se_gate(dev(x), fc1, fc2, act1, "sigmoid", split=False)
    torch.cuda.synchronize()
    assert plain.shape == (B, Co)
    check(report, f"se_gate plain B{B} {H}x{W} C{C} Cr{Cr}", plain, ref)
    if fused:
        assert torch.equal(plain, got[:, 0])
    from vipcup_amd import _abi                                     # and the C entry point itself, whatever the size
    import ctypes as Ct
    out = torch.empty((B, Co), dtype=torch.float16, device="cuda")
    xd = dev(x)
    st = _abi.lib().vip_se_gate_f16(Ct.c_void_p(xd.data_ptr()), Ct.c_void_p(fc1.w.data_ptr()), Ct.c_void_p(fc1.bias.data_ptr()),
                                    Ct.c_void_p(fc2.w.data_ptr()), Ct.c_void_p(fc2.bias.data_ptr()), Ct.c_void_p(out.data_ptr()),
                                    B, H * W, C, C, fc1.cout, fc1.ldw, Co, fc2.ldw, ops._act(act1), 4, 0,
                                    Ct.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    torch.cuda.synchronize()
    check(report, f"vip_se_gate_f16 B{B} {H}x{W} C{C} Cr{Cr}", out, ref)


@pytest.mark.parametrize("k,s,C,H", [(3, 1, 64, 14), (3, 2, 72, 15), (5, 1, 40, 12), (5, 2, 48, 13), (7, 1, 96, 11)])
def test_dwconv(k, s, C, H, report):
    ops = _ops()
    g = torch.Generator().manual_seed(k * 100 + s * 10 + C)
    x = h(torch.randn(2, H, H + 1, C, generator=g))
    w = torch.randn(k, k, C, 1, generator=g) / k            # depthwise filters are fp32 at the boundary
    b = torch.randn(C, generator=g) * 0.1
    p = k // 2
    ref = R.act(R.dwconv2d(x, w, b, s, (p, p, p, p)), "gelu")
    got = ops.dwconv2d(dev(x), w[..., 0].contiguous().cuda(), b.cuda(), k, s, (p, p, p, p), act="gelu")
    torch.cuda.synchronize()
    check(report, f"dwconv k{k} s{s} C{C}", got, ref)


# depthwise conv + squeeze-excite gate with the pool's partial sums left by the depthwise kernel: ensemble block shapes (EfficientNet
# B4 / V2-T MBConv, GCViT FeatExtract), maps with one partly idle tile group per image (7x7: 8 tiles of 16 slots), several groups per
# image, channel counts that leave idle chunk lanes (cb = 12 / 16 against C8 = 18, 30), a stride-2 case and a wide gate (fallbacks)
@pytest.mark.parametrize("k,s,C,H,W,B,Cr,act", [(3, 1, 64, 56, 56, 3, 16, "gelu"), (3, 1, 144, 7, 7, 5, 8, "silu"),
                                                 (5, 1, 240, 13, 12, 4, 16, "silu"), (3, 1, 96, 25, 25, 2, 24, "silu"),
                                                 (5, 1, 672, 12, 12, 3, 32, "silu"), (3, 1, 8, 9, 31, 2, 8, "relu"),
                                                 (7, 1, 48, 10, 10, 2, 8, None), (3, 2, 72, 15, 15, 2, 8, "silu"),
                                                 (3, 1, 1152, 6, 6, 2, 288, "silu")])
def test_dwconv_se_pool(k, s, C, H, W, B, Cr, act, report):
    ops = _ops()
    g = torch.Generator().manual_seed(k * 100 + C + H)
    x = h(torch.randn(B, H, W, C, generator=g) + 0.2)
    w = torch.randn(k, k, C, 1, generator=g) / k
    b = torch.randn(C, generator=g) * 0.1
    w1 = h(torch.randn(C, Cr, generator=g) / math.sqrt(C) * 3)
    b1 = torch.randn(Cr, generator=g) * 0.1
    w2 = h(torch.randn(Cr, C, generator=g) / math.sqrt(Cr) * 2)
    b2 = torch.randn(C, generator=g) * 0.1
    p = k // 2
    pad = (p, p, p, p)
    y32 = R.act(R.dwconv2d(x, w, b, s, pad), act)                                   # fp32 map
    ref_gate = R.act(R.dense(R.act(R.dense(y32.mean(dim=(1, 2)), w1, b1), "silu"), w2, b2), "sigmoid")
    fc1, fc2 = ops.make_dense_weight(w1, b1), ops.make_dense_weight(w2, b2)
    wd, bd = w[..., 0].contiguous().cuda(), b.cuda()
    hh, gate = ops.dwconv2d_se(dev(x), wd, bd, k, s, pad, act, fc1, fc2, "silu", "sigmoid")
    plain_h = ops.dwconv2d(dev(x), wd, bd, k, s, pad, act=act)
    plain_gate = ops.se_gate(plain_h, fc1, fc2, "silu", "sigmoid")
    torch.cuda.synchronize()
    assert torch.equal(hh, plain_h)                                                  # the map itself is the same kernel arithmetic
    gsum = gate[:, 0].float() + gate[:, 1].float()
    small = C * Cr + Cr * C <= 256 * 1024
    from vipcup_amd import _abi
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    # the fallbacks (stride 2, wide gate, maps whose image-aligned tile groups would idle > 15 % of the lanes) pool the fp16-rounded map
    pooled_form = small and _abi.lib().vip_dwconv2d_pool_parts(B, H, W, C, k, s, Ho, Wo) > 0
    if (k, C, H) == (3, 64, 56):
        assert pooled_form
    report(f"[ops] dwconv_se k{k} s{s} C{C} {H}x{W}: {'pooling form' if pooled_form else 'two plain launches'}")
    check(report, f"dwconv_se k{k} s{s} C{C} {H}x{W} B{B} gate vs fp32", gsum, ref_gate, tol=2e-5 if pooled_form else 1e-4)
    # against the two plain launches: those pool the fp16-ROUNDED map, the pooling form the fp32 outputs - they differ by the rounding
    # noise of the mean, far below a gate's own fp16 ulp
    d = (gsum.cpu() - (plain_gate[:, 0].float() + plain_gate[:, 1].float()).cpu()).abs().max().item()
    report(f"[ops] dwconv_se k{k} s{s} C{C} {H}x{W}: max |gate - gate(two launches)| = {d:.2e}")
    assert d <= 2e-4
    again_h, again_gate = ops.dwconv2d_se(dev(x), wd, bd, k, s, pad, act, fc1, fc2, "silu", "sigmoid")
    torch.cuda.synchronize()
    assert torch.equal(again_gate, gate) and torch.equal(again_h, hh)                # fixed summation order: bit-reproducible


# the matrix-core depthwise kernel (k 7 / 5, stride 1, C % 16 == 0): tile tails in both axes, maps smaller than a tile, several
# tiles per image and several images per band, asymmetric padding (TF SAME on even sizes is symmetric here; VALID = no padding),
# every activation epilogue; the filter is fp32 at the boundary and the kernel carries it as hi + lo fp16 - checked to 1e-3 of |y|max
@needs_experiments
@pytest.mark.parametrize("k,C,H,W,B,pad,act", [(7, 96, 35, 21, 3, (3, 3, 3, 3), None), (7, 32, 7, 7, 5, (3, 3, 3, 3), "gelu"),
                                               (7, 16, 50, 17, 2, (0, 0, 0, 0), "relu"), (7, 48, 16, 16, 9, (2, 4, 1, 5), "silu"),
                                               (5, 336, 14, 14, 2, (2, 2, 2, 2), "silu"), (5, 16, 33, 40, 3, (1, 3, 4, 0), None),
                                               (5, 80, 3, 19, 4, (2, 2, 2, 2), "sigmoid")])
def test_dwconv_matrix_cores(k, C, H, W, B, pad, act, report):
    ops = _ops()
    g = torch.Generator().manual_seed(k * 1000 + C + H)
    x = h(torch.randn(B, H, W, C, generator=g) + 0.3)
    w = torch.randn(k, k, C, 1, generator=g) / k
    b = torch.randn(C, generator=g) * 0.1
    ref = R.act(R.dwconv2d(x, w, b, 1, pad), act)
    os.environ["VIP_DW_MFMA"] = "1"                      # opt-in kernel (read per call)
    try:
        got = ops.dwconv2d(dev(x), w[..., 0].contiguous().cuda(), b.cuda(), k, 1, pad, act=act)
        torch.cuda.synchronize()
    finally:
        del os.environ["VIP_DW_MFMA"]
    assert got.shape == ref.shape
    check(report, f"dwconv(mfma) k{k} C{C} {H}x{W} B{B} pad{pad} {act}", got, ref, tol=1e-3)
    valu = ops.dwconv2d(dev(x), w[..., 0].contiguous().cuda(), b.cuda(), k, 1, pad, act=act)   # the default (VALU) kernel
    torch.cuda.synchronize()
    assert (got.float() - valu.float()).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("C", [64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048])
def test_layernorm(C, report):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    x = h(torch.randn(37, C, generator=g) * 3 + 0.5)
    gamma = torch.randn(C, generator=g) * 0.1 + 1
    beta = torch.randn(C, generator=g) * 0.1
    ref = R.layernorm(x, gamma, beta, 1e-5)
    got = ops.layernorm(dev(x), gamma.cuda(), beta.cuda(), 1e-5)
    torch.cuda.synchronize()
    check(report, f"layernorm C{C}", got, ref)


def test_pools(report):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = h(torch.randn(2, 13, 25, 64, generator=g))
    got = ops.pool2d(dev(x), 2, 2, (0, 1, 0, 1), ops.POOL_AVG_VALID)
    check(report, "avgpool same 2x2/2 odd", got, R.avgpool_same(x, 2, 2))
    x2 = h(torch.randn(2, 14, 14, 32, generator=g)) - 2.0  # mostly negative: zero padding must win the max
    got = ops.pool2d(dev(x2), 3, 2, (1, 1, 1, 1), ops.POOL_MAX_ZEROPAD)
    check(report, "maxpool 3x3/2 zero-pad", got, R.maxpool_valid(x2, 3, 2, (1, 1, 1, 1)))
    got = ops.pool2d(dev(x), 3, 2, (1, 1, 1, 1), ops.POOL_AVG_FULL)
    check(report, "avgpool 3x3/2 zero-pad count-all", got, R.avgpool_valid(x, 3, 2, (1, 1, 1, 1)))
    got = ops.global_avgpool(dev(x))
    check(report, "global_avgpool", got, R.global_avgpool(x))


@pytest.mark.parametrize("M,K,N,act", [(5, 1536, 1536, "sigmoid"), (256, 2048, 512, "relu"), (37, 264, 72, None), (300, 512, 128, "silu")])
def test_split_vector_chain(M, K, N, act, report):
    """vip_global_avgpool_split_f16 -> vip_gemm_split2_f16: a pooled vector and a Dense on it carried as hi + lo fp16 planes
    against the fp32 oracle - ~22 bits instead of 11 (rows > 256 go through the host's chunking; K, N off the tile sizes)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    x = h(torch.randn(M, 6, 5, K, generator=g) + 0.4)
    w = h(torch.randn(K, N, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    cw = ops.make_dense_weight(w, b)
    pooled = ops.global_avgpool(dev(x), split=True)
    torch.cuda.synchronize()
    assert pooled.shape == (M, 2, K)
    pref = x.mean(dim=(1, 2))
    check(report, f"global_avgpool split M{M} C{K}", pooled[:, 0].float() + pooled[:, 1].float(), pref, tol=1e-6)
    assert torch.equal(pooled[:, 0], ops.global_avgpool(dev(x)))
    assert (pooled[:, 1].float().abs() <= pooled[:, 0].float().abs() * 2.0 ** -11 + 1e-7).all()
    got = ops.dense_split(pooled, cw, act=act)
    torch.cuda.synchronize()
    assert got.shape == (M, 2, N)
    ref = R.act(R.dense(pref, w, b), act)
    check(report, f"dense_split(split in) M{M} K{K} N{N} {act}", got[:, 0].float() + got[:, 1].float(), ref, tol=2e-5)
    one = ops.dense_split(pooled[:, 0].contiguous(), cw, act=act)           # hi plane alone: the 11-bit vector it replaces
    torch.cuda.synchronize()
    e2 = (got[:, 0].float() + got[:, 1].float() - ref.cuda()).abs().max().item()
    e1 = (one[:, 0].float() + one[:, 1].float() - ref.cuda()).abs().max().item()
    report(f"[ops] dense on pooled vector M{M} K{K}: max err {e1:.2e} (fp16 vector) -> {e2:.2e} (hi + lo planes)")
    assert e2 < e1


@pytest.mark.parametrize("k,groups,cin,cout", [(1, 1, 264, 128), (3, 2, 64, 96), (3, 1, 8, 32)])
def test_exact_weight_leg(k, groups, cin, cout, report, monkeypatch):
    monkeypatch.setenv("VIP_OFFSET_CALIBRATION", "1")     # the K-doubled twins are only built when the opt-in second pass will read them
    """ops.exact_weights(): a layer that went through ops.calibration() runs with [w | fp16(W32 - w)] along K on doubled input
    channels - same kernels, ~22-bit weights.  Against the fp32-weight oracle its error is the output rounding alone."""
    ops = _ops()
    g = torch.Generator().manual_seed(k * 100 + cin)
    x = h(torch.rand(2, 9, 9, cin, generator=g) + 0.2)                      # non-zero mean: the weight-rounding offset shows
    w = torch.randn(k, k, cin // groups, cout, generator=g) / math.sqrt(k * k * cin // groups)
    b = torch.randn(cout, generator=g) * 0.1
    pad = (k // 2,) * 4
    ref = R.act(R.conv2d(x, w, b, 1, pad, groups), "relu")
    ops.KEEP_ROUNDING_ERROR = True
    try:
        cw = ops.make_conv_weight(w, b, groups)
    finally:
        ops.KEEP_ROUNDING_ERROR = False
    b0 = cw.bias.clone()
    with ops.calibration():
        ops.conv2d(dev(x), cw, 1, pad, act="relu")
    assert cw.exact is not None and cw.exact.cin_g == 2 * cw.cin_g and torch.equal(cw.exact.bias, b0) and cw.err is None
    cw.bias = b0                                                            # plain fp16 weights, no bias correction
    y16 = ops.conv2d(dev(x), cw, 1, pad, act="relu")
    with ops.exact_weights():
        y22 = ops.conv2d(dev(x), cw, 1, pad, act="relu")
    torch.cuda.synchronize()
    e16 = (y16.float().cpu() - ref).abs().max().item()
    e22 = (y22.float().cpu() - ref).abs().max().item()
    report(f"[ops] exact-weight leg k{k} g{groups} Cin{cin}: max err {e16:.2e} (fp16 weights) -> {e22:.2e} (two-term, K doubled)")
    check(report, f"exact_weights conv k{k} g{groups}", y22, ref, tol=6e-4)
    assert e22 < e16
    ops.drop_exact_weights()
    assert cw.exact is None


@pytest.mark.parametrize("B,HW,C,N", [(3, 49, 768, 1), (2, 36, 512, 3), (5, 1, 96, 2)])
def test_gap_ln_dense_head(B, HW, C, N, report):
    """vip_gap_ln_dense_f32 vs pool -> LayerNorm -> Dense of the oracle (tfimm convnext.py:432-436), fp32 throughout."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 100 + C)
    x = h(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    gamma = torch.randn(C, generator=g) * 0.3 + 1
    beta = torch.randn(C, generator=g) * 0.1
    w = torch.randn(C, N, generator=g) / math.sqrt(C)
    b = torch.randn(N, generator=g)
    got = ops.gap_ln_dense_f32(dev(x), gamma.cuda(), beta.cuda(), 1e-6, w.t().contiguous().cuda(), b.cuda())
    ref = R.dense(R.layernorm(x.mean(1), gamma, beta, 1e-6), w, b)
    check(report, f"gap_ln_dense_f32 B{B} HW{HW} C{C} N{N}", got, ref, tol=1e-5)


# MBConv expand 1x1 + depthwise in one launch: every (k, stride), the three slab widths (64 / 48 / 32), K below / at a k-step boundary,
# two-term expand weights, tile tails, maps smaller than a tile, TF SAME padding on even and odd sizes (asymmetric for stride 2)
@needs_experiments
@pytest.mark.parametrize("k,s,Cin,Ce,H,W,B,hilo", [(3, 1, 32, 192, 13, 11, 2, True), (3, 2, 24, 144, 20, 20, 2, True),
                                                   (5, 1, 56, 336, 9, 17, 3, False), (5, 2, 32, 192, 15, 15, 2, True),
                                                   (3, 2, 56, 336, 7, 7, 3, False), (5, 1, 112, 672, 6, 6, 2, False),
                                                   (3, 1, 128, 416, 10, 10, 1, False), (3, 1, 8, 32, 3, 3, 5, False)])
def test_mbconv_expand_dw(k, s, Cin, Ce, H, W, B, hilo, report):
    """vip_mbconv_expand_dw_f16 vs conv1x1 -> silu -> zero-pad -> depthwise -> silu of the oracle (kecam efficientnet_v2.py:63-90),
    and vs the two-launch path of the product (same rounding points: agreement to an fp16 ulp of the output)."""
    ops = _ops()
    g = torch.Generator().manual_seed(k * 100 + s * 10 + Cin)
    x = h(torch.randn(B, H, W, Cin, generator=g))
    we = torch.randn(1, 1, Cin, Ce, generator=g) / math.sqrt(Cin)
    be = torch.randn(Ce, generator=g) * 0.2
    wd = torch.randn(k, k, Ce, 1, generator=g) / k
    bd = torch.randn(Ce, generator=g) * 0.1
    pt, pb = ((k - 1) // 2, k // 2) if s == 1 else same_pad_tf(H, k, s)
    pl, pr = ((k - 1) // 2, k // 2) if s == 1 else same_pad_tf(W, k, s)
    pad = (pt, pb, pl, pr)
    hid = h(R.act(R.conv2d(x, we, be), "silu"))                      # the expanded activations are fp16 in both paths
    ref = R.act(R.dwconv2d(hid, wd, bd, s, pad), "silu")
    cw = ops.make_conv_weight(we, be, hilo=hilo)
    assert (cw.w_lo is not None) == hilo
    wdd, bdd = wd[..., 0].contiguous().cuda(), bd.cuda()
    two = ops.mbconv_expand_dw(dev(x), cw, wdd, bdd, k, s, pad, act="silu")        # default: conv2d + dwconv2d
    os.environ["VIP_MBCONV_FUSED"] = "1"
    try:
        got = ops.mbconv_expand_dw(dev(x), cw, wdd, bdd, k, s, pad, act="silu")
    finally:
        del os.environ["VIP_MBCONV_FUSED"]
    torch.cuda.synchronize()
    assert got.shape == ref.shape == two.shape
    check(report, f"mbconv_expand_dw k{k} s{s} Cin{Cin} Ce{Ce} {H}x{W} hilo={hilo}", got, ref, tol=3e-3)
    d = (got.float() - two.float()).abs().max().item()
    report(f"[ops] mbconv_expand_dw vs conv2d + dwconv2d: max |diff| {d:.2e}")
    assert d <= 4e-3 * ref.abs().max().item()


def same_pad_tf(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def test_score_kernels(report):
    """vip_head_prob_f32 / vip_prob_to_score_f32 / vip_ensemble_mean_f32 vs main.py:109-114,142-143 restated with torch on the CPU."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    for N in (1, 2, 5):
        z = torch.randn(300, N, generator=g) * 4
        z[0, 0] = 60.0                                   # saturated logits stay finite
        z[1, 0] = -60.0
        p = ops.head_prob(z.cuda())
        ref = torch.sigmoid(z) if N == 1 else torch.softmax(z, dim=-1)
        check(report, f"head_prob N{N}", p, ref, tol=2e-6)
        sc = ops.binary_score(p)
        check(report, f"binary_score N{N}", sc, ref[:, 0] if N == 1 else 1.0 - ref[:, 0], tol=2e-6)
    rows = torch.rand(7, 1001, generator=g)
    buf = torch.zeros(7, 1024, device="cuda")
    buf[:, :1001] = rows.cuda()
    check(report, "ensemble_mean (strided rows)", ops.ensemble_mean(buf[:, :1001]), rows.mean(0), tol=1e-6)
    out = torch.empty(3, 300, device="cuda")
    ops.binary_score(torch.rand(300, 1, generator=g).cuda(), out=out[1])       # a row of the members x images matrix
    torch.cuda.synchronize()


def test_scale_add_act_two_outputs(report):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    x = dev(torch.randn(3, 9, 9, 64, generator=g))
    sc = dev(torch.rand(3, 64, generator=g))
    res = dev(torch.randn(3, 9, 9, 64, generator=g))
    y, y2 = ops.scale_add_act(x, sc, res, None, act2="silu")
    y_ref = ops.scale_add_act(x, sc, res, None)
    y2_ref = ops.scale_add_act(y_ref, None, None, "silu")
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(y2, y2_ref)


def test_scale_add_act_and_head(report):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = h(torch.randn(3, 7, 7, 128, generator=g))
    s = h(torch.rand(3, 128, generator=g))
    r = h(torch.randn(3, 7, 7, 128, generator=g))
    got = ops.scale_add_act(dev(x), dev(s), dev(r), "relu")
    check(report, "scale_add_act", got, torch.relu(x * s[:, None, None, :] + r))
    w = torch.randn(128, 3, generator=g) / 11
    b = torch.randn(3, generator=g)
    got = ops.gap_dense_f32(dev(x), w.t().contiguous().cuda(), b.cuda())
    check(report, "gap_dense_f32", got, R.dense(R.global_avgpool(x), w, b), tol=1e-5)


@pytest.mark.parametrize("ws,heads,nW,global_q", [(7, 2, 3, False), (7, 4, 2, True), (14, 8, 1, False), (14, 8, 2, True),
                                                  (7, 16, 1, True)])
def test_window_attention(ws, heads, nW, global_q, report):
    """vip_window_attn_fwd_f16 vs attention.py:60-80 restated (window partition done by the oracle)."""
    ops = _ops()
    B, C, hd = 2, heads * 32, 32
    Hp = Wp = ws * nW
    g = torch.Generator().manual_seed(ws * 1000 + heads * 10 + nW)
    nq = 2 if global_q else 3
    qkv = h(torch.randn(B, Hp, Wp, nq * C, generator=g))
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    qg = h(torch.randn(B, ws, ws, C, generator=g)) if global_q else None
    # oracle: window partition -> [B_, N, nq, heads, hd]
    win = R.window_partition(qkv, ws).reshape(-1, ws * ws, nq, heads, hd).permute(2, 0, 3, 1, 4)
    B_ = win.shape[1]
    if global_q:
        k, v = win[0], win[1]
        q = torch.repeat_interleave(qg, B_ // B, dim=0).reshape(B_, ws * ws, heads, hd).permute(0, 2, 1, 3)
    else:
        q, k, v = win[0], win[1], win[2]
    o = gcvit_ref.window_attention_core(q, k, v, table, ws, hd ** -0.5)
    ref = R.window_reverse(o.permute(0, 2, 1, 3).reshape(B_, ws * ws, C), ws, Hp, Wp, C)
    got = ops.window_attention(dev(qkv), None if qg is None else dev(qg).reshape(B, ws * ws, C), table.cuda(), heads, ws,
                               hd ** -0.5)
    torch.cuda.synchronize()
    check(report, f"window_attn ws{ws} heads{heads} nW{nW} global={global_q}", got, ref, tol=3e-3)


@needs_experiments
@pytest.mark.parametrize("B,heads,nW,global_q", [(128, 8, 1, False), (131, 8, 1, True), (40, 8, 2, False), (70, 4, 2, True)])
def test_window_attention_pipelined(B, heads, nW, global_q, report):
    """ws 14 with >= 1024 (window, head) items: the persistent LDS-DMA kernel (window_attn_pipe_kernel) - 2..5 items per workgroup,
    item counts that are not a multiple of the grid, several windows per image, global query; and bit-identical to the one-item
    kernel (VIP_ATTN_PIPE only changes the schedule) and to itself across launches (no race in the DMA double buffer)."""
    ops = _ops()
    import os
    os.environ["VIP_ATTN_PIPE"] = "1"          # opt-in kernel (read per call by the C entry point)
    ws, C, hd = 14, heads * 32, 32
    Hp = Wp = ws * nW
    assert B * nW * nW * heads >= 1024
    g = torch.Generator().manual_seed(B + heads * 10 + nW)
    nq = 2 if global_q else 3
    qkv = h(torch.randn(B, Hp, Wp, nq * C, generator=g))
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    qg = h(torch.randn(B, ws, ws, C, generator=g)) if global_q else None
    win = R.window_partition(qkv, ws).reshape(-1, ws * ws, nq, heads, hd).permute(2, 0, 3, 1, 4)
    B_ = win.shape[1]
    if global_q:
        k, v = win[0], win[1]
        q = torch.repeat_interleave(qg, B_ // B, dim=0).reshape(B_, ws * ws, heads, hd).permute(0, 2, 1, 3)
    else:
        q, k, v = win[0], win[1], win[2]
    o = gcvit_ref.window_attention_core(q, k, v, table, ws, hd ** -0.5)
    ref = R.window_reverse(o.permute(0, 2, 1, 3).reshape(B_, ws * ws, C), ws, Hp, Wp, C)
    qd, gd, td = dev(qkv), None if qg is None else dev(qg).reshape(B, ws * ws, C), table.cuda()
    got = ops.window_attention(qd, gd, td, heads, ws, hd ** -0.5)
    torch.cuda.synchronize()
    check(report, f"window_attn pipelined B{B} heads{heads} nW{nW} global={global_q}", got, ref, tol=3e-3)
    for _ in range(3):
        again = ops.window_attention(qd, gd, td, heads, ws, hd ** -0.5)
        torch.cuda.synchronize()
        assert torch.equal(got, again)
    os.environ["VIP_ATTN_PIPE"] = "0"
    plain = ops.window_attention(qd, gd, td, heads, ws, hd ** -0.5)
    torch.cuda.synchronize()
    del os.environ["VIP_ATTN_PIPE"]
    assert torch.equal(got, plain), "the pipelined kernel changes the schedule, not the arithmetic"


@pytest.mark.parametrize("heads", [2, 4, 8])
@pytest.mark.parametrize("B,nW,global_q", [(2, 1, False), (3, 2, True), (5, 3, False), (1, 8, True), (9, 4, False), (67, 4, True),
                                           (33, 8, False)])
def test_gcvit_attn_block_fused(B, nW, global_q, heads, report, monkeypatch):
    """vip_gcvit_attn_block_f16 (LayerNorm -> qkv -> window attention -> proj + residual in one launch, level-0 configuration)
    against block.py:58-79 restated in fp32 and against the four launches it replaces: both configurations (C = 64 / 2 heads: a wave per
    window; C = 128 / 4 heads: two waves per window and workgroup barriers), window counts that are not a multiple of the 4 windows of
    a workgroup pass, more windows than the persistent grid walks in one pass (1 072 windows > 4 x 256 workgroups at C = 128; 2 112 > 4 x 512 at C = 64), local
    and global query."""
    ops = _ops()
    ws, C, hd = (14 if heads == 8 else 7), 32 * heads, 32     # 8 heads: the level-2 configuration, one 8-wave workgroup per 196-token window
    if heads == 8:
        from vipcup_amd import _abi as _a
        if not _a.lib().vip_experiments_built():
            pytest.skip("the 14 x 14-window fused block is an experiments-build kernel (VIP_BUILD_EXPERIMENTS=1 python build.py)")
        if nW == 8:
            pytest.skip("64 windows of 196 tokens per image: covered by the smaller cases")
        nW, B = (nW + 1) // 2, (B if B < 60 else 131 * 2)      # 1 - 2 windows a side; 262 x 4 windows > one pass of the 256 workgroups
    Hp = Wp = ws * nW
    g = torch.Generator().manual_seed(B * 100 + nW * 10 + int(global_q) + heads * 1000)
    nq = 2 if global_q else 3
    x = h(torch.randn(B, Hp, Wp, C, generator=g) * 1.5 + 0.2)
    gam, bet = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
    wq, bq = h(torch.randn(C, nq * C, generator=g) / math.sqrt(C)), torch.randn(nq * C, generator=g) * 0.1
    wp, bp = h(torch.randn(C, C, generator=g) / math.sqrt(C)), torch.randn(C, generator=g) * 0.1
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    qg = h(torch.randn(B, ws, ws, C, generator=g)) if global_q else None
    # fp32 restatement
    xn = R.layernorm(x, gam, bet, 1e-5)
    qkv = R.dense(xn, wq, bq)
    win = R.window_partition(qkv, ws).reshape(-1, ws * ws, nq, heads, hd).permute(2, 0, 3, 1, 4)
    B_ = win.shape[1]
    if global_q:
        k, v = win[0], win[1]
        q = torch.repeat_interleave(qg, B_ // B, dim=0).reshape(B_, ws * ws, heads, hd).permute(0, 2, 1, 3)
    else:
        q, k, v = win[0], win[1], win[2]
    o = gcvit_ref.window_attention_core(q, k, v, table, ws, hd ** -0.5)
    att = R.window_reverse(o.permute(0, 2, 1, 3).reshape(B_, ws * ws, C), ws, Hp, Wp, C)
    ref = x + R.dense(att, wp, bp)
    cq, cp = ops.make_dense_weight(wq, bq), ops.make_dense_weight(wp, bp)
    ln = (gam.cuda(), bet.cuda(), 1e-5)
    xd, gd, td = dev(x), None if qg is None else dev(qg).reshape(B, ws * ws, C), table.cuda()
    from vipcup_amd import _abi
    assert _abi.lib().vip_gcvit_attn_block_supported(C, heads, ws) == 1 and _abi.lib().vip_gcvit_attn_block_supported(512, 16, 7) == 0
    monkeypatch.setattr(ops, "_GCVIT_BLOCK14", True)          # the ws 14 form is opt-in (not faster than the four launches)
    got = ops.gcvit_attn_block(xd, gd, ln, cq, cp, td, heads, ws, hd ** -0.5)
    torch.cuda.synchronize()
    check(report, f"gcvit_attn_block fused C{C} B{B} nW{nW} global={global_q}", got, ref, tol=3e-3)
    monkeypatch.setattr(ops, "_GCVIT_BLOCK_FUSED", False)
    four = ops.gcvit_attn_block(xd, gd, ln, cq, cp, td, heads, ws, hd ** -0.5)
    torch.cuda.synchronize()
    monkeypatch.setattr(ops, "_GCVIT_BLOCK_FUSED", True)
    check(report, f"gcvit_attn_block four launches C{C} B{B} nW{nW} global={global_q}", four, ref, tol=3e-3)
    d = (got.float() - four.float()).abs().max().item()
    report(f"[ops] gcvit_attn_block C{C} B{B} nW{nW} global={global_q}: max |fused - four launches| = {d:.2e} (|y| max {ref.abs().max().item():.2f})")
    assert d <= 2e-3 * ref.abs().max().item()          # the same roundings; the proj sum in another order: an fp16 ulp here and there
    again = ops.gcvit_attn_block(xd, gd, ln, cq, cp, td, heads, ws, hd ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(got, again)


def test_window_attention_softmax_spike(report):
    """One key dominating one query (large logit) must not overflow and must pick that key's value."""
    ops = _ops()
    ws, heads, C = 7, 2, 64
    g = torch.Generator().manual_seed(11)
    qkv = h(torch.randn(1, 7, 7, 3 * C, generator=g))
    qkv[0, 3, 3, 0:32] = 12.0          # q of token (3,3), head 0
    qkv[0, 5, 1, C:C + 32] = 12.0      # k of token (5,1), head 0  -> logit 12*12*32/sqrt(32) = 814
    table = torch.zeros(169, heads)
    win = R.window_partition(qkv, ws).reshape(-1, 49, 3, heads, 32).permute(2, 0, 3, 1, 4)
    o = gcvit_ref.window_attention_core(win[0], win[1], win[2], table, ws, 32 ** -0.5)
    ref = R.window_reverse(o.permute(0, 2, 1, 3).reshape(1, 49, C), ws, 7, 7, C)
    got = ops.window_attention(dev(qkv), None, table.cuda(), heads, ws, 32 ** -0.5)
    torch.cuda.synchronize()
    check(report, "window_attn spike", got, ref, tol=3e-3)


def test_mul_channel_slices(report):
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    a = h(torch.randn(3, 5, 7, 64, generator=g))
    b = h(torch.randn(3, 5, 7, 224, generator=g))
    got = ops.mul(dev(a), dev(b), 32, a_off=16, b_off=96)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu().float(), h(a[..., 16:48] * b[..., 96:128]))
    report("[ops] mul: channel-slice product exact (fp32 product, one rounding)")


@pytest.mark.parametrize("B,H,W,Cin,Cout,act,use_res", [(4, 56, 56, 32, 192, "silu", False), (2, 28, 28, 112, 672, "silu", False),
                                                        (3, 28, 28, 160, 40, None, True), (2, 24, 24, 256, 72, "relu", False)])
def test_conv2d_two_term_weights(B, H, W, Cin, Cout, act, use_res, report):
    """vip_conv2d_hilo_nhwc_f16: with w + w_lo the error that is COHERENT over pixels (the per-channel mean of y - y_ref,
    what fp16 weight rounding leaves behind) drops by orders of magnitude; per-element error stays at fp16 output rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    # channel means that vary along K: the weight-rounding errors add up coherently over pixels and error-diffusion
    # rounding (which assumes a flat mean) cannot cancel them
    x = h(torch.rand(B, H, W, Cin, generator=g) * 0.5 + torch.rand(Cin, generator=g) * 2.0)
    w = torch.randn(1, 1, Cin, Cout, generator=g) / math.sqrt(Cin)  # fp32 weights (NOT fp16-representable)
    bias = torch.randn(Cout, generator=g) * 0.1
    res = h(torch.randn(B, H, W, Cout, generator=g)) if use_res else None
    ref = R.act(R.conv2d(x, w, bias, 1, (0, 0, 0, 0), 1), act)
    if use_res:
        ref = ref + res
    rd = None if res is None else dev(res)
    one = ops.conv2d(dev(x), ops.make_conv_weight(w, bias), act=act, residual=rd).float().cpu()
    cw = ops.make_conv_weight(w, bias, hilo=True)
    assert cw.w_lo is not None and cw.err is None
    two = ops.conv2d(dev(x), cw, act=act, residual=rd).float().cpu()
    torch.cuda.synchronize()
    check(report, f"conv2d two-term weights {B}x{H}x{W}x{Cin}->{Cout}", two, ref)
    coh1 = (one - ref).mean(dim=(0, 1, 2)).abs().max().item()
    coh2 = (two - ref).mean(dim=(0, 1, 2)).abs().max().item()
    report(f"[ops] per-channel mean error: fp16 weights {coh1:.3e} -> two-term {coh2:.3e}")
    assert coh2 < 0.25 * coh1, (coh1, coh2)
    with pytest.raises(Exception):
        ops.conv2d(dev(x), cw, gate=split_gate(torch.rand(B, Cin)))   # gated layers cannot carry them
