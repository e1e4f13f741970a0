"""CPU: independent NumPy re-derivations of the oracle primitives on tiny shapes (the reference has no tests or
golden vectors for these — SURVEY.md F3 — so the oracle is cross-checked against first-principles loops)."""
import math

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import gcvit_ref, kecam_ref, ops_ref as R


def _conv_np(x, w, stride, pad, groups=1):
    pt, pb, pl, pr = pad
    B, H, W, C = x.shape
    kh, kw, cg, co = w.shape
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C))
    xp[:, pt:pt + H, pl:pl + W] = x
    Ho, Wo = (H + pt + pb - kh) // stride + 1, (W + pl + pr - kw) // stride + 1
    y = np.zeros((B, Ho, Wo, co))
    og = co // groups
    for o in range(co):
        g = o // og
        for i in range(Ho):
            for j in range(Wo):
                patch = xp[:, i * stride:i * stride + kh, j * stride:j * stride + kw, g * cg:(g + 1) * cg]
                y[:, i, j, o] = (patch * w[None, :, :, :, o]).sum(axis=(1, 2, 3))
    return y


@pytest.mark.parametrize("stride,pad,groups", [(1, (1, 1, 1, 1), 1), (2, (0, 1, 0, 1), 1), (2, (1, 1, 1, 1), 2)])
def test_conv2d(stride, pad, groups):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 6, 7, 4))
    w = rng.normal(size=(3, 3, 4 // groups, 6))
    got = R.conv2d(torch.tensor(x, dtype=torch.float32), torch.tensor(w, dtype=torch.float32), None, stride, pad, groups)
    assert np.allclose(got.numpy(), _conv_np(x, w, stride, pad, groups), atol=1e-4)


def test_same_padding_puts_the_odd_pixel_after():
    assert R.same_pad(224, 3, 2) == (0, 1) and R.same_pad(13, 2, 2) == (0, 1) and R.same_pad(25, 3, 1) == (1, 1)


def test_avgpool_same_divides_by_valid_taps():
    x = torch.arange(2 * 5 * 5 * 1, dtype=torch.float32).reshape(2, 5, 5, 1)
    y = R.avgpool_same(x, 2, 2).numpy()
    xn = x.numpy()
    assert y.shape == (2, 3, 3, 1)
    assert np.isclose(y[0, 0, 0, 0], xn[0, 0:2, 0:2, 0].mean())
    assert np.isclose(y[0, 2, 2, 0], xn[0, 4, 4, 0])              # single valid tap: divisor 1, not 4
    assert np.isclose(y[0, 0, 2, 0], xn[0, 0:2, 4, 0].mean())     # two valid taps


def test_maxpool_sees_zero_padding():
    x = -torch.ones(1, 4, 4, 1)
    y = R.maxpool_valid(x, 3, 2, (1, 1, 1, 1))
    assert y[0, 0, 0, 0].item() == 0.0 and y[0, 1, 1, 0].item() == -1.0


def test_layernorm_and_gelu():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(5, 16))
    g, b = rng.normal(size=16), rng.normal(size=16)
    want = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-5) * g + b
    got = R.layernorm(torch.tensor(x, dtype=torch.float32), torch.tensor(g, dtype=torch.float32),
                      torch.tensor(b, dtype=torch.float32), 1e-5)
    assert np.allclose(got.numpy(), want, atol=1e-5)
    v = np.array([-2.0, -0.5, 0.0, 0.7, 3.0])
    want = np.array([0.5 * t * (1 + math.erf(t / math.sqrt(2))) for t in v])
    assert np.allclose(R.act(torch.tensor(v, dtype=torch.float32), "gelu").numpy(), want, atol=1e-6)


@settings(max_examples=20, deadline=None)
@given(st.integers(1, 3), st.integers(1, 3), st.sampled_from([2, 7]), st.integers(1, 5))
def test_window_partition_reverse_roundtrip(nh, nw, ws, c):
    x = torch.arange(2 * nh * ws * nw * ws * c, dtype=torch.float32).reshape(2, nh * ws, nw * ws, c)
    w = R.window_partition(x, ws)
    assert w.shape == (2 * nh * nw, ws, ws, c)
    assert torch.equal(R.window_reverse(w, ws, nh * ws, nw * ws, c), x)
    assert torch.equal(w[1], x[0, :ws, ws:2 * ws] if nw > 1 else x[0, ws:2 * ws, :ws] if nh > 1 else x[1, :ws, :ws])


def test_relative_position_index_formula():
    ws = 7
    idx = R.relative_position_index(ws)
    for q in (0, 13, 48):
        for k in (0, 24, 48):
            dy, dx = q // ws - k // ws, q % ws - k % ws
            assert idx[q, k].item() == (dy + ws - 1) * (2 * ws - 1) + (dx + ws - 1)


def test_window_attention_core_against_loops():
    rng = np.random.default_rng(2)
    ws, heads, hd = 2, 2, 4
    q, k, v = (torch.tensor(rng.normal(size=(3, heads, ws * ws, hd)), dtype=torch.float32) for _ in range(3))
    table = torch.tensor(rng.normal(size=((2 * ws - 1) ** 2, heads)), dtype=torch.float32)
    got = gcvit_ref.window_attention_core(q, k, v, table, ws, 0.5).numpy()
    idx = R.relative_position_index(ws).numpy()
    for b in range(3):
        for h in range(heads):
            s = (q[b, h].numpy() * 0.5) @ k[b, h].numpy().T + table.numpy()[idx, h]
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            assert np.allclose(got[b, h], p @ v[b, h].numpy(), atol=1e-5)


def test_bicubic_is_identity_at_scale_one_and_weights_sum_to_one():
    img = torch.rand(9, 11, 3) * 255
    assert torch.equal(R.resize_bicubic(img, 9, 11), img)
    for out, inp in ((224, 200), (200, 256), (7, 20)):
        idx, w = R.bicubic_weights_and_indices(out, inp)
        assert torch.allclose(w.sum(1), torch.ones(out), atol=1e-6)
        assert idx.min() >= 0 and idx.max() <= inp - 1
    # interior sample of a linear ramp is reproduced (Keys cubic has linear precision away from the borders)
    ramp = torch.arange(40, dtype=torch.float32)[None, :, None].expand(4, 40, 1).contiguous()
    up = R.resize_bicubic(ramp, 4, 80)
    x = 20
    assert abs(up[0, x, 0].item() - ((x + 0.5) * 0.5 - 0.5)) < 2e-3


def test_make_divisible_and_eca_kernel():
    assert [kecam_ref.make_divisible(v * 1.4, 8) for v in (16, 24, 40, 80, 112, 192, 320)] == [24, 32, 56, 112, 160, 272, 448]
    assert kecam_ref.make_divisible(192 * 0.25 / 4, 1) == 12
    assert [kecam_ref.eca_kernel_size(c) for c in (256, 512, 1536)] == [5, 5, 5]


def test_fold_bn_equals_unfolded():
    import vipcup_amd  # noqa: F401
    from vipcup_amd.synth import fold_bn
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.normal(size=(1, 5, 5, 4)), dtype=torch.float32)
    w = torch.tensor(rng.normal(size=(3, 3, 4, 6)), dtype=torch.float32)
    g, b, m = (torch.tensor(rng.normal(size=6), dtype=torch.float32) for _ in range(3))
    var = torch.tensor(rng.uniform(0.5, 1.5, size=6), dtype=torch.float32)
    wf, bf = fold_bn(w, g, b, m, var, 1e-5)
    want = R.batchnorm(R.conv2d(x, w, None, 1, (1, 1, 1, 1)), g, b, m, var, 1e-5)
    assert torch.allclose(R.conv2d(x, wf, bf, 1, (1, 1, 1, 1)), want, atol=1e-5)


def test_diffusion_rounding_keeps_row_sums():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(8, 2048, generator=g) * 0.05
    q = ops.diffuse_round_f16(w).float()
    near = w.half().float()
    ulp = 2.0 ** -11 * 0.05 * 4
    assert (q - w).abs().max() <= ulp                                          # each value within one ulp
    assert ((q - w).sum(1).abs() <= 2e-5).all()                                # the row-sum error does not grow with K
    assert (near - w).sum(1).abs().mean() > 5 * (q - w).sum(1).abs().mean()    # nearest rounding: ~sqrt(K) half-ulps
