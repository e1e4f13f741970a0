"""CPU: a SECOND, independent derivation of the oracle's trickiest floating-point semantics - float64 NumPy with explicit loops, written
from the text of the reference sources (file:line cited per case) and sharing no helper with oracle/*.py - compared with the oracle at
1e-6 (VERDICT r3 item 7 ii / SURVEY.md section 8(c)(iii)).  It cannot pin the oracle to TensorFlow (none here: "parity unpinned" stays),
but a mistake would now have to be made twice, in two different formulations, to go unnoticed.

  * GCViT WindowAttention core incl. the relative-position index          models/gcvit/layers/attention.py:39-50,69-79
  * ResNeSt split attention (radix 2) incl. rsoftmax and the strided tail  kecam resnest/resnest.py:16-24,27-66
  * NFNet ScaledStandardizedConv2D                                         kecam nfnets/nfnets.py:42-81
  * TF "SAME" padding for stride-2 convolutions (odd / even sizes), AveragePooling2D("same") valid-count divisor, zero-padded
    AveragePooling2D / MaxPool2D                                            resnet_rs_model.py:207-212, resnest.py:63-65, feature.py:151-152
"""
import math

import numpy as np
import pytest
import torch

from oracle import gcvit_ref, kecam_ref
from oracle import ops_ref as R

TOL = 1e-6


def t32(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def conv_loops(x, w, b, stride, pad, groups=1):
    """direct NHWC convolution in float64: y[b,i,j,o] = sum_{r,s,c} xpad[b, i*st + r, j*st + s, g*cg + c] * w[r,s,c,o]"""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    B, H, W, C = x.shape
    kh, kw, cg, O = w.shape
    pt, pb, pl, pr = pad
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C))
    xp[:, pt:pt + H, pl:pl + W] = x
    Ho, Wo = (H + pt + pb - kh) // stride + 1, (W + pl + pr - kw) // stride + 1
    y = np.zeros((B, Ho, Wo, O))
    og = O // groups
    for o in range(O):
        g = o // og
        for i in range(Ho):
            for j in range(Wo):
                patch = xp[:, i * stride:i * stride + kh, j * stride:j * stride + kw, g * cg:(g + 1) * cg]
                y[:, i, j, o] = (patch * w[None, :, :, :, o]).sum((1, 2, 3))
    if b is not None:
        y += np.asarray(b, np.float64)
    return y


def test_window_attention_core_and_index_float64():
    """attention.py:39-50: coords = meshgrid(arange(ws), arange(ws), indexing='ij'); rel = coords[:, :, None] - coords[:, None, :];
    rel[0] += ws - 1; rel[1] += ws - 1; rel[0] *= 2 ws - 1; index = rel.sum(0).  :69-79: q *= scale; attn = q k^T + table[index];
    softmax; attn v."""
    rng = np.random.default_rng(0)
    for ws, heads in ((7, 2), (14, 1), (3, 3)):
        N, hd = ws * ws, 8
        q, k, v = (rng.standard_normal((2, heads, N, hd)) for _ in range(3))
        table = rng.standard_normal(((2 * ws - 1) ** 2, heads)) * 0.5
        scale = hd ** -0.5
        want = np.zeros((2, heads, N, hd))
        for b in range(2):
            for h in range(heads):
                for i in range(N):
                    yi, xi = divmod(i, ws)                                   # token order: row-major over the window (window.py:3-15)
                    logits = np.empty(N)
                    for j in range(N):
                        yj, xj = divmod(j, ws)
                        idx = (yi - yj + ws - 1) * (2 * ws - 1) + (xi - xj + ws - 1)
                        logits[j] = (q[b, h, i] * scale) @ k[b, h, j] + table[idx, h]
                    e = np.exp(logits - logits.max())
                    want[b, h, i] = (e / e.sum()) @ v[b, h]
        got = gcvit_ref.window_attention_core(t32(q), t32(k), t32(v), t32(table), ws, scale).double().numpy()
        assert np.abs(got - want).max() <= 5 * TOL, ws                      # fp32 oracle vs float64: softmax over <= 196 terms


def test_split_attention_float64():
    """resnest.py:27-66 with groups (radix) 2: per half a 3x3 conv (torch padding 1, no bias) on ITS half of the input channels ->
    concat -> BN -> ReLU -> sum of the halves -> GAP -> 1x1 conv + bias -> BN -> ReLU -> 1x1 conv + bias -> rsoftmax over the radix axis
    of the [B, 1, radix, C] view (:16-24) -> weighted sum of the halves; stride 2: ZeroPadding2D(1) + AveragePooling2D(3, 2) (:63-65)."""
    rng = np.random.default_rng(1)
    B, H, W, Cin, filters, eps = 2, 6, 5, 8, 6, 1e-5
    red = 4
    p, name = {}, "blk_"
    x = rng.standard_normal((B, H, W, Cin))
    for i in (1, 2):
        p[f"{name}1_g{i}_conv/kernel"] = rng.standard_normal((3, 3, Cin // 2, filters)) / 6
    for tag, c in (("1_", 2 * filters), ("2_", red)):
        p[f"{name}{tag}bn/gamma"] = rng.uniform(0.5, 1.5, c)
        p[f"{name}{tag}bn/beta"] = rng.standard_normal(c) * 0.1
        p[f"{name}{tag}bn/moving_mean"] = rng.standard_normal(c) * 0.1
        p[f"{name}{tag}bn/moving_variance"] = rng.uniform(0.5, 1.5, c)
    p[f"{name}2_conv/kernel"], p[f"{name}2_conv/bias"] = rng.standard_normal((1, 1, filters, red)) / 3, rng.standard_normal(red) * 0.1
    p[f"{name}3_conv/kernel"], p[f"{name}3_conv/bias"] = rng.standard_normal((1, 1, red, 2 * filters)) / 2, rng.standard_normal(2 * filters) * 0.1

    def bn(y, tag):
        g, b_, m, v_ = (p[f"{name}{tag}bn/{k}"] for k in ("gamma", "beta", "moving_mean", "moving_variance"))
        return g * (y - m) / np.sqrt(v_ + eps) + b_

    halves = [conv_loops(x[..., i * Cin // 2:(i + 1) * Cin // 2], p[f"{name}1_g{i + 1}_conv/kernel"], None, 1, (1, 1, 1, 1)) for i in range(2)]
    logits = np.maximum(bn(np.concatenate(halves, -1), "1_"), 0.0)
    l0, l1 = logits[..., :filters], logits[..., filters:]
    gap = (l0 + l1).mean((1, 2))                                                   # [B, filters]
    a = np.maximum(bn(gap @ p[f"{name}2_conv/kernel"][0, 0] + p[f"{name}2_conv/bias"], "2_"), 0.0)
    a = a @ p[f"{name}3_conv/kernel"][0, 0] + p[f"{name}3_conv/bias"]              # [B, 2 filters] = (radix, filters)
    a = a.reshape(B, 2, filters)
    a = np.exp(a - a.max(1, keepdims=True))
    a = a / a.sum(1, keepdims=True)                                                # softmax over the radix axis
    out = a[:, 0][:, None, None, :] * l0 + a[:, 1][:, None, None, :] * l1
    pt = {k: t32(v) for k, v in p.items()}
    got = kecam_ref.split_attention_conv2d(pt, name, t32(x), filters, 1, eps, groups=2).double().numpy()
    assert np.abs(got - out).max() <= 20 * TOL
    # strided tail: zero padding 1 + 3x3 / 2 average over ALL nine taps (the zeros count)
    outp = np.zeros((B, H + 2, W + 2, filters))
    outp[:, 1:-1, 1:-1] = out
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    want2 = np.stack([[outp[:, 2 * i:2 * i + 3, 2 * j:2 * j + 3].mean((1, 2)) for j in range(Wo)] for i in range(Ho)])
    want2 = want2.transpose(2, 0, 1, 3)
    got2 = kecam_ref.split_attention_conv2d(pt, name, t32(x), filters, 2, eps, groups=2).double().numpy()
    assert got2.shape == want2.shape and np.abs(got2 - want2).max() <= 20 * TOL


def test_scaled_standardized_conv_float64():
    """nfnets.py:64-70: mean, var = moments(kernel, axes=[0, 1, 2]) (biased), scale = rsqrt(max(var * fan_in, eps)) * (gain * gamma),
    kernel' = (kernel - mean) * scale; fan_in = kh * kw * Cin_g; then a plain convolution with torch padding k // 2 and the Conv2D bias"""
    rng = np.random.default_rng(2)
    for k, groups, stride in ((3, 2, 1), (1, 1, 1), (3, 1, 2)):
        Cin, O = 8, 6
        w = rng.standard_normal((k, k, Cin // groups, O)) * 0.3 + 0.05
        gain = rng.uniform(0.5, 1.5, O)
        bias = rng.standard_normal(O) * 0.1
        x = rng.standard_normal((2, 7, 6, Cin))
        gamma = 1.7881293296813965
        fan_in = k * k * (Cin // groups)
        wp = np.empty_like(w)
        for o in range(O):
            m = w[..., o].mean()
            var = ((w[..., o] - m) ** 2).mean()
            wp[..., o] = (w[..., o] - m) / math.sqrt(max(var * fan_in, 1e-5)) * gain[o] * gamma
        want = conv_loops(x, wp, bias, stride, (k // 2,) * 4, groups)
        p = {"c_conv/kernel": t32(w), "c_conv/gain": t32(gain), "c_conv/bias": t32(bias)}
        got = kecam_ref.std_conv(p, "c_", t32(x), k, stride, groups).double().numpy()
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (k, groups, stride)   # fp32 conv over <= 72 terms


@pytest.mark.parametrize("n,k,s", [(7, 3, 2), (8, 3, 2), (13, 5, 2), (12, 2, 2), (9, 3, 1), (25, 3, 2), (200, 3, 2)])
def test_tf_same_padding_rule(n, k, s):
    """TensorFlow "SAME": out = ceil(n / s), total = max((out - 1) s + k - n, 0), before = total // 2, after = total - before: the odd
    pixel goes to the bottom / right (the asymmetric stride-2 case of EfficientNetV1-B4, efficientnet_v2.py:80-85,125)"""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    assert R.same_pad(n, k, s) == (total // 2, total - total // 2)
    # and the windows it implies cover the input: first window starts at -before, last one ends at or after n - 1
    assert -(total // 2) + (out - 1) * s + k - 1 >= n - 1


def test_same_conv_and_pool_divisors_float64():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 7, 6, 4))
    w = rng.standard_normal((3, 3, 4, 5)) / 6
    # stride-2 SAME conv on a 7 x 6 map: rows pad (1, 1), columns pad (0, 1)
    want = conv_loops(x, w, None, 2, (1, 1, 0, 1))
    got = R.conv2d_same(t32(x), t32(w), None, 2).double().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= 5 * TOL
    # AveragePooling2D(2, 2, "same") on odd sizes: the divisor is the number of IN-IMAGE taps (resnet_rs_model.py:207-212)
    Ho, Wo = 4, 3
    want = np.zeros((2, Ho, Wo, 4))
    for i in range(Ho):
        for j in range(Wo):
            win = x[:, 2 * i:min(2 * i + 2, 7), 2 * j:min(2 * j + 2, 6)]
            want[:, i, j] = win.sum((1, 2)) / (win.shape[1] * win.shape[2])
    assert np.abs(R.avgpool_same(t32(x), 2, 2).double().numpy() - want).max() <= TOL
    # ZeroPadding2D(1) + MaxPool2D(3, 2): the border zeros TAKE PART in the max (gcvit feature.py:151-152) - visible on an all-negative map
    neg = -np.abs(x) - 0.1
    xp = np.zeros((2, 9, 8, 4))
    xp[:, 1:-1, 1:-1] = neg
    want = np.stack([[xp[:, 2 * i:2 * i + 3, 2 * j:2 * j + 3].max((1, 2)) for j in range(3)] for i in range(4)]).transpose(2, 0, 1, 3)
    got = R.maxpool_valid(t32(neg), 3, 2, (1, 1, 1, 1)).double().numpy()
    assert np.abs(got - want).max() <= TOL and (got[:, 0] == 0).all()            # the top row of windows sees the zero border
