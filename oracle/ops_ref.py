"""TEST INFRASTRUCTURE — CPU fp32 restatement of the TensorFlow/Keras primitives the reference's
scoring path is built from.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product (vip-cup-2022_amd/) never does.

PARITY UNPINNED: the arithmetic of these primitives lives in un-vendored third-party code
(TensorFlow/Keras, version not pinned by the reference: README.md:104 names only a docker tag), the
reference ships no tests, golden vectors or weights, and TensorFlow cannot be
imported in the build container (ModuleNotFoundError).  The semantics below are
restated from the published behaviour of those ops; each function cites the reference call site
that uses it.  They are cross-checked by independent NumPy re-derivations in tests/test_oracle_*.py.

Tensors are torch.float32, NHWC; weights are in Keras layouts (Conv2D HWIO, DepthwiseConv2D HWC1,
Dense [in,out]).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# activations — Keras names (kecam common_layers.py:48-71, tfimm layers/factory.py:6-13)
# ---------------------------------------------------------------------------------------------
def act(x, name):
    if name in (None, "none", "linear"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name in ("silu", "swish"):
        return x * torch.sigmoid(x)
    if name == "gelu":  # Keras "gelu" is the exact erf form (approximate=False)
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


# ---------------------------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------------------------
def zero_pad(x, pad):
    """tf.keras.layers.ZeroPadding2D(((top,bottom),(left,right))) on NHWC."""
    pt, pb, pl, pr = pad
    return F.pad(x, (0, 0, pl, pr, pt, pb))


def same_pad(size, k, s):
    """TensorFlow padding="SAME": total = max((ceil(n/s)-1)*s + k - n, 0), extra pixel at the end."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def conv2d(x, kernel_hwio, bias=None, stride=1, pad=(0, 0, 0, 0), groups=1):
    """tf.keras.layers.Conv2D(padding="valid") after an explicit zero pad (pt,pb,pl,pr).
    kernel HWIO [kh,kw,Cin/groups,Cout] (resnet_rs_model.py:71-82; kecam common_layers.py:230-248)."""
    s = (stride, stride) if isinstance(stride, int) else stride
    xp = zero_pad(x, pad).permute(0, 3, 1, 2)
    w = kernel_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(xp, w, bias, stride=s, groups=groups)
    return y.permute(0, 2, 3, 1).contiguous()


def conv2d_same(x, kernel_hwio, bias=None, stride=1, groups=1):
    """tf.keras.layers.Conv2D(padding="same")."""
    s = (stride, stride) if isinstance(stride, int) else stride
    kh, kw = kernel_hwio.shape[:2]
    pt, pb = same_pad(x.shape[1], kh, s[0])
    pl, pr = same_pad(x.shape[2], kw, s[1])
    return conv2d(x, kernel_hwio, bias, s, (pt, pb, pl, pr), groups)


def dwconv2d(x, kernel_hwc1, bias=None, stride=1, pad=(0, 0, 0, 0)):
    """tf.keras.layers.DepthwiseConv2D(padding="valid") after explicit pad; kernel [kh,kw,C,1]
    (gcvit/layers/feature.py:93; tfimm convnext.py:192-198)."""
    C = x.shape[-1]
    xp = zero_pad(x, pad).permute(0, 3, 1, 2)
    w = kernel_hwc1.permute(2, 3, 0, 1)  # [C,1,kh,kw]
    y = F.conv2d(xp, w, bias, stride=stride, groups=C)
    return y.permute(0, 2, 3, 1).contiguous()


def dense(x, kernel_io, bias=None):
    """tf.keras.layers.Dense: x @ kernel + bias over the last axis."""
    y = x @ kernel_io
    return y if bias is None else y + bias


def batchnorm(x, gamma, beta, mean, var, eps):
    """BatchNormalization in inference mode: gamma*(x-mean)/sqrt(var+eps)+beta."""
    return (x - mean) * (gamma / torch.sqrt(var + eps)) + beta


def layernorm(x, gamma, beta, eps):
    """tf.keras.layers.LayerNormalization(axis=-1): biased variance over the last axis."""
    m = x.mean(dim=-1, keepdim=True)
    v = ((x - m) ** 2).mean(dim=-1, keepdim=True)
    return (x - m) / torch.sqrt(v + eps) * gamma + beta


# ---------------------------------------------------------------------------------------------
# pooling
# ---------------------------------------------------------------------------------------------
def avgpool_same(x, k, s):
    """tf.keras.layers.AveragePooling2D(padding="same"): the divisor counts only in-image taps
    (resnet_rs_model.py:207-212; kecam aotnet.py:105)."""
    pt, pb = same_pad(x.shape[1], k, s)
    pl, pr = same_pad(x.shape[2], k, s)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    ones = F.pad(torch.ones_like(x[..., :1]).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    num = F.avg_pool2d(xp, k, s) * (k * k)
    den = F.avg_pool2d(ones, k, s) * (k * k)
    return (num / den).permute(0, 2, 3, 1).contiguous()


def avgpool_valid(x, k, s, pad=(0, 0, 0, 0)):
    """ZeroPadding2D + AveragePooling2D(padding="valid"): padded zeros count in the divisor
    (kecam resnest.py:63-65)."""
    xp = zero_pad(x, pad).permute(0, 3, 1, 2)
    return F.avg_pool2d(xp, k, s).permute(0, 2, 3, 1).contiguous()


def maxpool_valid(x, k, s, pad=(0, 0, 0, 0)):
    """ZeroPadding2D + MaxPool2D(padding="valid"): the max sees ZEROS in the border, not -inf
    (gcvit/layers/feature.py:151-152; kecam aotnet.py:329-330)."""
    xp = zero_pad(x, pad).permute(0, 3, 1, 2)
    return F.max_pool2d(xp, k, s).permute(0, 2, 3, 1).contiguous()


def global_avgpool(x):
    """GlobalAveragePooling2D: mean over H,W."""
    return x.mean(dim=(1, 2))


# ---------------------------------------------------------------------------------------------
# tf.image.resize(method="bicubic", antialias=False) — legacy ResizeBicubic kernel with
# half_pixel_centers=True (dataset/dataset.py:34).  Keys cubic a=-0.5; the fractional offset is
# quantised to a 1024-entry coefficient table; taps that fall outside the image get weight 0 and
# the remaining weights are renormalised to sum 1.
# ---------------------------------------------------------------------------------------------
_TABLE_SIZE = 1024


def _bicubic_table():
    """InitCoeffsTable(a = -0.5) of TensorFlow's resize_bicubic_op.cc: evaluated in double, stored as float."""
    a = -0.5
    t = torch.zeros((_TABLE_SIZE + 1) * 2, dtype=torch.float64)
    for i in range(_TABLE_SIZE + 1):
        x = i * 1.0 / _TABLE_SIZE
        t[i * 2] = ((a + 2) * x - (a + 3)) * x * x + 1
        x += 1.0
        t[i * 2 + 1] = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return t.to(torch.float32)


_TABLE = None


def bicubic_weights_and_indices(out_size, in_size):
    """Per output coordinate: 4 source indices (clamped) and 4 fp32 weights (GetWeightsAndIndices with
    HalfPixelScaler, use_keys_cubic=True: out-of-image taps get weight 0, then renormalise)."""
    global _TABLE
    if _TABLE is None:
        _TABLE = _bicubic_table()
    f32 = torch.float32
    scale = torch.tensor(in_size, dtype=f32) / torch.tensor(out_size, dtype=f32)
    idx = torch.zeros((out_size, 4), dtype=torch.long)
    wts = torch.zeros((out_size, 4), dtype=f32)
    for o in range(out_size):
        in_loc_f = (torch.tensor(o, dtype=f32) + 0.5) * scale - 0.5
        fl = torch.floor(in_loc_f)
        in_loc = int(fl)
        delta = in_loc_f - fl
        offset = int(torch.round(delta * _TABLE_SIZE))  # lrintf: round-half-even, same as torch.round
        w = [_TABLE[offset * 2 + 1], _TABLE[offset * 2], _TABLE[(_TABLE_SIZE - offset) * 2],
             _TABLE[(_TABLE_SIZE - offset) * 2 + 1]]
        s = torch.tensor(0.0, dtype=f32)
        for t in range(4):
            want = in_loc - 1 + t
            got = min(max(want, 0), in_size - 1)
            idx[o, t] = got
            wts[o, t] = w[t] if got == want else 0.0
            s = s + wts[o, t]
        if abs(float(s)) >= 1000.0 * 1.17549435e-38:
            wts[o] = wts[o] * (torch.tensor(1.0, dtype=f32) / s)
    return idx, wts


def resize_bicubic(img_hwc, out_h, out_w):
    """float32 [H,W,C] -> [out_h,out_w,C], no clamping (values may overshoot [0,255]).  Interpolate1D order:
    ((v0*w0 + v1*w1) + v2*w2) + v3*w3, first along x on the four source rows, then along y."""
    H, W, _ = img_hwc.shape
    iy, wy = bicubic_weights_and_indices(out_h, H)
    ix, wx = bicubic_weights_and_indices(out_w, W)
    rows = None
    for t in range(4):
        term = img_hwc[:, ix[:, t], :] * wx[None, :, t, None]          # [H,out_w,C]
        rows = term if rows is None else rows + term
    out = None
    for t in range(4):
        term = rows[iy[:, t], :, :] * wy[:, t, None, None]             # [out_h,out_w,C]
        out = term if out is None else out + term
    return out


def decode_resize_normalize(rgb_u8_hwc, out_h, out_w):
    """build_decoder.decode after the JPEG decode (dataset/dataset.py:31-38): cast f32 -> resize (always
    taken: `img_size != (200, 200)` compares a list with a tuple) -> /255."""
    img = torch.as_tensor(rgb_u8_hwc).to(torch.float32)
    return resize_bicubic(img, out_h, out_w) / 255.0


# ---------------------------------------------------------------------------------------------
# GCViT helpers (gcvit/layers/window.py:3-15, attention.py:39-50)
# ---------------------------------------------------------------------------------------------
def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C)


def window_reverse(windows, ws, H, W, C):
    x = windows.reshape(-1, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, H, W, C)


def relative_position_index(ws):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"), dim=0).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0)
    return (rel[:, :, 0] + ws - 1) * (2 * ws - 1) + (rel[:, :, 1] + ws - 1)
