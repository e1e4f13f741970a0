"""TEST INFRASTRUCTURE — CPU fp32 restatement of the reference ResNet-RS graph
(models/resnet_rs/resnet_rs_model.py).  PARITY UNPINNED (see oracle/ops_ref.py header): the reference
ships no weights, tests or golden vectors for this model and TensorFlow is unavailable.

BatchNorm is applied UN-folded here (the product folds it into the conv weights), so the parity test
also covers the folding.
"""
import torch

from . import ops_ref as R

# models/resnet_rs/block_args.py:1-44
BLOCK_ARGS = {50: [(64, 3), (128, 4), (256, 6), (512, 3)], 101: [(64, 3), (128, 4), (256, 23), (512, 3)],
              200: [(64, 3), (128, 24), (256, 36), (512, 3)]}          # block_args.py:2-25


def _fixed_padding(x, k):
    """model_utils.py:22-46"""
    total = k - 1
    beg = total // 2
    return R.zero_pad(x, (beg, total - beg, beg, total - beg))


def _conv_fixed(p, name, x, k, strides):
    """Conv2DFixedPadding (resnet_rs_model.py:64-84): strides>1 -> fixed_padding + VALID, else SAME; no bias."""
    if strides > 1:
        return R.conv2d(_fixed_padding(x, k), p[f"{name}/kernel"], None, strides)
    return R.conv2d_same(x, p[f"{name}/kernel"], None, 1)


def _bn(p, name, x, eps):
    return R.batchnorm(x, p[f"{name}/gamma"], p[f"{name}/beta"], p[f"{name}/moving_mean"],
                       p[f"{name}/moving_variance"], eps)


def stem(p, x, eps, act, first_strides):
    """STEM (:87-142)"""
    for i, s in zip(range(1, 5), (first_strides, 1, 1, 2)):
        x = R.act(_bn(p, f"stem_batch_norm_{i}", _conv_fixed(p, f"stem_conv_{i}", x, 3, s), eps), act)
    return x


def se(p, name, x):
    """SE (:145-183): GAP -> 1x1 conv (bias, relu) -> 1x1 conv (bias, sigmoid) -> multiply"""
    s = R.global_avgpool(x)[:, None, None, :]
    s = R.act(R.conv2d(s, p[name + "se_reduce/kernel"], p[name + "se_reduce/bias"]), "relu")
    s = R.act(R.conv2d(s, p[name + "se_expand/kernel"], p[name + "se_expand/bias"]), "sigmoid")
    return x * s


def bottleneck(p, name, x, strides, use_projection, eps, act, se_ratio):
    """BottleneckBlock (:186-282); survival_probability = 0 for RS-50 so the Dropout is absent."""
    shortcut = x
    if use_projection:
        if strides == 2:
            shortcut = R.avgpool_same(x, 2, 2)
            shortcut = _conv_fixed(p, name + "projection_conv", shortcut, 1, 1)
        else:
            shortcut = _conv_fixed(p, name + "projection_conv", x, 1, strides)
        shortcut = _bn(p, name + "projection_batch_norm", shortcut, eps)
    y = R.act(_bn(p, name + "batch_norm_1", _conv_fixed(p, name + "conv_1", x, 1, 1), eps), act)
    y = R.act(_bn(p, name + "batch_norm_2", _conv_fixed(p, name + "conv_2", y, 3, strides), eps), act)
    y = _bn(p, name + "batch_norm_3", _conv_fixed(p, name + "conv_3", y, 1, 1), eps)
    if 0 < se_ratio < 1:
        y = se(p, name, y)
    return R.act(y + shortcut, act)


def forward_features(p, x, depth=50, eps=1e-5, act="relu", se_ratio=0.25, first_strides=2, block_args=None,
                     collect=None):
    """ResNetRS (:329-513) up to the last block; x float32 NHWC [B,H,W,3] in [0,1].
    ``collect``: optional list receiving the stem output and every block output (diagnostics)."""
    x = stem(p, x, eps, act, first_strides)
    if collect is not None:
        collect.append(x)
    for gi, (f, reps) in enumerate(block_args or BLOCK_ARGS[depth]):
        for bi in range(reps):
            x = bottleneck(p, f"c{gi + 2}_block_{bi}_", x, (1 if gi == 0 else 2) if bi == 0 else 1, bi == 0, eps,
                           act, se_ratio)
            if collect is not None:
                collect.append(x)
    return x


def forward_logits(p, x, **kw):
    """head (:468-476): GAP -> (Dropout no-op) -> Dense; returns pre-activation logits [B, classes]."""
    f = forward_features(p, x, **kw)
    return R.dense(R.global_avgpool(f), p["predictions/kernel"], p["predictions/bias"])


def predict_logits(member, params, x):
    """Uniform entry used by tests / bench: member name -> logits."""
    assert member.startswith("resnet_rs") and int(member[9:]) in BLOCK_ARGS, member
    return forward_logits(params, x, depth=int(member[9:]))
