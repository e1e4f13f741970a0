"""ResNet-RS on the HIP operator set — host-side mirror of the reference constructors
``ResNetRS`` / ``ResNetRS50`` (models/resnet_rs/resnet_rs_model.py:329-540) with the same argument
meaning (``depth``, ``se_ratio``, ``classes``, ``first_strides``, ``bn_epsilon``, ``activation``).

Every Conv2D+BatchNormalization(+Activation) group of the reference is ONE implicit-GEMM launch (BN is
folded into the weights at load time, kecam model_surgery.py:407-421); the SE excite multiply, the
residual Add and the block's output activation are one fused elementwise launch.
"""
from typing import Dict, List

import torch

from . import ops
from .pipeline import keras_predict
from .synth import ParamGen, fold_bn

# models/resnet_rs/block_args.py:1-44 (depth -> [(input_filters, num_repeats)])
BLOCK_ARGS = {
    50: [(64, 3), (128, 4), (256, 6), (512, 3)],
    101: [(64, 3), (128, 4), (256, 23), (512, 3)],
    152: [(64, 3), (128, 8), (256, 36), (512, 3)],
    200: [(64, 3), (128, 24), (256, 36), (512, 3)],
    270: [(64, 4), (128, 29), (256, 53), (512, 4)],
    350: [(64, 4), (128, 36), (256, 72), (512, 4)],
    420: [(64, 4), (128, 44), (256, 87), (512, 4)],
}


def fixed_padding(kernel_size: int):
    """model_utils.py:22-46: (pad_beg, pad_end) = ((k-1)//2, k-1-(k-1)//2) on both spatial axes."""
    total = kernel_size - 1
    beg = total // 2
    return (beg, total - beg, beg, total - beg)


def synth_params(depth: int = 50, seed: int = 1006, classes: int = 1, se_ratio: float = 0.25,
                 block_args: List = None) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint with the reference's Keras variable names (resnet_rs_model.py:97-282,468-476)."""
    g = ParamGen(seed)
    widths = [(3, 32), (32, 32), (32, 64), (64, 64)]
    for i, (ci, co) in enumerate(widths, 1):
        g.conv(f"stem_conv_{i}", 3, 3, ci, co)
        g.bn(f"stem_batch_norm_{i}", co)
    cin = 64
    for gi, (f, reps) in enumerate(block_args or BLOCK_ARGS[depth]):
        for bi in range(reps):
            n = f"c{gi + 2}_block_{bi}_"
            if bi == 0:
                g.conv(n + "projection_conv", 1, 1, cin, 4 * f)
                g.bn(n + "projection_batch_norm", 4 * f)
            g.conv(n + "conv_1", 1, 1, cin, f)
            g.bn(n + "batch_norm_1", f)
            g.conv(n + "conv_2", 3, 3, f, f)
            g.bn(n + "batch_norm_2", f)
            # damped residual branch (the reference zero-initialises the block-final BN gamma; an undamped
            # random 16-block ReLU net is chaotic: it doubles any perturbation per block)
            g.conv(n + "conv_3", 1, 1, f, 4 * f, gain=0.25)
            g.bn(n + "batch_norm_3", 4 * f)
            if 0 < se_ratio < 1:
                r = max(1, int(f * 4 * se_ratio))
                g.conv(n + "se_reduce", 1, 1, 4 * f, r, bias=True)
                g.conv(n + "se_expand", 1, 1, r, 4 * f, bias=True, gain=1.0)
            cin = 4 * f
    g.dense("predictions", cin, classes, gain=1.0)
    return g.p


@keras_predict
class ResNetRS:
    """Inference-only ResNet-RS.  ``params`` is a checkpoint dict (see synth_params)."""

    def __init__(self, params: Dict[str, torch.Tensor], depth: int = 50, bn_epsilon: float = 1e-5,
                 activation: str = "relu", se_ratio: float = 0.25, classes: int = 1, first_strides: int = 2,
                 block_args: List = None, device="cuda", classifier_activation: str = "default"):
        self.act = activation
        self.head_act = classifier_activation          # resnet_rs_model.py:337,474-476
        self.first_strides = first_strides
        self.classes = classes
        self.device = device
        self.block_args = block_args or BLOCK_ARGS[depth]
        self.has_se = 0 < se_ratio < 1
        p = params

        def cbn(conv, bn, pad_cin=None, hilo=False):
            w, b = fold_bn(p[f"{conv}/kernel"], p[f"{bn}/gamma"], p[f"{bn}/beta"], p[f"{bn}/moving_mean"],
                           p[f"{bn}/moving_variance"], bn_epsilon)
            return ops.make_conv_weight(w, b, device=device, pad_cin_to=pad_cin, hilo=hilo)

        self.stem = [cbn(f"stem_conv_{i}", f"stem_batch_norm_{i}", 8 if i == 1 else None) for i in range(1, 5)]
        self.blocks = []
        for gi, (f, reps) in enumerate(self.block_args):
            for bi in range(reps):
                n = f"c{gi + 2}_block_{bi}_"
                blk = {"stride": (1 if gi == 0 else 2) if bi == 0 else 1, "proj": None}
                # the 1x1 convolutions of the two high-resolution groups (>= 25 x 25 pixels) with K <= 256 carry two-term
                # weights (HBM-bound streaming-kernel shapes; see ops.make_conv_weight / DESIGN.md Numerics)
                hl = gi <= 1
                if bi == 0:
                    blk["proj"] = cbn(n + "projection_conv", n + "projection_batch_norm", hilo=hl)
                blk["c1"] = cbn(n + "conv_1", n + "batch_norm_1", hilo=hl)
                blk["c2"] = cbn(n + "conv_2", n + "batch_norm_2")
                blk["c3"] = cbn(n + "conv_3", n + "batch_norm_3", hilo=hl)
                if self.has_se:
                    blk["se_r"] = ops.make_conv_weight(p[n + "se_reduce/kernel"], p[n + "se_reduce/bias"], device=device)
                    blk["se_e"] = ops.make_conv_weight(p[n + "se_expand/kernel"], p[n + "se_expand/bias"], device=device)
                self.blocks.append(blk)
        self.head_w = p["predictions/kernel"].t().contiguous().to(device=device, dtype=torch.float32)
        self.head_b = p["predictions/bias"].to(device=device, dtype=torch.float32)

    # resnet_rs_model.py:186-282
    def _bottleneck(self, x, blk):
        s = blk["stride"]
        shortcut = x
        if blk["proj"] is not None:
            if s == 2:
                # AveragePooling2D(2,2,"same") then 1x1 stride-1 projection (:206-218)
                H, W = x.shape[1], x.shape[2]
                sc = ops.pool2d(x, 2, 2, (0, H % 2, 0, W % 2), ops.POOL_AVG_VALID)
                shortcut = ops.conv2d(sc, blk["proj"])
            else:
                shortcut = ops.conv2d(x, blk["proj"], stride=s)
        y = ops.conv2d(x, blk["c1"], act=self.act)
        y = ops.conv2d(y, blk["c2"], stride=s, pad=(1, 1, 1, 1) if s == 1 else fixed_padding(3), act=self.act)
        y = ops.conv2d(y, blk["c3"])
        scale = None
        if self.has_se:
            # SE (:145-183): squeeze -> 1x1 relu -> 1x1 sigmoid
            scale = ops.se_gate(y, blk["se_r"], blk["se_e"], "relu", "sigmoid")
        return ops.scale_add_act(y, scale, shortcut, self.act)

    def features(self, x: torch.Tensor, collect=None) -> torch.Tensor:
        """x: fp16 NHWC with the RGB axis zero-padded to 8 channels."""
        assert x.shape[-1] == 8, "input must be NHWC fp16 with channels padded to 8"
        fs = self.first_strides
        # STEM (:87-142): strides fs,1,1,2 ; stride>1 uses fixed_padding + VALID, stride 1 uses SAME
        y = ops.conv2d(x, self.stem[0], stride=fs, pad=fixed_padding(3) if fs > 1 else (1, 1, 1, 1), act=self.act)
        y = ops.conv2d(y, self.stem[1], pad=(1, 1, 1, 1), act=self.act)
        y = ops.conv2d(y, self.stem[2], pad=(1, 1, 1, 1), act=self.act)
        y = ops.conv2d(y, self.stem[3], stride=2, pad=fixed_padding(3), act=self.act)
        if collect is not None:
            collect.append(y)
        for blk in self.blocks:
            y = self._bottleneck(y, blk)
            if collect is not None:
                collect.append(y)
        return y

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 ``[B, classes]`` pre-activation outputs of the ``predictions`` Dense."""
        return ops.gap_dense_f32(self.features(x), self.head_w, self.head_b)

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        """``model.predict`` equivalent (main.py:109): sigmoid for 1 class, softmax otherwise (host, B x classes floats)."""
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))


def ResNetRS50(params, classes=1, first_strides=2, device="cuda", classifier_activation="default"):
    """resnet_rs_model.py:516-540"""
    return ResNetRS(params, depth=50, classes=classes, first_strides=first_strides, device=device,
                    classifier_activation=classifier_activation)
