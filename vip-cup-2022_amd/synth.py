"""Seeded synthetic checkpoints.

The reference ships no trained weights (ckpts/ holds only ckpts.json; README.md:13), so every model
in this package can be instantiated from a deterministic synthetic checkpoint: a flat
``dict[str, torch.FloatTensor]`` keyed by the Keras variable names of the reference's layers
(e.g. ``stem_conv_1/kernel``, ``c2_block_0_batch_norm_1/moving_variance``) in Keras layouts
(Conv2D HWIO, DepthwiseConv2D [kh,kw,C,1], Dense [in,out]).  A real ``.h5`` importer only has to
produce the same dictionary.

Distributions (chosen so fp32 activations stay O(1..100) through the whole depth, which keeps the
fp16-vs-fp32 parity check meaningful): conv / dense kernels N(0, gain/fan_in); BatchNorm
gamma~U(.5,1.5), beta~N(0,.1), moving_mean~N(0,.1), moving_variance~U(.5,1.5); LayerNorm
gamma = 1+N(0,.1), beta~N(0,.1); biases N(0,.05).
"""
import math
from typing import Dict

import torch


class ParamGen:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)
        self.p: Dict[str, torch.Tensor] = {}

    def _n(self, shape, std=1.0, mean=0.0):
        return torch.randn(shape, generator=self.g, dtype=torch.float32) * std + mean

    def _u(self, shape, lo, hi):
        return torch.rand(shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo

    def conv(self, name, kh, kw, cin_g, cout, bias=False, gain=2.0):
        fan_in = kh * kw * cin_g
        self.p[f"{name}/kernel"] = self._n((kh, kw, cin_g, cout), math.sqrt(gain / fan_in))
        if bias:
            self.p[f"{name}/bias"] = self._n((cout,), 0.05)

    def dwconv(self, name, k, c, bias=False, gain=2.0):
        self.p[f"{name}/depthwise_kernel"] = self._n((k, k, c, 1), math.sqrt(gain / (k * k)))
        if bias:
            self.p[f"{name}/bias"] = self._n((c,), 0.05)

    def dense(self, name, cin, cout, bias=True, gain=1.0):
        self.p[f"{name}/kernel"] = self._n((cin, cout), math.sqrt(gain / cin))
        if bias:
            self.p[f"{name}/bias"] = self._n((cout,), 0.05)

    def bn(self, name, c):
        self.p[f"{name}/gamma"] = self._u((c,), 0.5, 1.5)
        self.p[f"{name}/beta"] = self._n((c,), 0.1)
        self.p[f"{name}/moving_mean"] = self._n((c,), 0.1)
        self.p[f"{name}/moving_variance"] = self._u((c,), 0.5, 1.5)

    def ln(self, name, c):
        self.p[f"{name}/gamma"] = self._n((c,), 0.1, 1.0)
        self.p[f"{name}/beta"] = self._n((c,), 0.1)

    def raw(self, name, tensor):
        self.p[name] = tensor.to(torch.float32)

    def trunc_normal(self, name, shape, std):
        t = self._n(shape, std)
        self.p[name] = t.clamp_(-2 * std, 2 * std)


def fold_bn(kernel_hwio, gamma, beta, mean, var, eps, conv_bias=None):
    """Fold an inference BatchNorm into the preceding conv (kecam model_surgery.py:407-421):
    w' = w * gamma/sqrt(var+eps) (per output channel), b' = (b - mean) * gamma/sqrt(var+eps) + beta."""
    s = gamma / torch.sqrt(var + eps)
    w = kernel_hwio * s  # broadcasts over the last (output-channel) axis
    b0 = conv_bias if conv_bias is not None else torch.zeros_like(mean)
    return w, (b0 - mean) * s + beta
