"""HorNet on the HIP operator set - host-side mirror of kecam ``HorNet`` / ``HorNetBase``
(models/keras_cv_attention_models/hornet/hornet.py:84-107 gnconv, :110-124 block, :127-177 HorNet, :196-198 HorNetBase):
a ConvNeXt-shaped network whose token mixer is the recursive gated convolution.  ``HorNetBase-200x200`` is a member of
the reference's earlier ensembles (main.py:47); the global-filter ("GF") variants need an FFT and are not built.
"""
from typing import Dict

import torch

from . import ops
from .pipeline import keras_predict
from .synth import ParamGen

LN_EPS = 1e-5   # common_layers.py:8,215-219

CONFIGS = {      # hornet.py:127-135 defaults, :181 / :186-187 / :196-197 / :206-207
    "hornet_tiny": dict(num_blocks=(2, 3, 18, 2), embed_dim=64, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
    "hornet_small": dict(num_blocks=(2, 3, 18, 2), embed_dim=96, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
    "hornet_base": dict(num_blocks=(2, 3, 18, 2), embed_dim=128, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
    "hornet_large": dict(num_blocks=(2, 3, 18, 2), embed_dim=192, mlp_ratio=4, gn_split=(2, 3, 4, 5), scale=0.3333333),
}


def split_dims(c: int, gn_split: int):
    """hornet.py:87: [C / 2^(n-1), ..., C / 2, C]"""
    return [c // (2 ** i) for i in range(gn_split)][::-1]


def synth_params(cfg: Dict, seed: int, classes: int = 1) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    c = cfg["embed_dim"]
    g.conv("stem_conv", 4, 4, 3, c, bias=True, gain=1.0)
    g.ln("stem_ln", c)
    for si, nb in enumerate(cfg["num_blocks"]):
        st = f"stack{si + 1}_"
        if si > 0:
            g.ln(f"{st}ln", c)
            g.conv(f"{st}conv", 2, 2, c, 2 * c, bias=True, gain=1.0)
            c *= 2
        dims = split_dims(c, cfg["gn_split"][si])
        for bi in range(nb):
            b = f"{st}block{bi + 1}_"
            g.ln(f"{b}attn_ln", c)
            g.conv(f"{b}gnconv_pre_conv", 1, 1, c, 2 * c, bias=True, gain=1.0)
            g.dwconv(f"{b}gnconv_list_dw_conv", 7, sum(dims), bias=True, gain=9.0)     # x scale (1/3) in the graph
            for i in range(1, len(dims)):
                g.conv(f"{b}gnconv_pw{i}_conv", 1, 1, dims[i - 1], dims[i], bias=True, gain=1.0)
            g.conv(f"{b}gnconv_output_conv", 1, 1, c, c, bias=True, gain=1.0)
            g.raw(f"{b}1_gamma/weight", g._u((c,), 0.1, 0.4))        # trained layer scales (the 1e-6 init would mute the block)
            g.ln(f"{b}mlp_ln", c)
            g.dense(f"{b}mlp_Dense_0", c, int(c * cfg["mlp_ratio"]), gain=2.0)
            g.dense(f"{b}mlp_Dense_1", int(c * cfg["mlp_ratio"]), c)
            g.raw(f"{b}2_gamma/weight", g._u((c,), 0.1, 0.4))
    g.ln("pre_output_ln", c)
    g.dense("predictions", c, classes)
    return g.p


class _LN:
    def __init__(self, p, name, dev):
        self.g = p[f"{name}/gamma"].to(dev, torch.float32).contiguous()
        self.b = p[f"{name}/beta"].to(dev, torch.float32).contiguous()

    def __call__(self, x):
        return ops.layernorm(x, self.g, self.b, LN_EPS)


@keras_predict
class HorNet:
    def __init__(self, params: Dict[str, torch.Tensor], num_blocks, embed_dim, mlp_ratio=4, gn_split=(2, 3, 4, 5),
                 scale=0.3333333, classes: int = 1, first_strides: int = 2, device="cuda", classifier_activation: str = "default"):
        p, dev = params, device
        self.classes, self.first_strides = classes, first_strides
        self.head_act = classifier_activation
        self.stem = ops.make_conv_weight(p["stem_conv/kernel"], p["stem_conv/bias"], device=dev, pad_cin_to=8)
        self.stem_ln = _LN(p, "stem_ln", dev)
        self.stages = []
        c = embed_dim
        for si, nb in enumerate(num_blocks):
            st = f"stack{si + 1}_"
            down = None
            if si > 0:
                down = (_LN(p, f"{st}ln", dev), ops.make_conv_weight(p[f"{st}conv/kernel"], p[f"{st}conv/bias"], device=dev))
                c *= 2
            dims = split_dims(c, gn_split[si])
            blocks = []
            for bi in range(nb):
                b = f"{st}block{bi + 1}_"
                kpre, bpre = p[f"{b}gnconv_pre_conv/kernel"], p[f"{b}gnconv_pre_conv/bias"]
                d0 = dims[0]
                g1, g2 = p[f"{b}1_gamma/weight"], p[f"{b}2_gamma/weight"]
                blk = dict(
                    dims=dims, ln1=_LN(p, f"{b}attn_ln", dev), ln2=_LN(p, f"{b}mlp_ln", dev),
                    # tf.split of the 2C-channel projection (:88-89) = two Dense layers on the rows of its kernel
                    pre_pw=ops.make_conv_weight(kpre[..., :d0], bpre[:d0], device=dev),
                    pre_dw=ops.make_conv_weight(kpre[..., d0:], bpre[d0:], device=dev),
                    # `dw_list *= scale` (:95) folded into the depthwise filter and its bias
                    dw=ops.make_dw_weight(p[f"{b}gnconv_list_dw_conv/depthwise_kernel"] * scale, None, dev),
                    dwb=(p[f"{b}gnconv_list_dw_conv/bias"] * scale).to(dev, torch.float32).contiguous(),
                    pw=[ops.make_conv_weight(p[f"{b}gnconv_pw{i}_conv/kernel"], p[f"{b}gnconv_pw{i}_conv/bias"], device=dev)
                        for i in range(1, len(dims))],
                    # ChannelAffine layer scales (:116,121) folded into the rows of each branch's last layer
                    out=ops.make_conv_weight(p[f"{b}gnconv_output_conv/kernel"] * g1, p[f"{b}gnconv_output_conv/bias"] * g1, device=dev),
                    fc1=ops.make_dense_weight(p[f"{b}mlp_Dense_0/kernel"], p[f"{b}mlp_Dense_0/bias"], dev),
                    fc2=ops.make_dense_weight(p[f"{b}mlp_Dense_1/kernel"] * g2[None, :], p[f"{b}mlp_Dense_1/bias"] * g2, dev))
                blocks.append(blk)
            self.stages.append((down, blocks))
        self.head_ln = _LN(p, "pre_output_ln", dev)
        self.head_w = p["predictions/kernel"].t().contiguous().to(dev, torch.float32)
        self.head_b = p["predictions/bias"].to(dev, torch.float32)

    def _block(self, x, blk):
        dims = blk["dims"]
        a = blk["ln1"](x)
        pw = ops.conv2d(a, blk["pre_pw"])                                   # [B,H,W,d0]
        dwl = ops.dwconv2d(ops.conv2d(a, blk["pre_dw"]), blk["dw"], blk["dwb"], 7, 1, (3, 3, 3, 3))   # [B,H,W,sum(dims)]
        nn = ops.mul(pw, dwl, dims[0], 0, 0)                                # pw_first * dw_list[0]  (:98)
        off = dims[0]
        for i, w in enumerate(blk["pw"], start=1):                          # (:99-101)
            nn = ops.mul(ops.conv2d(nn, w), dwl, dims[i], 0, off)
            off += dims[i]
        x = ops.conv2d(nn, blk["out"], residual=x)                          # gamma1 * gnconv + x  (:116-118)
        return ops.mlp(x, blk["fc1"], blk["fc2"], act="gelu", residual=x, ln=(blk["ln2"].g, blk["ln2"].b, LN_EPS))

    def features(self, x, collect=None):
        """x: fp16 NHWC, RGB padded to 8 channels"""
        assert x.shape[-1] == 8
        s = self.first_strides * 2                                          # hornet.py:144
        y = self.stem_ln(ops.conv2d(x, self.stem, stride=s))
        for down, blocks in self.stages:
            if down is not None:
                y = ops.conv2d(down[0](y), down[1], stride=2)
            for blk in blocks:
                y = self._block(y, blk)
            if collect is not None:
                collect.append(y)
        return y

    def logits(self, x):
        n = self.head_ln                                                    # avg_pool -> pre_output_ln -> Dense  (:166-171), fp32
        return ops.gap_ln_dense_f32(self.features(x), n.g, n.b, LN_EPS, self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))
