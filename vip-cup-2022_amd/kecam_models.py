"""The four keras_cv_attention_models ("kecam") ensemble members on the HIP operator set — host-side mirrors
of ``ResNest50`` (resnest/resnest.py:76-77 on aotnet/aotnet.py:284-377), ``EfficientNetV2T``
(efficientnet/efficientnet_v2.py:268-275), ``EfficientNetV1B4`` (efficientnet/efficientnet_v1.py:68-73) and
``ECA_NFNetL0`` (nfnets/nfnets.py:316-320), with the shared blocks of common_layers.py.

Load-time folding (all exact in fp32, then cast to fp16 once):
  * every BatchNormalization into the conv before it (model_surgery.py:407-421);
  * ScaledStandardizedConv2D weight standardisation, its gain, the activation gamma, and for NFNet blocks the
    pre-activation beta / the alpha * attn_gain of the residual branch (nfnets.py:64-70,137,160-167);
  * the radix-2 r-softmax of ResNeSt as sigmoid(+-(a0 - a1)) by differencing the last attention conv
    (resnest.py:16-24); the sum of the two radix splits before the squeeze by duplicating the weight rows;
  * squeeze-excite hidden widths that are not multiples of 8 are zero-padded (padded units output act(0)=0
    for relu / swish and meet zero rows in the next conv);
  * ECA's Conv1D over channels as a banded [C,C] matrix for the GEMM kernel (common_layers.py:335-353).
"""
import math
import os
from typing import Dict

import torch

from . import ops
from .pipeline import keras_predict
from .synth import ParamGen, fold_bn

PAD1 = (1, 1, 1, 1)


def make_divisible(vv, divisor=4, min_value=None, limit_round_down=0.9):
    """common_layers.py:398-406"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(vv + divisor / 2) // divisor * divisor)
    if new_v < limit_round_down * vv:
        new_v += divisor
    return new_v


def _r8(n):
    return (n + 7) // 8 * 8


def same_pad(size, k, s):
    """TensorFlow SAME: the odd pixel goes after"""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def _cbn(p, conv, bn, eps, dev, pad_cin=None, conv_bias=None, hilo=False):
    w, b = fold_bn(p[f"{conv}conv/kernel"], p[f"{bn}bn/gamma"], p[f"{bn}bn/beta"], p[f"{bn}bn/moving_mean"],
                   p[f"{bn}bn/moving_variance"], eps, conv_bias)
    return ops.make_conv_weight(w, b, device=dev, pad_cin_to=pad_cin, hilo=hilo)


def _head(p, dev):
    return (p["predictions/kernel"].t().contiguous().to(dev, torch.float32), p["predictions/bias"].to(dev, torch.float32))


@keras_predict
class _Base:
    classes = 1

    def logits(self, x):
        return ops.gap_dense_f32(self.features(x), self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))


# ------------------------------------------------------------------------------------------------
# ResNest50
# ------------------------------------------------------------------------------------------------
RESNEST50 = dict(num_blocks=(3, 4, 6, 3), out_channels=(256, 512, 1024, 2048), strides=(1, 2, 2, 2), stem_width=64)
RESNEST200 = dict(RESNEST50, num_blocks=(3, 24, 36, 3), stem_width=128)        # resnest.py:84-85
# ResNetD (resnet_family/resnet_deep.py:13-36): the same AotNet skeleton (deep stem, "avg" shortcut) with a plain
# 3x3 conv + BN + ReLU where ResNeSt has its split attention (aotnet.py:78-81,89-91)
RESNET200D = dict(RESNEST50, num_blocks=(3, 24, 36, 3), attn=None)


def resnest_synth_params(seed: int, classes: int = 1, cfg=RESNEST50) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    sw = cfg["stem_width"]
    for i, (ci, co) in enumerate([(3, sw // 2), (sw // 2, sw // 2), (sw // 2, sw)], 1):
        g.conv(f"stem_{i}_conv", 3, 3, ci, co)
        g.bn(f"stem_{i}_bn" if i < 3 else "stem_bn", co)
    cin = sw
    for si, (nb, oc, st) in enumerate(zip(cfg["num_blocks"], cfg["out_channels"], cfg["strides"])):
        hid = int(oc * 0.25)
        inter = max(hid * 2 // 4, 32)
        for bi in range(nb):
            n = f"stack{si + 1}_block{bi + 1}_"
            s = st if bi == 0 else 1
            if bi == 0 and (s != 1 or cin != oc):
                g.conv(f"{n}shortcut_conv", 1, 1, cin, oc)
                g.bn(f"{n}shortcut_bn", oc)
            g.conv(f"{n}deep_1_conv", 1, 1, cin, hid)
            g.bn(f"{n}deep_1_bn", hid)
            if cfg.get("attn", "sa") is None:
                g.conv(f"{n}deep_2_conv", 3, 3, hid, hid)
                g.bn(f"{n}deep_2_bn", hid)
            else:
                for r in (1, 2):
                    g.conv(f"{n}deep_2_sa_1_g{r}_conv", 3, 3, hid // 2, hid)
                g.bn(f"{n}deep_2_sa_1_bn", 2 * hid)
                g.conv(f"{n}deep_2_sa_2_conv", 1, 1, hid, inter, bias=True)
                g.bn(f"{n}deep_2_sa_2_bn", inter)
                g.conv(f"{n}deep_2_sa_3_conv", 1, 1, inter, 2 * hid, bias=True, gain=1.0)
            g.conv(f"{n}deep_3_conv", 1, 1, hid, oc, gain=0.25)
            g.bn(f"{n}3_bn", oc)
            cin = oc
    g.dense("predictions", cin, classes)
    return g.p


class ResNest(_Base):
    def __init__(self, params, cfg=RESNEST50, classes=1, eps=1e-5, first_strides=2, device="cuda", classifier_activation="default"):
        p, dev = params, device
        self.head_act = classifier_activation
        self.cfg, self.classes, self.first_strides = cfg, classes, first_strides
        self.stem = [_cbn(p, "stem_1_", "stem_1_", eps, dev, pad_cin=8), _cbn(p, "stem_2_", "stem_2_", eps, dev),
                     _cbn(p, "stem_3_", "stem_", eps, dev)]
        self.blocks = []
        cin = cfg["stem_width"]
        for si, (nb, oc, st) in enumerate(zip(cfg["num_blocks"], cfg["out_channels"], cfg["strides"])):
            hid = int(oc * 0.25)
            for bi in range(nb):
                n = f"stack{si + 1}_block{bi + 1}_"
                s = st if bi == 0 else 1
                blk = {"stride": s, "hid": hid, "sc": None}
                hl = si <= 1                       # two-term weights on the >= 25 x 25-pixel stages (K <= 256 layers only)
                if bi == 0 and (s != 1 or cin != oc):
                    blk["sc"] = _cbn(p, f"{n}shortcut_", f"{n}shortcut_", eps, dev, hilo=hl)
                blk["d1"] = _cbn(p, f"{n}deep_1_", f"{n}deep_1_", eps, dev, hilo=hl)
                blk["d3"] = _cbn(p, f"{n}deep_3_", f"{n}3_", eps, dev, hilo=hl)
                if cfg.get("attn", "sa") is None:                       # ResNetD: conv3x3 (stride here) + BN + ReLU
                    blk["d2"] = _cbn(p, f"{n}deep_2_", f"{n}deep_2_", eps, dev)
                    self.blocks.append(blk)
                    cin = oc
                    continue
                # the two radix convs = one grouped conv (groups=2) with the "1_" BN folded per output channel
                kcat = torch.cat([p[f"{n}deep_2_sa_1_g1_conv/kernel"], p[f"{n}deep_2_sa_1_g2_conv/kernel"]], dim=3)
                w, b = fold_bn(kcat, *[p[f"{n}deep_2_sa_1_bn/{k}"] for k in ("gamma", "beta", "moving_mean", "moving_variance")], eps)
                blk["sa1"] = ops.make_conv_weight(w, b, groups=2, device=dev)
                # squeeze = mean(split0 + split1): duplicate the rows of the first attention conv
                w2, b2 = fold_bn(p[f"{n}deep_2_sa_2_conv/kernel"],
                                 *[p[f"{n}deep_2_sa_2_bn/{k}"] for k in ("gamma", "beta", "moving_mean", "moving_variance")],
                                 eps, p[f"{n}deep_2_sa_2_conv/bias"])
                blk["sa2"] = ops.make_conv_weight(torch.cat([w2, w2], dim=2), b2, device=dev)
                # r-softmax over the radix pair = sigmoid(a0 - a1), sigmoid(a1 - a0)
                w3, b3 = p[f"{n}deep_2_sa_3_conv/kernel"], p[f"{n}deep_2_sa_3_conv/bias"]
                wd, bd = w3[..., :hid] - w3[..., hid:], b3[:hid] - b3[hid:]
                blk["sa3"] = ops.make_conv_weight(torch.cat([wd, -wd], dim=3), torch.cat([bd, -bd]), device=dev)
                self.blocks.append(blk)
                cin = oc
        self.stage_ends = list(torch.tensor(cfg["num_blocks"]).cumsum(0).tolist())
        self.head_w, self.head_b = _head(p, dev)

    def features(self, x, collect=None):
        assert x.shape[-1] == 8
        y = ops.conv2d(x, self.stem[0], stride=self.first_strides, pad=PAD1, act="relu")
        y = ops.conv2d(y, self.stem[1], pad=PAD1, act="relu")
        y = ops.conv2d(y, self.stem[2], pad=PAD1, act="relu")                 # stem_3 conv + stem_bn + relu
        y = ops.pool2d(y, 3, 2, PAD1, ops.POOL_MAX_ZEROPAD)
        for i, blk in enumerate(self.blocks):
            s = blk["stride"]
            if blk["sc"] is not None:
                H, W = y.shape[1], y.shape[2]
                sc = ops.pool2d(y, 2, 2, (0, H % 2, 0, W % 2), ops.POOL_AVG_VALID) if s > 1 else y
                sc = ops.conv2d(sc, blk["sc"])
            else:
                sc = y
            d = ops.conv2d(y, blk["d1"], act="relu")
            if "d2" in blk:                                                     # ResNetD
                d = ops.conv2d(d, blk["d2"], stride=s, pad=PAD1, act="relu")
                y = ops.conv2d(d, blk["d3"], residual=sc, act_post="relu")
                if collect is not None and (i + 1) in self.stage_ends:
                    collect.append(y)
                continue
            lg = ops.conv2d(d, blk["sa1"], pad=PAD1, act="relu")                # [B,H,W,2*hid]
            a = ops.se_gate(lg, blk["sa2"], blk["sa3"], "relu", "sigmoid")      # split: the r-softmax weights keep ~22 bits
            d = ops.radix_combine(lg, a, 2)
            if s > 1:
                d = ops.pool2d(d, 3, 2, PAD1, ops.POOL_AVG_FULL)
            y = ops.conv2d(d, blk["d3"], residual=sc, act_post="relu")
            if collect is not None and (i + 1) in self.stage_ends:
                collect.append(y)
        return y


# ------------------------------------------------------------------------------------------------
# EfficientNetV2T / EfficientNetV1B4
# ------------------------------------------------------------------------------------------------
EFFNET = {
    "EfficientNetV2T": dict(expands=[1, 4, 4, 4, 6, 6], out_channels=[24, 40, 48, 104, 128, 208], depthes=[2, 4, 4, 6, 9, 14],
                            strides=[1, 2, 2, 2, 1, 2], se_ratios=[0, 0, 0, 0.25, 0.25, 0.25], kernel_sizes=[3] * 6,
                            first_conv_filter=24, output_conv_filter=1024, is_torch_mode=True),
    # get_expanded_width_depth(1.4, 1.8) (efficientnet_v1.py:9-18,69)
    "EfficientNetV1B4": dict(expands=[1, 6, 6, 6, 6, 6, 6], out_channels=[ii * 1.4 for ii in [16, 24, 40, 80, 112, 192, 320]],
                             depthes=[int(math.ceil(ii * 1.8)) for ii in [1, 2, 2, 3, 3, 4, 1]],
                             strides=[1, 2, 2, 2, 1, 2, 1], se_ratios=[0.25] * 7, kernel_sizes=[3, 3, 5, 3, 5, 5, 3],
                             first_conv_filter=32 * 1.4, output_conv_filter=1280 * 1.4, is_torch_mode=False),
    # efficientnet_v2.py:300-325 (members of the earlier ensembles): TF-SAME padding / BN eps 1e-3 like every non-"T" variant
    "EfficientNetV2M": dict(expands=[1, 4, 4, 4, 6, 6, 6], out_channels=[24, 48, 80, 160, 176, 304, 512],
                            depthes=[3, 5, 5, 7, 14, 18, 5], strides=[1, 2, 2, 2, 1, 2, 1],
                            se_ratios=[0, 0, 0, 0.25, 0.25, 0.25, 0.25], kernel_sizes=[3] * 7, first_conv_filter=24,
                            output_conv_filter=1280, is_torch_mode=False),
    "EfficientNetV2L": dict(expands=[1, 4, 4, 4, 6, 6, 6], out_channels=[32, 64, 96, 192, 224, 384, 640],
                            depthes=[4, 7, 7, 10, 19, 25, 7], strides=[1, 2, 2, 2, 1, 2, 1],
                            se_ratios=[0, 0, 0, 0.25, 0.25, 0.25, 0.25], kernel_sizes=[3] * 7, first_conv_filter=32,
                            output_conv_filter=1280, is_torch_mode=False),
}


def _effnet_blocks(c):
    """yields (name, cin, hidden, out, stride, expand, shortcut, k, se_reduction, fused)"""
    pre = make_divisible(c["first_conv_filter"], 8)
    for i, (e, oc, d, s, se, k) in enumerate(zip(c["expands"], c["out_channels"], c["depthes"], c["strides"],
                                                 c["se_ratios"], c["kernel_sizes"])):
        out = make_divisible(oc, 8)
        for b in range(d):
            st = s if b == 0 else 1
            hidden = make_divisible(pre * e, 8)
            red = make_divisible(hidden * (se / e), 1) if se > 0 else 0     # se_module(divisor=1) (:90-94)
            yield (f"stack_{i}_block{b}_", pre, hidden, out, st, e, out == pre and st == 1, k, red, se == 0, i)
            pre = out


def effnet_synth_params(name: str, seed: int, classes: int = 1) -> Dict[str, torch.Tensor]:
    c = EFFNET[name]
    g = ParamGen(seed)
    boost = all(se > 0 for se in c["se_ratios"])   # V1: every block is gated -> compensate (see below)
    stem = make_divisible(c["first_conv_filter"], 8)
    g.conv("stem_conv", 3, 3, 3, stem)
    g.bn("stem_bn", stem)
    last = stem
    for (n, cin, hid, out, st, e, sc, k, red, fused, _) in _effnet_blocks(c):
        if fused and e != 1:
            g.conv(f"{n}sortcut_conv", 3, 3, cin, hid)
            g.bn(f"{n}sortcut_bn", hid)
        elif e != 1:
            g.conv(f"{n}sortcut_conv", 1, 1, cin, hid)
            g.bn(f"{n}sortcut_bn", hid)
        if not fused:
            g.dwconv(f"{n}MB_dw_", k, hid)
            g.bn(f"{n}MB_dw_bn", hid)
        if red > 0:
            g.conv(f"{n}se_1_conv", 1, 1, hid, red, bias=True)
            g.conv(f"{n}se_2_conv", 1, 1, red, hid, bias=True, gain=1.0)
        if fused and e == 1:
            g.conv(f"{n}fu_conv", 3, 3, hid, out)
            g.bn(f"{n}fu_bn", out)
        else:
            # the SE gate (~0.5) quarters the variance of the branch: compensate so the signal survives 30+ blocks
            g.conv(f"{n}MB_pw_conv", 1, 1, hid, out, gain=(0.5 if sc else 4.0) if (red > 0 and boost) else (0.25 if sc else 1.0))
            g.bn(f"{n}MB_pw_bn", out)
        last = out
    post = make_divisible(c["output_conv_filter"], 8)
    g.conv("post_conv", 1, 1, last, post)
    g.bn("post_bn", post)
    g.dense("predictions", post, classes)
    return g.p


class EfficientNet(_Base):
    def __init__(self, params, name: str, classes=1, first_strides=2, device="cuda", classifier_activation="default"):
        p, dev = params, device
        self.head_act = classifier_activation
        c = EFFNET[name]
        self.c, self.classes, self.first_strides = c, classes, first_strides
        self.torch_mode = c["is_torch_mode"]
        eps = 1e-5 if self.torch_mode else 1e-3
        self.stem = _cbn(p, "stem_", "stem_", eps, dev, pad_cin=8)
        self.blocks = []
        div = first_strides                               # cumulative stride of the block's input
        for (n, cin, hid, out, st, e, sc, k, red, fused, stage) in _effnet_blocks(c):
            blk = dict(stride=st, expand=e, shortcut=sc, k=k, fused=fused, stage=stage, exp=None, dw=None, se=None)
            # 1x1 convolutions with Cin <= 256 on the high-resolution stages (input stride <= 8: >= 25 x 25 pixels, i.e. the
            # HBM-bound streaming-kernel shapes at batch 256) carry two-term weights: their fp16 weight rounding is 2/3 of
            # EfficientNetV1-B4's weight-induced logit error (Numerics), and there the second MFMA is nearly free
            if e != 1:
                blk["exp"] = _cbn(p, f"{n}sortcut_", f"{n}sortcut_", eps, dev, hilo=(div <= 8))
            div *= st
            if not fused:
                s_ = p[f"{n}MB_dw_bn/gamma"] / torch.sqrt(p[f"{n}MB_dw_bn/moving_variance"] + eps)
                b = p[f"{n}MB_dw_bn/beta"] - p[f"{n}MB_dw_bn/moving_mean"] * s_
                blk["dw"] = (ops.make_dw_weight(p[f"{n}MB_dw_/depthwise_kernel"], s_, dev), b.to(dev, torch.float32).contiguous())
            if red > 0:
                blk["se"] = (ops.make_conv_weight(p[f"{n}se_1_conv/kernel"], p[f"{n}se_1_conv/bias"], device=dev,
                                                  pad_cout_to=_r8(red)),
                             ops.make_conv_weight(p[f"{n}se_2_conv/kernel"], p[f"{n}se_2_conv/bias"], device=dev,
                                                  pad_cin_to=_r8(red)))
            if fused and e == 1:
                blk["out"] = _cbn(p, f"{n}fu_", f"{n}fu_", eps, dev)
            else:
                blk["out"] = _cbn(p, f"{n}MB_pw_", f"{n}MB_pw_", eps, dev, hilo=(red == 0 and div <= 8))   # gated (SE) layers cannot
            self.blocks.append(blk)
        self.post = _cbn(p, "post_", "post_", eps, dev)
        self.head_w, self.head_b = _head(p, dev)

    def _pad(self, size_hw, k, s):
        if self.torch_mode:
            return (k // 2,) * 4
        pt, pb = same_pad(size_hw[0], k, s)
        pl, pr = same_pad(size_hw[1], k, s)
        return (pt, pb, pl, pr)

    def features(self, x, collect=None):
        assert x.shape[-1] == 8
        act = "silu"
        y = ops.conv2d(x, self.stem, stride=self.first_strides, pad=self._pad(x.shape[1:3], 3, self.first_strides), act=act)
        for i, blk in enumerate(self.blocks):                        # inverted_residual_block (:47-108)
            inp, s, k = y, blk["stride"], blk["k"]
            if blk["fused"]:
                pad = self._pad(y.shape[1:3], 3, s)
                if blk["expand"] != 1:
                    h = ops.conv2d(y, blk["exp"], stride=s, pad=pad, act=act)
                    y = ops.conv2d(h, blk["out"], residual=inp if blk["shortcut"] else None)
                else:
                    y = ops.conv2d(y, blk["out"], stride=s, pad=pad, act=act, residual=inp if blk["shortcut"] else None)
            else:
                pad = self._pad(y.shape[1:3], k, s)
                if blk["se"] is not None and os.environ.get("VIP_MBCONV_FUSED", "0") != "1":
                    # depthwise conv that leaves the squeeze-excite pool's partial sums: the gate kernel does not read h again
                    e = ops.conv2d(y, blk["exp"], act=act) if blk["exp"] is not None else y
                    h, a = ops.dwconv2d_se(e, blk["dw"][0], blk["dw"][1], k, s, pad, act, blk["se"][0], blk["se"][1], act, "sigmoid")
                else:
                    if blk["exp"] is not None:                       # expand 1x1 + depthwise: one launch where the C ABI takes the shape
                        h = ops.mbconv_expand_dw(y, blk["exp"], blk["dw"][0], blk["dw"][1], k, s, pad, act=act)
                    else:
                        h = ops.dwconv2d(y, blk["dw"][0], blk["dw"][1], k, s, pad, act=act)
                    a = ops.se_gate(h, blk["se"][0], blk["se"][1], act, "sigmoid") if blk["se"] is not None else None
                y = ops.conv2d(h, blk["out"], residual=inp if blk["shortcut"] else None, gate=a)   # h * se folded in
            last_of_stage = i + 1 == len(self.blocks) or self.blocks[i + 1]["stage"] != blk["stage"]
            if collect is not None and last_of_stage:
                collect.append(y)
        return ops.conv2d(y, self.post, act=act)


# ------------------------------------------------------------------------------------------------
# ECA_NFNetL0
# ------------------------------------------------------------------------------------------------
NFNET_L0 = dict(num_blocks=(1, 2, 6, 3), out_channels=(256, 512, 1536, 1536), strides=(1, 2, 2, 2), stem_width=128,
                alpha=0.2, channel_ratio=0.25, group_size=64, num_features_factor=1.5)
NFNET_L2 = dict(NFNET_L0, num_blocks=(3, 6, 18, 9), num_features_factor=2)     # nfnets.py:329-332 (NormFreeNet default factor :210)
SWISH_GAMMA = 1.7881293296813965  # nfnets.py:34


def eca_kernel_size(filters, gamma=2.0, beta=1.0):
    tt = int((math.log(float(filters)) / math.log(2.0) + beta) / gamma)
    return max(tt if tt % 2 else tt + 1, 3)


def nfnet_synth_params(seed: int, classes: int = 1, cfg=NFNET_L0) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)

    def sconv(name, k, cin_g, cout):
        g.conv(f"{name}conv", k, k, cin_g, cout, bias=True)
        g.raw(f"{name}conv/gain", g._u((cout,), 0.8, 1.2))

    sw = cfg["stem_width"]
    cin = 3
    for i, wd in enumerate((sw // 8, sw // 4, sw // 2, sw), 1):
        sconv(f"stem_{i}_", 3, cin, wd)
        cin = wd
    for si, (nb, oc, st) in enumerate(zip(cfg["num_blocks"], cfg["out_channels"], cfg["strides"])):
        hid = int(oc * cfg["channel_ratio"])
        for bi in range(nb):
            n = f"stack{si + 1}_block{bi + 1}_"
            s = st if bi == 0 else 1
            if s > 1 or cin != oc:
                sconv(f"{n}shortcut_", 1, cin, oc)
            sconv(f"{n}deep_1_", 1, cin, hid)
            sconv(f"{n}deep_2_", 3, cfg["group_size"], hid)
            sconv(f"{n}deep_3_", 3, cfg["group_size"], hid)
            sconv(f"{n}deep_4_", 1, hid, oc)
            g.raw(f"{n}eca_conv1d/kernel", g._n((eca_kernel_size(oc), 1, 1), 0.5))
            cin = oc
    post = make_divisible(cfg["num_features_factor"] * cfg["out_channels"][-1], 8)
    sconv("post_", 1, cin, post)
    g.dense("predictions", post, classes)
    return g.p


def _std_fold(p, name, scale=1.0, gamma=SWISH_GAMMA, eps=1e-5):
    """ScaledStandardizedConv2D weight (nfnets.py:64-70) times an extra folded scalar; returns (kernel, bias)"""
    w = p[f"{name}conv/kernel"]
    mean = w.mean(dim=(0, 1, 2), keepdim=True)
    var = w.var(dim=(0, 1, 2), unbiased=False, keepdim=True)
    fan_in = w.shape[0] * w.shape[1] * w.shape[2]
    sc = torch.rsqrt(torch.clamp(var * fan_in, min=eps)) * (p[f"{name}conv/gain"] * gamma)
    return (w - mean) * sc * scale, p[f"{name}conv/bias"]


class NormFreeNet(_Base):
    def __init__(self, params, cfg=NFNET_L0, classes=1, first_strides=2, device="cuda", classifier_activation="default"):
        p, dev = params, device
        self.head_act = classifier_activation
        self.cfg, self.classes, self.first_strides = cfg, classes, first_strides
        mk = lambda name, groups=1, in_scale=1.0, out_scale=1.0, pad_cin=None: ops.make_conv_weight(  # noqa: E731
            _std_fold(p, name, in_scale * out_scale)[0], _std_fold(p, name)[1] * out_scale, groups=groups, device=dev,
            pad_cin_to=pad_cin)
        self.stem = [mk(f"stem_{i}_", pad_cin=8 if i == 1 else None) for i in range(1, 5)]
        alpha = cfg["alpha"]
        beta_list = [(1 + alpha ** 2 * ii) ** -0.5 for ii in range(max(cfg["num_blocks"]) + 1)]
        pre_beta = 1.0
        self.blocks = []
        cin = cfg["stem_width"]
        for si, (nb, oc, st) in enumerate(zip(cfg["num_blocks"], cfg["out_channels"], cfg["strides"])):
            betas = beta_list[:nb + 1]
            betas[0] = pre_beta
            hid = int(oc * cfg["channel_ratio"])
            groups = hid // cfg["group_size"]
            for bi in range(nb):
                n = f"stack{si + 1}_block{bi + 1}_"
                s = st if bi == 0 else 1
                beta = betas[bi]
                blk = {"stride": s, "sc": None, "last": bi == nb - 1}
                if s > 1 or cin != oc:
                    blk["sc"] = mk(f"{n}shortcut_", in_scale=beta)           # conv(beta * a) = beta * conv(a)
                blk["d1"] = mk(f"{n}deep_1_", in_scale=beta)
                blk["d2"] = mk(f"{n}deep_2_", groups=groups)
                blk["d3"] = mk(f"{n}deep_3_", groups=groups)
                blk["d4"] = mk(f"{n}deep_4_", out_scale=2.0 * alpha)         # attn_gain * alpha (nfnets.py:136,160-167)
                # ECA Conv1D(k) over channels as a banded matrix: out[c] = sum_j w[j] * g[c + j - k//2]
                k = eca_kernel_size(oc)
                wk = p[f"{n}eca_conv1d/kernel"].reshape(k)
                band = torch.zeros(oc, oc)
                for j in range(k):
                    off = j - k // 2
                    idx = torch.arange(max(0, -off), min(oc, oc - off))
                    band[idx + off, idx] = wk[j]
                # the squeeze sees deep_4's output BEFORE attn_gain * alpha, which is folded into d4: undo it here
                blk["eca"] = ops.make_dense_weight(band / (2.0 * alpha), None, dev)
                self.blocks.append(blk)
                cin = oc
            pre_beta = betas[-1]
        self.post = mk("post_")
        self.head_w, self.head_b = _head(p, dev)

    def features(self, x, collect=None):
        assert x.shape[-1] == 8
        act = "silu"
        fs = self.first_strides
        y = ops.conv2d(x, self.stem[0], stride=fs, pad=PAD1, act=act)          # stem (:182-191)
        y = ops.conv2d(y, self.stem[1], pad=PAD1, act=act)
        y = ops.conv2d(y, self.stem[2], pad=PAD1, act=act)
        y = ops.conv2d(y, self.stem[3], stride=2, pad=PAD1)
        pre = ops.scale_add_act(y, None, None, act)                            # act(x); beta folded into the convs
        for bi, blk in enumerate(self.blocks):                                 # block (:116-168)
            s = blk["stride"]
            if blk["sc"] is not None:
                H, W = pre.shape[1], pre.shape[2]
                sc = ops.pool2d(pre, 2, 2, (0, H % 2, 0, W % 2), ops.POOL_AVG_VALID) if s > 1 else pre
                sc = ops.conv2d(sc, blk["sc"])
            else:
                sc = y
            d = ops.conv2d(pre, blk["d1"], act=act)
            d = ops.conv2d(d, blk["d2"], stride=s, pad=PAD1, act=act)
            d = ops.conv2d(d, blk["d3"], pad=PAD1, act=act)
            d = ops.conv2d(d, blk["d4"])
            a = ops.dense_split(ops.global_avgpool(d, split=True), blk["eca"], act="sigmoid")   # pooled vector and gate: hi/lo planes
            if bi + 1 < len(self.blocks):      # the next block's act(x) is a second output of this launch
                y, pre = ops.scale_add_act(d, a, sc, None, act2=act)
            else:
                y = ops.scale_add_act(d, a, sc, None)
            if collect is not None and blk["last"]:
                collect.append(y)
        return ops.conv2d(y, self.post, act=act)
