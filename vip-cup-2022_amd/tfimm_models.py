"""tfimm ViT and ConvNeXt on the HIP operator set — host-side mirrors of
models/tfimm/architectures/vit.py (``ViTConfig`` :35-118, ``ViTMultiHeadAttention`` :121-167, ``ViTBlock``
:170-227, ``ViT`` :290-467) and convnext.py (``ConvNeXtConfig`` :66-135, ``ConvNeXtBlock`` :147-229,
``ConvNeXtStage`` :232-296, ``ConvNeXt`` :299-438), with the shared layers of
models/tfimm/layers/transformers.py (``PatchEmbeddings`` :79-173, ``MLP`` :176-214).
Only the constructor arguments that change the inference graph are kept.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import ops
from .pipeline import keras_predict
from .synth import ParamGen

LN_EPS = 1e-6  # norm_layer "layer_norm_eps_1e-6" (vit.py:55, convnext.py:124)


class _LN:
    def __init__(self, p, name, dev):
        self.g = p[f"{name}/gamma"].to(dev, torch.float32).contiguous()
        self.b = p[f"{name}/beta"].to(dev, torch.float32).contiguous()

    def __call__(self, x):
        return ops.layernorm(x, self.g, self.b, LN_EPS)


# ------------------------------------------------------------------------------------------------
# ViT
# ------------------------------------------------------------------------------------------------
@dataclass
class ViTConfig:
    """vit.py:35-66 (inference-relevant fields)"""
    name: str
    input_size: Tuple[int, int] = (224, 224)
    patch_size: int = 16
    embed_dim: int = 768
    nb_blocks: int = 12
    nb_heads: int = 12
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    nb_classes: int = 1
    interpolate_input: bool = False     # vit.py:58: adapt the position embeddings to the input's patch grid at run time

    @property
    def grid_size(self):
        return (self.input_size[0] // self.patch_size, self.input_size[1] // self.patch_size)

    @property
    def nb_patches(self):
        return self.grid_size[0] * self.grid_size[1]


def _tf_bicubic_taps(out_size: int, in_size: int):
    """indices [out, 4] and fp32 weights [out, 4] of tf.image.resize(method="bicubic", antialias=False) along one axis - the legacy
    kernel with half-pixel centres: Keys a = -0.5, the fractional offset quantised to the 1024-entry coefficient table
    (vip_bicubic_table_f32, the table the image pipeline uses on the GPU), taps outside the image dropped and the rest renormalised"""
    import ctypes as C
    import numpy as np
    from . import _abi
    table = np.zeros((1025 * 2,), dtype=np.float32)
    _abi.check(_abi.lib().vip_bicubic_table_f32(table.ctypes.data_as(C.c_void_p)), "vip_bicubic_table_f32")
    f32 = np.float32
    scale = f32(in_size) / f32(out_size)
    loc = (np.arange(out_size, dtype=f32) + f32(0.5)) * scale - f32(0.5)
    fl = np.floor(loc)
    off = np.rint((loc - fl) * f32(1024)).astype(np.int64)            # lrintf: round half to even
    w = np.stack([table[off * 2 + 1], table[off * 2], table[(1024 - off) * 2], table[(1024 - off) * 2 + 1]], 1)
    want = fl.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :]
    idx = np.clip(want, 0, in_size - 1)
    w = np.where(idx == want, w, f32(0)).astype(f32)
    ssum = ((w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3]
    ok = np.abs(ssum) >= f32(1000.0) * f32(1.17549435e-38)
    w = np.where(ok[:, None], w * (f32(1) / np.where(ok, ssum, f32(1)))[:, None], w).astype(f32)
    return idx, w


def interpolate_pos_embeddings(pos_embed: torch.Tensor, src_grid, tgt_grid, nb_tokens: int = 1) -> torch.Tensor:
    """tfimm/layers/transformers.py:13-47: position embeddings ``[N, D]`` of a ``src_grid`` patch grid (the first ``nb_tokens`` rows
    belong to the class token(s) and are kept) resampled to ``tgt_grid`` with tf.image.resize(bicubic).  Host arithmetic in fp32 at
    load time / first use of an input size (a [gh, gw, D] array of a few hundred KB), the same operation order as the GPU image
    resize: along x on the four source rows first, then along y, each as ((v0 w0 + v1 w1) + v2 w2) + v3 w3."""
    import numpy as np
    if tuple(src_grid) == tuple(tgt_grid):
        return pos_embed
    pe = pos_embed.detach().cpu().to(torch.float32).numpy()
    D = pe.shape[-1]
    src = pe[nb_tokens:].reshape(src_grid[0], src_grid[1], D)
    iy, wy = _tf_bicubic_taps(tgt_grid[0], src_grid[0])
    ix, wx = _tf_bicubic_taps(tgt_grid[1], src_grid[1])
    rows = None
    for t in range(4):
        term = src[:, ix[:, t], :] * wx[None, :, t, None]
        rows = term if rows is None else rows + term
    out = None
    for t in range(4):
        term = rows[iy[:, t], :, :] * wy[:, t, None, None]
        out = term if out is None else out + term
    out = np.concatenate([pe[:nb_tokens], out.reshape(tgt_grid[0] * tgt_grid[1], D).astype(np.float32)], 0)
    return torch.from_numpy(out)


def variant(cfg, overrides: Dict):
    """config dataclass with the fields a checkpoint's model_config overrides (tfimm serialises the dataclass itself,
    models/serialization.py:75-76); unknown keys are an error, fields this graph does not read are ignored by the caller"""
    import dataclasses
    if not overrides:
        return cfg
    known = {f.name for f in dataclasses.fields(cfg)}
    bad = set(overrides) - known
    if bad:
        raise ValueError(f"{cfg.name}: unknown config fields {sorted(bad)}")
    return dataclasses.replace(cfg, **overrides)


VIT_CONFIGS = {
    "vit_tiny_patch16_224": ViTConfig("vit_tiny_patch16_224", embed_dim=192, nb_heads=3),    # vit.py:470-481
    "vit_small_patch16_224": ViTConfig("vit_small_patch16_224", embed_dim=384, nb_heads=6),  # vit.py:530-541
    "vit_base_patch16_224": ViTConfig("vit_base_patch16_224", embed_dim=768, nb_heads=12),   # vit.py:598-613
}


def vit_synth_params(cfg: ViTConfig, seed: int) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    D, ps = cfg.embed_dim, cfg.patch_size
    g.conv("patch_embed/proj", ps, ps, 3, D, bias=True, gain=1.0)
    g.raw("cls_token", g._n((1, 1, D), 0.5))
    g.raw("pos_embed", g._n((1, cfg.nb_patches + 1, D), 0.5))
    hid = int(D * cfg.mlp_ratio)
    for j in range(cfg.nb_blocks):
        b = f"blocks/{j}"
        g.ln(f"{b}/norm1", D)
        g.dense(f"{b}/attn/qkv", D, 3 * D, bias=cfg.qkv_bias)
        g.dense(f"{b}/attn/proj", D, D, gain=0.25)
        g.ln(f"{b}/norm2", D)
        g.dense(f"{b}/mlp/fc1", D, hid, gain=2.0)
        g.dense(f"{b}/mlp/fc2", hid, D, gain=0.25)
    g.ln("norm", D)
    g.dense("head", D, cfg.nb_classes)
    return g.p


@keras_predict
class ViT:
    def __init__(self, params: Dict[str, torch.Tensor], cfg: ViTConfig, device="cuda"):
        p, dev = params, device
        self.cfg = cfg
        D = cfg.embed_dim
        assert D // cfg.nb_heads == 64, "vip_mhsa_fwd_f16 implements head_dim 64"
        self.patch = ops.make_conv_weight(p["patch_embed/proj/kernel"], p["patch_embed/proj/bias"], device=dev, pad_cin_to=8)
        self.cls = ops.to_act(p["cls_token"].reshape(D), dev)      # stored like an activation (packed pairs when strict, fp32 in f32 mode)
        self.pos = ops.to_act(p["pos_embed"].reshape(-1, D), dev)
        self._pos32 = p["pos_embed"].reshape(-1, D).to(torch.float32)                  # host copy: source of interpolated grids
        self._pos_by_grid = {tuple(cfg.grid_size): self.pos}
        self._act_dtype, self._dev = ops.act_dtype(), dev
        self.blocks = []
        for j in range(cfg.nb_blocks):
            b = f"blocks/{j}"
            qkv_bias = p.get(f"{b}/attn/qkv/bias")
            self.blocks.append(dict(
                n1=_LN(p, f"{b}/norm1", dev), n2=_LN(p, f"{b}/norm2", dev),
                qkv=ops.make_dense_weight(p[f"{b}/attn/qkv/kernel"], qkv_bias, dev),
                proj=ops.make_dense_weight(p[f"{b}/attn/proj/kernel"], p[f"{b}/attn/proj/bias"], dev),
                fc1=ops.make_dense_weight(p[f"{b}/mlp/fc1/kernel"], p[f"{b}/mlp/fc1/bias"], dev),
                fc2=ops.make_dense_weight(p[f"{b}/mlp/fc2/kernel"], p[f"{b}/mlp/fc2/bias"], dev)))
        self.norm = _LN(p, "norm", dev)
        self.head_w = p["head/kernel"].t().contiguous().to(dev, torch.float32)
        self.head_b = p["head/bias"].to(dev, torch.float32)

    def features(self, x, collect=None):
        """ViT.forward_features (vit.py:414-451) up to the final norm; returns [B, N, D] tokens."""
        cfg = self.cfg
        assert x.shape[-1] == 8
        if not cfg.interpolate_input and tuple(x.shape[1:3]) != tuple(cfg.input_size):
            raise ValueError(f"{cfg.name}: interpolate_input=False (vit.py:58): the input must be {cfg.input_size}, got {tuple(x.shape[1:3])}")
        B = x.shape[0]
        ps = cfg.patch_size
        pe = ops.conv2d(x, self.patch, stride=ps)                       # PatchEmbeddings (transformers.py:131-139), VALID: floor(H / ps)
        grid = (pe.shape[1], pe.shape[2])
        if grid not in self._pos_by_grid:                                # vit.py:425-433: embeddings resampled to the input's patch grid
            pos = interpolate_pos_embeddings(self._pos32, cfg.grid_size, grid, nb_tokens=1)
            self._pos_by_grid[grid] = ops.to_act(pos, self._dev, self._act_dtype)
        t = ops.vit_tokens(pe.reshape(B, grid[0] * grid[1], cfg.embed_dim), self.cls, self._pos_by_grid[grid])
        scale = (cfg.embed_dim // cfg.nb_heads) ** -0.5
        for blk in self.blocks:                                          # ViTBlock.call (vit.py:214-227)
            qkv = ops.dense(blk["n1"](t), blk["qkv"])
            att = ops.mhsa(qkv, cfg.nb_heads, scale)
            t = ops.dense(att, blk["proj"], residual=t)
            t = ops.mlp(t, blk["fc1"], blk["fc2"], act="gelu", residual=t, ln=(blk["n2"].g, blk["n2"].b, LN_EPS))
            if collect is not None:
                collect.append(t)
        return self.norm(t)

    def logits(self, x):
        t = self.features(x)                                             # head(norm(x)[:, 0]) (vit.py:441-461)
        return ops.cls_dense_f32(t, self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))


# ------------------------------------------------------------------------------------------------
# ConvNeXt
# ------------------------------------------------------------------------------------------------
@dataclass
class ConvNeXtConfig:
    """convnext.py:66-135 (inference-relevant fields)"""
    name: str
    patch_size: int = 4
    first_down: int = 1            # stem stride = first_down*2 (convnext.py:320-327): 2, NOT the stock 4
    embed_dim: Tuple = (96, 192, 384, 768)
    nb_blocks: Tuple = (3, 3, 9, 3)
    mlp_ratio: float = 4.0
    nb_classes: int = 1


CONVNEXT_CONFIGS = {
    "convnext_tiny_in22k": ConvNeXtConfig("convnext_tiny_in22k"),  # convnext.py:611-620
    # members of the earlier, larger ensembles (main.py:43-56 NAME2BS): same graph, wider / deeper
    "convnext_small_in22k": ConvNeXtConfig("convnext_small_in22k", nb_blocks=(3, 3, 27, 3)),                          # :623-632
    "convnext_base_in22k": ConvNeXtConfig("convnext_base_in22k", embed_dim=(128, 256, 512, 1024), nb_blocks=(3, 3, 27, 3)),    # :635-644
    "convnext_large_in22ft1k": ConvNeXtConfig("convnext_large_in22ft1k", embed_dim=(192, 384, 768, 1536), nb_blocks=(3, 3, 27, 3)),  # :518-527
    # the 384-pixel fine-tunes: the same graphs (the ensemble runs them at 200 x 200 anyway)            :575-596
    "convnext_base_384_in22ft1k": ConvNeXtConfig("convnext_base_384_in22ft1k", embed_dim=(128, 256, 512, 1024), nb_blocks=(3, 3, 27, 3)),
    "convnext_large_384_in22ft1k": ConvNeXtConfig("convnext_large_384_in22ft1k", embed_dim=(192, 384, 768, 1536), nb_blocks=(3, 3, 27, 3)),
}


def convnext_synth_params(cfg: ConvNeXtConfig, seed: int) -> Dict[str, torch.Tensor]:
    g = ParamGen(seed)
    g.conv("stem/0", cfg.patch_size, cfg.patch_size, 3, cfg.embed_dim[0], bias=True, gain=1.0)
    g.ln("stem/1", cfg.embed_dim[0])
    cprev = cfg.embed_dim[0]
    for j, (c, nb) in enumerate(zip(cfg.embed_dim, cfg.nb_blocks)):
        if j > 0:
            g.ln(f"stages/{j}/downsample/0", cprev)
            g.conv(f"stages/{j}/downsample/1", 2, 2, cprev, c, bias=True, gain=1.0)
        hid = int(cfg.mlp_ratio * c)
        for i in range(nb):
            b = f"stages/{j}/blocks/{i}"
            g.dwconv(f"{b}/conv_dw", 7, c, bias=True, gain=1.0)
            g.ln(f"{b}/norm", c)
            g.dense(f"{b}/mlp/fc1", c, hid, gain=2.0)
            g.dense(f"{b}/mlp/fc2", hid, c)
            g.raw(f"{b}/gamma", g._u((c,), 0.1, 0.4))   # layer scale (trained values; the 1e-6 init would mute the block)
        cprev = c
    g.ln("head/norm", cprev)
    g.dense("head/fc", cprev, cfg.nb_classes)
    return g.p


@keras_predict
class ConvNeXt:
    def __init__(self, params: Dict[str, torch.Tensor], cfg: ConvNeXtConfig, device="cuda"):
        p, dev = params, device
        self.cfg = cfg
        self.stem = ops.make_conv_weight(p["stem/0/kernel"], p["stem/0/bias"], device=dev, pad_cin_to=8)
        self.stem_norm = _LN(p, "stem/1", dev)
        self.stages = []
        for j, nb in enumerate(cfg.nb_blocks):
            st = {"down": None, "blocks": []}
            if j > 0:
                st["down"] = (_LN(p, f"stages/{j}/downsample/0", dev),
                              ops.make_conv_weight(p[f"stages/{j}/downsample/1/kernel"],
                                                   p[f"stages/{j}/downsample/1/bias"], device=dev))
            for i in range(nb):
                b = f"stages/{j}/blocks/{i}"
                gamma = p[f"{b}/gamma"]
                # x*gamma folded into fc2: (W x + b) * gamma = (W*gamma) x + b*gamma  (convnext.py:226)
                st["blocks"].append(dict(
                    dw=ops.make_dw_weight(p[f"{b}/conv_dw/depthwise_kernel"], None, dev),
                    dwb=p[f"{b}/conv_dw/bias"].to(dev, torch.float32).contiguous(),
                    norm=_LN(p, f"{b}/norm", dev),
                    fc1=ops.make_dense_weight(p[f"{b}/mlp/fc1/kernel"], p[f"{b}/mlp/fc1/bias"], dev),
                    fc2=ops.make_dense_weight(p[f"{b}/mlp/fc2/kernel"] * gamma, p[f"{b}/mlp/fc2/bias"] * gamma, dev)))
            self.stages.append(st)
        self.head_norm = _LN(p, "head/norm", dev)
        self.head_w = p["head/fc/kernel"].t().contiguous().to(dev, torch.float32)
        self.head_b = p["head/fc/bias"].to(dev, torch.float32)

    def features(self, x, collect=None):
        """ConvNeXt.forward_features (convnext.py:376-406)"""
        assert x.shape[-1] == 8
        cfg = self.cfg
        y = self.stem_norm(ops.conv2d(x, self.stem, stride=cfg.first_down * 2))   # VALID, stride 2
        for st in self.stages:
            if st["down"] is not None:
                ln, cw = st["down"]
                y = ops.conv2d(ln(y), cw, stride=2)                               # 2x2/2 VALID (convnext.py:260-267)
            for blk in st["blocks"]:                                              # ConvNeXtBlock.call (:220-229)
                h = ops.dwconv2d(y, blk["dw"], blk["dwb"], 7, 1, (3, 3, 3, 3))
                n = blk["norm"]                                                   # LN fused into the MLP launch
                y = ops.mlp(h, blk["fc1"], blk["fc2"], act="gelu", residual=y, ln=(n.g, n.b, LN_EPS))
            if collect is not None:
                collect.append(y)
        return y

    def logits(self, x):
        f = self.features(x)                                                      # pool -> norm -> fc (:432-436)
        n = self.head_norm
        return ops.gap_ln_dense_f32(f, n.g, n.b, LN_EPS, self.head_w, self.head_b)

    def predict(self, x):
        z = self.logits(x)
        return ops.head_prob(z, getattr(self, "head_act", "default"))
