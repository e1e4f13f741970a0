"""Ensemble member registry — the MI355X counterpart of the reference's ckpts/ckpts.json manifest
(``[name, [H, W], idx]``, main.py:171-198) plus the model constructors main.py:28-37 imports.

No trained checkpoints ship with the reference (README.md:13), so each member is instantiated from its
seeded synthetic checkpoint (synth.py); ``params`` dictionaries use the reference's Keras variable names.
"""
import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import gcvit, hornet, kecam_models as km, resnet_rs, tfimm_models as tm


@dataclass
class MemberSpec:
    name: str            # registry key
    ckpt_name: str       # directory name in the reference manifest (ckpts/ckpts.json)
    input_hw: int        # square input resolution from the manifest
    seed: int            # synthetic-checkpoint seed (1000 + manifest index)
    synth: Callable      # seed -> params dict
    ctor: Callable       # params -> model with .logits(x) / .predict(x)
    oracle: str          # module under oracle/ that restates the graph (used by tests / bench cpu leg only)
    gmac_per_image: float  # algorithmic GMAC / image
    head: str = "predictions"  # Keras name of the classifier Dense
    family: str = ""           # constructor-argument dialect for checkpoint variants (variant_kwargs): filled in below from `oracle`


MEMBERS: Dict[str, MemberSpec] = {
    "resnet_rs50": MemberSpec("resnet_rs50", "ResNetRS50-200x200", 200, 1006,
                              lambda seed: resnet_rs.synth_params(50, seed),
                              lambda p, **kw: resnet_rs.ResNetRS50(p, **kw), "resnet_rs_ref", 3.790),
    # members of the earlier, larger ensembles that are plain re-configurations of graphs built here (main.py:43-56)
    "resnet_rs101": MemberSpec("resnet_rs101", "ResNetRS101-200x200", 200, 1016, lambda seed: resnet_rs.synth_params(101, seed),
                               lambda p, **kw: resnet_rs.ResNetRS(p, depth=101, **kw), "resnet_rs_ref", 7.30),
    "resnet_rs200": MemberSpec("resnet_rs200", "ResNetRS200-200x200", 200, 1026, lambda seed: resnet_rs.synth_params(200, seed),
                               lambda p, **kw: resnet_rs.ResNetRS(p, depth=200, **kw), "resnet_rs_ref", 14.7),
    "convnext_small_in22k": MemberSpec("convnext_small_in22k", "convnext_small_in22k-200x200", 200, 1010,
                                       lambda seed: tm.convnext_synth_params(tm.CONVNEXT_CONFIGS["convnext_small_in22k"], seed),
                                       lambda p, **kw: tm.ConvNeXt(p, tm.variant(tm.CONVNEXT_CONFIGS["convnext_small_in22k"], kw)), "tfimm_ref", 25.9, "head/fc"),
    "convnext_base_in22k": MemberSpec("convnext_base_in22k", "convnext_base_in22k-200x200", 200, 1020,
                                      lambda seed: tm.convnext_synth_params(tm.CONVNEXT_CONFIGS["convnext_base_in22k"], seed),
                                      lambda p, **kw: tm.ConvNeXt(p, tm.variant(tm.CONVNEXT_CONFIGS["convnext_base_in22k"], kw)), "tfimm_ref", 45.8, "head/fc"),
    "convnext_large_in22ft1k": MemberSpec("convnext_large_in22ft1k", "convnext_large_in22ft1k-200x200", 200, 1030,
                                          lambda seed: tm.convnext_synth_params(tm.CONVNEXT_CONFIGS["convnext_large_in22ft1k"], seed),
                                          lambda p, **kw: tm.ConvNeXt(p, tm.variant(tm.CONVNEXT_CONFIGS["convnext_large_in22ft1k"], kw)), "tfimm_ref", 102.7, "head/fc"),
    "resnest200": MemberSpec("resnest200", "ResNest200-200x200", 200, 1021, lambda seed: km.resnest_synth_params(seed, cfg=km.RESNEST200),
                             lambda p, **kw: km.ResNest(p, cfg=km.RESNEST200, **kw), "kecam_ref", 29.0),
    "resnet200d": MemberSpec("resnet200d", "ResNet200D-200x200", 200, 1027, lambda seed: km.resnest_synth_params(seed, cfg=km.RESNET200D),
                             lambda p, **kw: km.ResNest(p, cfg=km.RESNET200D, **kw), "kecam_ref", 12.0),
    "eca_nfnet_l2": MemberSpec("eca_nfnet_l2", "ECA_NFNetL2-200x200", 200, 1025, lambda seed: km.nfnet_synth_params(seed, cfg=km.NFNET_L2),
                               lambda p, **kw: km.NormFreeNet(p, cfg=km.NFNET_L2, **kw), "kecam_ref", 10.6),
    "efficientnet_v2m": MemberSpec("efficientnet_v2m", "EfficientNetV2M-200x200", 200, 1023,
                                   lambda seed: km.effnet_synth_params("EfficientNetV2M", seed),
                                   lambda p, **kw: km.EfficientNet(p, "EfficientNetV2M", **kw), "kecam_ref", 4.3),
    "efficientnet_v2l": MemberSpec("efficientnet_v2l", "EfficientNetV2L-200x200", 200, 1024,
                                   lambda seed: km.effnet_synth_params("EfficientNetV2L", seed),
                                   lambda p, **kw: km.EfficientNet(p, "EfficientNetV2L", **kw), "kecam_ref", 9.8),
    "convnext_base_384_in22ft1k": MemberSpec("convnext_base_384_in22ft1k", "convnext_base_384_in22ft1k-200x200", 200, 1031,
        lambda seed: tm.convnext_synth_params(tm.CONVNEXT_CONFIGS["convnext_base_384_in22ft1k"], seed),
        lambda p, **kw: tm.ConvNeXt(p, tm.variant(tm.CONVNEXT_CONFIGS["convnext_base_384_in22ft1k"], kw)), "tfimm_ref", 45.8, "head/fc"),
    "convnext_large_384_in22ft1k": MemberSpec("convnext_large_384_in22ft1k", "convnext_large_384_in22ft1k-200x200", 200, 1033,
        lambda seed: tm.convnext_synth_params(tm.CONVNEXT_CONFIGS["convnext_large_384_in22ft1k"], seed),
        lambda p, **kw: tm.ConvNeXt(p, tm.variant(tm.CONVNEXT_CONFIGS["convnext_large_384_in22ft1k"], kw)), "tfimm_ref", 102.7, "head/fc"),
    "hornet_base": MemberSpec("hornet_base", "HorNetBase-200x200", 200, 1028,
                              lambda seed: hornet.synth_params(hornet.CONFIGS["hornet_base"], seed),
                              lambda p, **kw: hornet.HorNet(p, **hornet.CONFIGS["hornet_base"], **kw), "hornet_ref", 11.6),
    "gcvit_base": MemberSpec("gcvit_base", "GCViTBase-224x224", 224, 1022,
                             lambda seed: gcvit.synth_params(gcvit.NAME2CONFIG["gcvit_base"], seed),
                             lambda p, **kw: gcvit.GCViT(p, **gcvit.NAME2CONFIG["gcvit_base"], **kw), "gcvit_ref", 14.3, "head"),
    "gcvit_tiny": MemberSpec("gcvit_tiny", "GCViTTiny-224x224", 224, 1002,
                             lambda seed: gcvit.synth_params(gcvit.NAME2CONFIG["gcvit_tiny"], seed),
                             lambda p, **kw: gcvit.GCViTTiny(p, **kw), "gcvit_ref", 4.760, "head"),
    "convnext_tiny_in22k": MemberSpec("convnext_tiny_in22k", "convnext_tiny_in22k-200x200", 200, 1000,
                                      lambda seed: tm.convnext_synth_params(tm.CONVNEXT_CONFIGS["convnext_tiny_in22k"], seed),
                                      lambda p, **kw: tm.ConvNeXt(p, tm.variant(tm.CONVNEXT_CONFIGS["convnext_tiny_in22k"], kw)),
                                      "tfimm_ref", 13.33, "head/fc"),
    "resnest50": MemberSpec("resnest50", "ResNest50-200x200", 200, 1001, lambda seed: km.resnest_synth_params(seed),
                            lambda p, **kw: km.ResNest(p, **kw), "kecam_ref", 4.518),
    "efficientnet_v2t": MemberSpec("efficientnet_v2t", "EfficientNetV2T-200x200", 200, 1003,
                                   lambda seed: km.effnet_synth_params("EfficientNetV2T", seed),
                                   lambda p, **kw: km.EfficientNet(p, "EfficientNetV2T", **kw), "kecam_ref", 1.627),
    "efficientnet_v1b4": MemberSpec("efficientnet_v1b4", "EfficientNetV1B4-224x224", 224, 1004,
                                    lambda seed: km.effnet_synth_params("EfficientNetV1B4", seed),
                                    lambda p, **kw: km.EfficientNet(p, "EfficientNetV1B4", **kw), "kecam_ref", 1.502),
    "eca_nfnet_l0": MemberSpec("eca_nfnet_l0", "ECA_NFNetL0-200x200", 200, 1005, lambda seed: km.nfnet_synth_params(seed),
                               lambda p, **kw: km.NormFreeNet(p, **kw), "kecam_ref", 3.617),
    "vit_tiny_patch16_224": MemberSpec("vit_tiny_patch16_224", "vit_tiny_patch16_224-224x224", 224, 1008,
                                       lambda seed: tm.vit_synth_params(tm.VIT_CONFIGS["vit_tiny_patch16_224"], seed),
                                       lambda p, **kw: tm.ViT(p, tm.variant(tm.VIT_CONFIGS["vit_tiny_patch16_224"], kw)), "tfimm_ref", 1.253, "head"),
    "vit_small_patch16_224": MemberSpec("vit_small_patch16_224", "vit_small_patch16_224-224x224", 224, 1007,
                                        lambda seed: tm.vit_synth_params(tm.VIT_CONFIGS["vit_small_patch16_224"], seed),
                                        lambda p, **kw: tm.ViT(p, tm.variant(tm.VIT_CONFIGS["vit_small_patch16_224"], kw)), "tfimm_ref", 4.598, "head"),
    "vit_base_patch16_224": MemberSpec("vit_base_patch16_224", "vit_base_patch16_224-224x224", 224, 1009,
                                       lambda seed: tm.vit_synth_params(tm.VIT_CONFIGS["vit_base_patch16_224"], seed),
                                       lambda p, **kw: tm.ViT(p, tm.variant(tm.VIT_CONFIGS["vit_base_patch16_224"], kw)), "tfimm_ref", 17.56, "head"),
}

# order of ckpts/ckpts.json:2-8 (members are appended here as their graphs land)
ENSEMBLE: List[str] = ["convnext_tiny_in22k", "resnest50", "gcvit_tiny", "efficientnet_v2t", "efficientnet_v1b4",
                       "eca_nfnet_l0", "resnet_rs50"]
# BASELINE.json config 5 asks for 8 members: the shipped 7 + a tfimm ViT
ENSEMBLE8: List[str] = ENSEMBLE + ["vit_small_patch16_224"]
# BASELINE.json config 4: ResNet-RS + GCViT + 2x tfimm ViT
ENSEMBLE4: List[str] = ["resnet_rs50", "gcvit_tiny", "vit_tiny_patch16_224", "vit_small_patch16_224"]


_HEADS = None


def build_params(name: str, calibrated: bool = True):
    """Synthetic checkpoint of a member; with ``calibrated`` the classifier head is replaced by the committed
    calibrated head (synth_heads.npz, generated by tests/gen_synth_heads.py) when one exists."""
    global _HEADS
    spec = MEMBERS[name]
    params = spec.synth(spec.seed)
    if calibrated:
        if _HEADS is None:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_heads.npz")
            _HEADS = dict(np.load(path)) if os.path.exists(path) else {}
        if f"{name}/kernel" in _HEADS:
            params[f"{spec.head}/kernel"] = torch.from_numpy(_HEADS[f"{name}/kernel"]).clone()
            params[f"{spec.head}/bias"] = torch.from_numpy(_HEADS[f"{name}/bias"]).clone()
    return params


def construct(spec: MemberSpec, params, bias_calibration: bool = True, precision: str = None, calibration_batch=None,
              variant: Optional[dict] = None):
    """``spec.ctor(params)`` in a precision mode (``ops.PRECISION`` when None; env ``VIP_PRECISION``):

    * ``"fast"``: fp16 weights, followed on a GPU by one calibration pass (ops.calibration: the image-independent part of the fp16
      weight-rounding error is folded into the fp32 biases) over ``calibration_batch`` (a ``pipeline.DecodedBatch``; default: the
      seeded synthetic batch ``pipeline.calibration_batch()`` - pass a few REAL images when the checkpoints are real, the correction
      needs typical per-channel input means).  ``bias_calibration=False`` (or env ``VIP_BIAS_CALIBRATION=0``) gives the plain fp16 model.
    * ``"strict"``: every weight as an fp16 (hi, lo) pair (22 bits), nothing to calibrate; the model takes packed-strict inputs
      (``ops.PACKED``) and every member's logit is within the 1e-3 of BASELINE.json (measured <= 1e-4).
    * ``"f32"``: fp32 weights exactly as in the checkpoint, fp32 activations (round 3's strict mode; the fallback when an activation
      leaves the fp16 range of the packed storage).

    ``variant``: constructor arguments that differ from the member's defaults (``variant_kwargs``: what a checkpoint's model_config says
    about first_strides / classes / head activation).
    The mode is recorded as ``model.precision``; ``ensemble.member_input`` feeds each member the input dtype it was built for."""
    from . import ops, pipeline
    mode = precision or ops.PRECISION
    ctor = (lambda p: spec.ctor(p, **variant)) if variant else spec.ctor
    if mode in ("strict", "f32"):
        with ops.precision(mode):
            model = ctor(params)
        model.precision = mode
        return model
    with ops.precision("fast"):
        if os.environ.get("VIP_BIAS_CALIBRATION", "1") == "0":     # profiling runs: keep the calibration launches out of the trace
            bias_calibration = False
        if not (bias_calibration and torch.cuda.is_available()):
            model = ctor(params)
            model.precision = "fast"
            return model
        ops.KEEP_ROUNDING_ERROR = True
        try:
            model = ctor(params)
        finally:
            ops.KEEP_ROUNDING_ERROR = False
        model.precision = "fast"
        batch = calibration_batch if calibration_batch is not None else pipeline.calibration_batch()
        calibrate(model, batch.resized(spec.input_hw, spec.input_hw))
        torch.cuda.synchronize()
        return model


def calibrate(model, x):
    """The two calibration steps of a freshly constructed model (built under ``ops.KEEP_ROUNDING_ERROR``) on the batch ``x``:
    1. layer-wise: every conv / dense folds (W32 - W16) . E[input] into its fp32 bias (``ops.calibration``);
    2. whole-model, only with ``VIP_OFFSET_CALIBRATION=1``: what is left of the weight rounding at the logit is, to first order, the
       same offset for every image (the layer-wise step cannot see the part that goes through the nonlinearities and the zero-padded
       border taps).  Two more passes over the same batch with identical launches - fp16 weights as shipped (``ops.unfused``) and
       two-term ~22-bit weights with the uncorrected biases (``ops.exact_weights``) - and the difference of their mean logits is
       added to the head bias.  Inputs only, no labels, nothing but this library's own kernels.  OFF by default: measured on 64
       images it removes the weight offset (ResNeSt mean |dz| 8.9e-4 -> 6.6e-4, NFNet 2.2e-4 -> 1.9e-4) but on ResNet-RS-50 and
       GCViT that offset was cancelling part of the activation-rounding offset (5.0e-4 -> 6.8e-4, 1.37e-3 -> 1.47e-3), and on
       the EfficientNets the two legs' activation roundings differ by as much as the offset being estimated."""
    from . import ops
    with ops.calibration():
        model.logits(x)
    if os.environ.get("VIP_OFFSET_CALIBRATION", "0") == "1":
        with ops.unfused():
            z16 = model.logits(x).float()
        with ops.exact_weights():
            z22 = model.logits(x).float()
        model.head_b = (model.head_b + (z22 - z16).mean(0)[:model.head_b.numel()]).contiguous()
        model.offset_calibration = (z22 - z16).mean(0).tolist()
    ops.drop_exact_weights()
    return model


def build_member(name: str, calibrated: bool = True, bias_calibration: bool = True, precision: str = None,
                 calibration_batch=None) -> Tuple[MemberSpec, object]:
    spec = MEMBERS[name]
    return spec, construct(spec, build_params(name, calibrated), bias_calibration, precision, calibration_batch)


def by_ckpt_name(ckpt_name: str):
    """manifest directory name (ckpts/ckpts.json) -> registry key, or None if the graph is not built"""
    for k, spec in MEMBERS.items():
        if spec.ckpt_name == ckpt_name:
            return k
    return None


def _fold_mean_cls():
    from .pipeline import keras_predict

    @keras_predict
    class FoldMean:
        """mean over the fold checkpoints of one member (main.py:101-121)"""

        def __init__(self, folds):
            self.folds = folds
            self.precision = getattr(folds[0], "precision", "fast")

        def predict(self, x):
            ps = [m.predict(x) for m in self.folds]
            return ps[0] if len(ps) == 1 else sum(ps) / float(len(ps))
    return FoldMean


FoldMean = _fold_mean_cls()


def read_checkpoint(path: str):
    """One fold checkpoint -> ``{Keras variable name: torch tensor}``.  ``*.npz``: a flat dict of Keras-named arrays; ``*.h5`` /
    ``*.hdf5``: a Keras weight or model file (what the reference's ``ckpt/*.h5`` are, main.py:101-107,186-194), read by the pure-Python
    HDF5 reader ``h5lite`` - variable names with the ``:0`` suffix dropped, i.e. the same keys the ``.npz`` form uses."""
    if path.endswith((".h5", ".hdf5")):
        from . import h5lite
        arrays = h5lite.load_keras_weights(path)
    elif os.path.basename(path) == "saved_model.pb" or os.path.isdir(path):
        # a Keras SavedModel directory (main.py:103-104,186-194): the variables of variables/variables.{index,data-*} under their
        # graph names (tfbundle: written from the published formats, never pinned against TensorFlow output - see its header)
        from . import tfbundle
        arrays = tfbundle.load_savedmodel_weights(path)
    else:
        arrays = dict(np.load(path).items())
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrays.items()}


def match_variable_names(spec: MemberSpec, params):
    """A checkpoint exported from an enclosing Keras model carries that model's name in front of every variable
    (``convnext_tiny/stem/0/kernel`` for ``stem/0/kernel``): when the keys do not fit the graph as they are but do once ONE common leading
    scope is removed, remove it.  Anything else is left alone - the constructor then names the first missing variable."""
    if f"{spec.head}/kernel" in params:          # the classifier is where the graph expects it: nothing to do (and no synthesis cost)
        return params
    from . import timm_names
    if timm_names.looks_like_timm(params):
        # a timm (PyTorch) state_dict saved as .npz - where the tfimm members' weights come from: mapped by the reference's own rule
        # (tfimm/utils/timm.py:39-106) and re-laid-out (OIHW -> HWIO, Dense kernels transposed, :164-190)
        template = {k: tuple(v.shape) for k, v in spec.synth(spec.seed).items()}
        got = timm_names.from_timm_state_dict({k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in params.items()}, template)
        got.pop("__unused__", None)
        return {k: torch.from_numpy(v) for k, v in got.items()}
    want = set(spec.synth(spec.seed))
    if want <= set(params):
        return params
    keys = list(params)
    head = keys[0].split("/", 1)[0] + "/" if keys and "/" in keys[0] else None
    if head and all(k.startswith(head) for k in keys) and want <= {k[len(head):] for k in keys}:
        return {k[len(head):]: v for k, v in params.items()}
    return params


def load_model(path: str, compile: bool = False, precision: str = None, bias_calibration: bool = True, calibration_batch=None):
    """Counterpart of ``tf.keras.models.load_model(path, compile=False)`` (main.py:107) for this build's checkpoint formats:
    ``path`` = ``.../ckpts/<member directory>/ckpt/<fold>.npz`` (a flat dict of Keras-named arrays) or ``<fold>.h5`` (a Keras weight /
    model file, ``read_checkpoint``).  The member graph is picked from
    the directory name exactly as the reference picks its batch size from it (main.py:70-71,85); the returned object has the
    ``predict(dataset, steps, verbose) -> np.ndarray [n, C]`` of a Keras model.  ``precision`` / ``bias_calibration`` /
    ``calibration_batch``: see ``construct``; a strict model takes fp32 batches (``pipeline.build_dataset`` yields them when
    ``CFG.precision == "strict"`` or ``ops.PRECISION`` is)."""
    ap = os.path.abspath(path)
    if os.path.isdir(ap):                       # the SavedModel form: main.py:103-104 hands load_model the ckpt DIRECTORY
        ap = os.path.join(ap, "saved_model.pb")
    path = ap
    model_name = os.path.basename(os.path.dirname(os.path.dirname(ap)))
    key = by_ckpt_name(model_name)
    if key is None:
        raise ValueError(f"load_model: no graph for checkpoint directory {model_name!r}")
    spec = MEMBERS[key]
    variant = checkpoint_variant(spec, path)
    return construct(spec, match_variable_names(spec, read_checkpoint(path)), bias_calibration, precision, calibration_batch, variant)


# ---- graph variants from a checkpoint's model_config ---------------------------------------------------------------------------
# tf.keras.models.load_model (main.py:107) rebuilds the graph from the file's `model_config`, so whatever non-default constructor
# arguments the lost checkpoints were trained with (first_strides, number of classes, head activation) come with the
# file.  Here the member FAMILY is picked from the directory name (as the reference picks its batch size, main.py:70-71,85) and the
# variant arguments are read from model_config; a weight-only file (model.save_weights) has none and gets the constructor defaults.
def _walk_layers(cfg):
    """every {"class_name", "config"} node of a Keras model_config in serialisation order, nested models included"""
    if not isinstance(cfg, dict):
        return
    yield cfg
    inner = cfg.get("config")
    if isinstance(inner, dict):
        for layer in inner.get("layers", []) or []:
            yield from _walk_layers(layer)
    elif isinstance(inner, list):                      # old Sequential form: config IS the layer list
        for layer in inner:
            yield from _walk_layers(layer)


def variant_from_model_config(cfg: Optional[dict]) -> dict:
    """What a Keras ``model_config`` says about the arguments the member constructors expose:
    ``input_hw`` (InputLayer batch_input_shape), ``first_strides`` (a custom layer's own ``first_strides`` entry - gcvit ``Stem``,
    layers/embedding.py:25-29) or else ``stem_strides`` / ``stem_layer`` (RAW stride and name of the first convolution-class layer -
    ``Conv2D`` or a registered subclass like ``nfnets>ScaledStandardizedConv2D``; ``variant_kwargs`` maps it onto the family's
    ``first_strides``: HorNet's stem runs at twice that), ``n_layers``, ``classes`` and ``head_act`` (units / activation of the LAST Dense: resnet_rs_model.py:474-476, common_layers.py:278-283),
    and for a tfimm model (class ``...>ViT`` / ``...>ConvNeXt`` serialised by tfimm/models/serialization.py:21-89 as its config
    dataclass) the dataclass fields themselves under ``tfimm_cfg``.  Keys that the file does not determine are absent."""
    out: dict = {}
    if not cfg:
        return out
    first_conv, last_dense, n_layers = None, None, 0
    for node in _walk_layers(cfg):
        cname = str(node.get("class_name", ""))
        c = node.get("config") if isinstance(node.get("config"), dict) else {}
        short = cname.split(">")[-1]
        n_layers += 1
        if short == "InputLayer" and "input_hw" not in out:
            shp = c.get("batch_input_shape") or c.get("batch_shape")
            if shp and len(shp) == 4 and shp[1] and shp[2]:
                out["input_hw"] = (int(shp[1]), int(shp[2]))
        if "first_strides" in c and "first_strides" not in out:
            out["first_strides"] = int(c["first_strides"])
        # the stem convolution: the first layer of ANY convolution class that carries strides - plain `Conv2D`, or a registered
        # subclass such as `nfnets>ScaledStandardizedConv2D` (nfnets.py:41-81, the ECA-NFNet stem) - depthwise / separable ones excepted
        if first_conv is None and short.endswith("Conv2D") and not short.startswith(("Depthwise", "Separable")) and c.get("strides"):
            first_conv = c
        if short == "Dense":
            last_dense = c
        if short in ("ViT", "ConvNeXt") and "nb_classes" in c:
            out["tfimm_cfg"] = dict(c)
            out["classes"] = int(c["nb_classes"])
            if c.get("input_size"):
                out["input_hw"] = (int(c["input_size"][0]), int(c["input_size"][1]))
        if short == "GCViT":
            for k_src, k_dst in (("num_classes", "classes"), ("head_act", "head_act"), ("first_strides", "first_strides")):
                if k_src in c:
                    out[k_dst] = c[k_src]
    if "first_strides" not in out and first_conv is not None:
        st = first_conv["strides"]
        # RAW stride of the stem convolution: variant_kwargs turns it into the constructor's first_strides per family
        out["stem_strides"] = int(st[0] if isinstance(st, (list, tuple)) else st)
        out["stem_layer"] = str(first_conv.get("name", ""))
    out["n_layers"] = n_layers
    if last_dense is not None and "classes" not in out:
        out["classes"] = int(last_dense["units"])
        out["head_act"] = last_dense.get("activation") or "linear"
    return out


def variant_kwargs(spec: MemberSpec, info: dict) -> dict:
    """``variant_from_model_config`` output -> keyword arguments of ``spec.ctor`` (only what differs from the defaults is passed)"""
    kw: dict = {}
    if not info:
        return kw
    hw = info.get("input_hw")
    if hw is not None and tuple(hw) != (spec.input_hw, spec.input_hw):
        raise ValueError(f"{spec.ckpt_name}: the checkpoint's model_config was built for {hw[0]}x{hw[1]} inputs, the manifest says "
                         f"{spec.input_hw}x{spec.input_hw}")
    fam = spec.oracle
    classes, act, fs = info.get("classes"), info.get("head_act"), info.get("first_strides")
    if fam == "tfimm_ref":                              # dataclass fields (tfimm_models.variant keeps the ones the graph uses)
        c = info.get("tfimm_cfg") or {}
        for k in ("nb_classes", "patch_size", "first_down"):
            if k in c:
                kw[k] = c[k]
        if classes is not None and "nb_classes" not in kw:
            kw["nb_classes"] = classes
        return kw
    if fs is None and "stem_strides" in info:
        # the constructor argument behind the stem convolution's stride: HorNet's stem runs at first_strides * 2 (kecam
        # hornet/hornet.py:144), every other family's at first_strides itself (resnet_rs_model.py:97-104, aotnet.py:326,
        # efficientnet_v2.py:155-156, nfnets.py:182-191)
        ss = int(info["stem_strides"])
        if fam == "hornet_ref":
            if ss % 2:
                raise ValueError(f"{spec.ckpt_name}: stem convolution {info.get('stem_layer')!r} has stride {ss}; HorNet's is 2 * first_strides")
            fs = ss // 2
        else:
            fs = ss
    if fs is None and info.get("n_layers", 0) > 1 and fam != "gcvit_ref":
        raise ValueError(f"{spec.ckpt_name}: the checkpoint's model_config lists {info['n_layers']} layers but no stem convolution with "
                         "strides could be identified - refusing to assume first_strides")
    if classes is not None and classes != 1:
        kw["classes"] = int(classes)
    if fs is not None and fs != 2:
        kw["first_strides"] = int(fs)
    default_act = "sigmoid" if (classes or 1) == 1 else "softmax"
    if act is not None and act != default_act:
        if act not in ("sigmoid", "softmax", "linear"):
            raise ValueError(f"{spec.ckpt_name}: classifier activation {act!r} in model_config is not supported")
        kw["head_act" if fam == "gcvit_ref" else "classifier_activation"] = act
    return kw


def checkpoint_variant(spec: MemberSpec, path: str) -> dict:
    """constructor keyword arguments a checkpoint file asks for ({} for weight-only files and .npz)"""
    if os.path.basename(path) == "saved_model.pb" or os.path.isdir(path):
        from . import tfbundle
        return variant_kwargs(spec, variant_from_model_config(tfbundle.load_savedmodel_config(path)))
    if not path.endswith((".h5", ".hdf5")):
        return {}
    from . import h5lite
    return variant_kwargs(spec, variant_from_model_config(h5lite.load_keras_model_config(path)))
