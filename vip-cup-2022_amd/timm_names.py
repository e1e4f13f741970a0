"""timm (PyTorch) state_dict -> the Keras variable names / layouts the tfimm members are built from.

The reference's ViT and ConvNeXt members are ports of timm models: their weights originate as a PyTorch ``state_dict`` and are mapped
onto the Keras variables by name (``/root/reference/models/tfimm/utils/timm.py:39-106``, ``convert_tf_weight_name_to_pt_weight_name``)
and re-laid-out (``:109-229``, ``load_pytorch_weights_in_tf2_model``: OIHW -> HWIO for 4-D kernels ``:164-170``, a plain transpose for
Dense kernels, ``.gamma`` / ``.beta`` parameters renamed first ``:121-135``, then squeeze / expand / reshape to the variable's shape
``:178-190``).  This module restates that rule so that a timm ``state_dict`` saved as ``.npz`` (``np.savez(path, **{k: v.numpy()})``)
loads into ``zoo`` like any other checkpoint: ``zoo.read_checkpoint`` recognises the PyTorch naming (dots, ``.weight``) and calls
``from_timm_state_dict``.  Pure numpy; nothing of timm or torch.nn is needed.
"""
import re
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

NO, SIMPLE, CONV2D = "no", "simple", "conv2d"           # TransposeType (timm.py:30-36)


def tf_to_timm_name(tf_name: str, rank: Optional[int] = None, has_scope: bool = False) -> Tuple[str, str]:
    """A Keras variable name -> (PyTorch parameter name, transposition), the rule of timm.py:39-106.

    ``has_scope``: the name still carries the model's own top-level scope ("convnext_tiny/stem/0/kernel:0" - what ``variable.name``
    gives the reference), which the rule drops (``:76-77``); this build's checkpoint keys are stored without it."""
    n = tf_name.replace(":0", "")
    n = re.sub(r"/[^/]*___([^/]*)/", r"/\1/", n)                      # '$1___$2' -> $2
    n = n.replace("_._", "/")
    n = n.replace("/remove/", "/")
    n = re.sub(r"//+", "/", n)
    parts = n.split("/")
    if has_scope and len(parts) > 1:
        parts = parts[1:]
    last = parts[-1]
    if last in ("kernel", "depthwise_kernel") and rank == 4:
        tr = CONV2D
    elif last in ("kernel", "pointwise_kernel", "depthwise_kernel") or "emb_projs" in parts or "out_projs" in parts:
        tr = SIMPLE
    else:
        tr = NO
    if last in ("kernel", "depthwise_kernel", "embeddings", "gamma"):
        parts[-1] = "weight"
    elif last == "beta":
        parts[-1] = "bias"
    elif last == "moving_mean":
        parts[-1] = "running_mean"
    elif last == "moving_variance":
        parts[-1] = "running_var"
    return ".".join(parts), tr


def normalise_state_dict(state: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """timm.py:121-135: parameters that PyTorch itself calls ``.beta`` / ``.gamma`` (ResMLP affine, ConvNeXt layer scale) are renamed
    to ``.bias`` / ``.weight`` so that the rule above (TF ``beta`` -> ``bias``, ``gamma`` -> ``weight``) finds them."""
    out = {}
    for k, v in state.items():
        if k.endswith(".beta"):
            k = k[:-len(".beta")] + ".bias"
        elif k.endswith(".gamma"):
            k = k[:-len(".gamma")] + ".weight"
        out[k] = np.asarray(v)
    return out


def to_keras_layout(array: np.ndarray, transpose: str, shape: Tuple[int, ...]) -> np.ndarray:
    """timm.py:164-190: OIHW -> HWIO / transpose, then squeeze, expand or reshape to the Keras variable's shape"""
    a = np.asarray(array)
    if transpose == CONV2D:
        a = np.transpose(a, (2, 3, 1, 0))
    elif transpose == SIMPLE:
        a = np.transpose(a)
    if len(shape) < a.ndim:
        a = np.squeeze(a)
    elif len(shape) > a.ndim:
        a = np.expand_dims(a, 0)
    if tuple(a.shape) != tuple(shape):
        a = np.reshape(a, shape)                                     # raises like the reference when the sizes differ
    return np.ascontiguousarray(a)


def looks_like_timm(keys: Iterable[str]) -> bool:
    """PyTorch naming: dotted paths ending in weight / bias / running_* (a Keras checkpoint uses '/' and kernel / gamma / beta)"""
    keys = list(keys)
    return bool(keys) and all("/" not in k for k in keys) and any(k.endswith((".weight", ".bias")) for k in keys)


def from_timm_state_dict(state: Dict[str, np.ndarray], template: Dict[str, Tuple[int, ...]], allow_missing: Iterable[str] = ()) -> Dict[str, np.ndarray]:
    """``state``: a timm state_dict as numpy arrays; ``template``: {Keras variable name: shape} of the member (``zoo.build_params``
    gives it).  Returns {Keras name: array in Keras layout}.  A variable whose PyTorch name is absent raises (timm.py:152-161) unless
    its name matches one of the ``allow_missing`` regular expressions; PyTorch entries nothing asked for are reported in the
    returned dict under the key ``"__unused__"`` (``num_batches_tracked`` excepted, timm.py:205-209)."""
    st = normalise_state_dict(state)
    out: Dict[str, np.ndarray] = {}
    used = set()
    for name, shape in template.items():
        pt, tr = tf_to_timm_name(name, len(shape))
        if pt not in st:
            if any(re.search(pat, name) for pat in allow_missing):
                continue
            raise KeyError(f"{pt} (for Keras variable {name}) not found in the PyTorch state_dict")
        out[name] = to_keras_layout(st[pt], tr, tuple(shape))
        used.add(pt)
    out["__unused__"] = np.array(sorted(k for k in st if k not in used and "num_batches_tracked" not in k), dtype=object)
    return out
