"""Extracts the reference's architecture tables as DATA, without importing it (TensorFlow is not installed): the source
files are parsed with `ast` and only literal values are kept - module-level dict/list constants, default arguments,
literal assignments and literal call arguments of the constructor functions the shipped ensemble goes through.

    python tools/extract_reference_configs.py [/root/reference] -> tests/golden/ref_configs.json

tests/test_reference_configs.py compares the product's (and the oracle's) tables with this file."""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_configs.json")


def lit(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return None


def jsonable(v):
    if isinstance(v, dict):
        return {str(k): jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return v


def module_constants(path, names):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id in names and lit(node.value) is not None:
                out[node.targets[0].id] = jsonable(lit(node.value))
    return out


def function_literals(path, fname):
    """{'defaults': {arg: value}, 'assigns': {name: value}, 'calls': [{'func', 'args', 'kwargs'}]} - literals only"""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == fname:
            a = node.args
            pos = a.posonlyargs + a.args
            defaults = {}
            for arg, d in zip(pos[len(pos) - len(a.defaults):], a.defaults):
                if lit(d) is not None or (isinstance(d, ast.Constant) and d.value is None):
                    defaults[arg.arg] = jsonable(lit(d))
            for arg, d in zip(a.kwonlyargs, a.kw_defaults):
                if d is not None and lit(d) is not None:
                    defaults[arg.arg] = jsonable(lit(d))
            assigns, calls = {}, []
            for sub in ast.walk(node):
                if isinstance(sub, ast.Assign) and len(sub.targets) == 1 and isinstance(sub.targets[0], ast.Name):
                    if lit(sub.value) is not None:
                        assigns[sub.targets[0].id] = jsonable(lit(sub.value))
                if isinstance(sub, ast.Call):
                    f = sub.func
                    name = f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else None)
                    args = [jsonable(lit(x)) for x in sub.args if lit(x) is not None]
                    kwargs = {k.arg: jsonable(lit(k.value)) for k in sub.keywords if k.arg and lit(k.value) is not None}
                    if name and (args or kwargs):
                        calls.append({"func": name, "args": args, "kwargs": kwargs})
            return {"defaults": defaults, "assigns": assigns, "calls": calls}
    raise KeyError(f"{fname} not found in {path}")


K = "models/keras_cv_attention_models/"
T = "models/tfimm/architectures/"
out = {
    "_source": "awsaf49/vip-cup-2022 (reference checkout), parsed with ast by tools/extract_reference_configs.py",
    "main.py": module_constants("main.py", {"NAME2BS"}),
    "ckpts.json": json.load(open(os.path.join(REF, "ckpts", "ckpts.json"))) if os.path.exists(os.path.join(REF, "ckpts", "ckpts.json")) else None,
    "resnet_rs/block_args.py": module_constants("models/resnet_rs/block_args.py", {"BLOCK_ARGS"}),
    "gcvit/models/gcvit.py": module_constants("models/gcvit/models/gcvit.py", {"NAME2CONFIG"}),
    "efficientnet_v2.py": {n: function_literals(K + "efficientnet/efficientnet_v2.py", n)
                           for n in ("EfficientNetV2", "EfficientNetV2T", "EfficientNetV2M", "EfficientNetV2L")},
    "efficientnet_v1.py": {n: function_literals(K + "efficientnet/efficientnet_v1.py", n)
                           for n in ("get_expanded_width_depth", "EfficientNetV1", "EfficientNetV1B4")},
    "resnest.py": {n: function_literals(K + "resnest/resnest.py", n) for n in ("ResNest", "ResNest50", "ResNest200")},
    "nfnets.py": {n: function_literals(K + "nfnets/nfnets.py", n)
                  for n in ("NormFreeNet", "NormFreeNet_Light", "ECA_NFNetL0", "ECA_NFNetL2")},
    "hornet.py": {n: function_literals(K + "hornet/hornet.py", n)
                  for n in ("HorNet", "HorNetTiny", "HorNetSmall", "HorNetBase", "HorNetLarge", "gnconv", "block")},
    "resnet_deep.py": {n: function_literals(K + "resnet_family/resnet_deep.py", n) for n in ("ResNetD", "ResNet200D")},
    "aotnet.py": {n: function_literals(K + "aotnet/aotnet.py", n) for n in ("AotNet",)},
    "vit.py": {n: function_literals(T + "vit.py", n) for n in ("vit_tiny_patch16_224", "vit_small_patch16_224", "vit_base_patch16_224")},
    "convnext.py": {n: function_literals(T + "convnext.py", n)
                    for n in ("convnext_tiny_in22k", "convnext_small_in22k", "convnext_base_in22k", "convnext_large_in22ft1k",
                              "convnext_base_384_in22ft1k", "convnext_large_384_in22ft1k")},
}
json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
