"""Derive the Keras layer-name FRAGMENTS of the shipped ensemble members from the reference's constructors with `ast`
(nothing is imported or executed) -> tests/golden/ref_varnames.json.

A Keras variable is addressed as "<layer name>/<variable>" and the reference builds layer names by string concatenation
through nested helpers (`name and name + "conv"`, f"stack{i}_block{j}_", "blocks/{idx}" ...).  Evaluating those chains needs
the constructors to run; what an AST pass CAN pin is every string literal that takes part in a `name=` argument (or in an
assignment to a *name* variable), with format holes kept as holes.  tests/test_reference_varnames.py then requires every key
of this build's checkpoints (`spec.synth()`, the names `main.py` loads from .npz files) to be a concatenation of fragments of
its family - i.e. no layer name in this build is invented - and every weight-carrying fragment on the members' constructor
path to be used by some key.

    python tools/extract_reference_varnames.py [--ref /root/reference] [--out tests/golden/ref_varnames.json]
"""
import argparse
import ast
import json
import os
import re

HOLE = "\x00"      # an inherited / unknown string (a variable): splits a template into fragments
NUM = "\x01"       # a format hole inside a literal (f"blocks/{i}", "stack{}_".format(i)): a number

# family -> (source files, constructor functions / classes on the shipped members' path (reverse check), members of the family)
FAMILIES = {
    "resnet_rs": (["models/resnet_rs/resnet_rs_model.py"],
                  ["Conv2DFixedPadding", "STEM", "SE", "BottleneckBlock", "BlockGroup", "ResNetRS"]),
    "gcvit": (["models/gcvit/models/gcvit.py", "models/gcvit/layers/attention.py", "models/gcvit/layers/block.py",
               "models/gcvit/layers/embedding.py", "models/gcvit/layers/feature.py", "models/gcvit/layers/level.py"],
              ["GCViT", "WindowAttention", "GCViTBlock", "Stem", "Mlp", "SE", "ReduceSize", "FeatExtract", "GlobalQueryGen", "GCViTLevel"]),
    "tfimm_convnext": (["models/tfimm/architectures/convnext.py", "models/tfimm/layers/transformers.py", "models/tfimm/layers/norm.py"],
                       ["ConvNeXtBlock", "ConvNeXtStage", "ConvNeXt", "MLP"]),
    "tfimm_vit": (["models/tfimm/architectures/vit.py", "models/tfimm/layers/transformers.py"],
                  ["ViT", "ViTBlock", "ViTMultiHeadAttention", "MLP", "PatchEmbeddings"]),
    "kecam_resnest": (["models/keras_cv_attention_models/aotnet/aotnet.py", "models/keras_cv_attention_models/resnest/resnest.py",
                       "models/keras_cv_attention_models/common_layers.py"],
                      ["AotNet", "aot_stack", "aot_block", "deep_stem", "conv_shortcut_branch", "deep_branch", "attn_block",
                       "split_attention_conv2d", "batchnorm_with_activation", "conv2d_no_bias"]),
    "kecam_efficientnet": (["models/keras_cv_attention_models/efficientnet/efficientnet_v2.py",
                            "models/keras_cv_attention_models/efficientnet/efficientnet_v1.py",
                            "models/keras_cv_attention_models/common_layers.py"],
                           ["EfficientNetV2", "inverted_residual_block", "se_module", "batchnorm_with_activation", "conv2d_no_bias",
                            "output_block"]),
    "kecam_nfnet": (["models/keras_cv_attention_models/nfnets/nfnets.py", "models/keras_cv_attention_models/common_layers.py"],
                    ["NormFreeNet", "ScaledStandardizedConv2D", "ZeroInitGain", "std_conv2d_with_init", "block", "stack", "stem",
                     "eca_module", "output_block"]),
}
WEIGHT_CALLEES = ("Conv2D", "Conv1D", "Dense", "DepthwiseConv2D", "BatchNormalization", "LayerNormalization", "ScaledStandardizedConv2D",
                  "Conv2DFixedPadding", "add_weight", "norm_layer")


def sym(e):
    """expression -> list of alternative templates (strings with HOLE markers)"""
    if isinstance(e, ast.Constant):
        return [e.value] if isinstance(e.value, str) else [HOLE]
    if isinstance(e, ast.JoinedStr):
        outs = [""]
        for v in e.values:
            parts = [v.value] if isinstance(v, ast.Constant) else [NUM]
            outs = [o + p for o in outs for p in parts]
        return outs
    if isinstance(e, ast.BinOp) and isinstance(e.op, ast.Add):
        return [a + b for a in sym(e.left) for b in sym(e.right)]
    if isinstance(e, ast.BinOp) and isinstance(e.op, ast.Mod) and isinstance(e.left, ast.Constant) and isinstance(e.left.value, str):
        return [re.sub(r"%[sd]", NUM, e.left.value)]
    if isinstance(e, ast.BoolOp):                      # `name and name + "conv"` -> the last operand
        return sym(e.values[-1])
    if isinstance(e, ast.IfExp):
        return sym(e.body) + sym(e.orelse)
    if (isinstance(e, ast.Call) and isinstance(e.func, ast.Attribute) and e.func.attr == "format"
            and isinstance(e.func.value, ast.Constant) and isinstance(e.func.value.value, str)):
        return [re.sub(r"\{[^}]*\}", NUM, e.func.value.value)]
    return [HOLE]


def callee_name(call):
    f = call.func
    return f.attr if isinstance(f, ast.Attribute) else (f.id if isinstance(f, ast.Name) else "?")


def fragments_of(template):
    """literal pieces of a template (split at inherited strings); format holes inside a piece stay as \\d+"""
    out = []
    for piece in template.split(HOLE):
        if piece.strip(NUM) == "":
            continue
        out.append("".join(r"\d+" if ch == NUM else re.escape(ch) for ch in piece))
    return out


def scan(path, rel, on_path):
    src = open(path).read()
    tree = ast.parse(src)
    out = []

    def visit(node, scope):
        for child in ast.iter_child_nodes(node):
            sc = scope
            if isinstance(child, (ast.FunctionDef, ast.ClassDef)):
                sc = scope + [child.name]
            if isinstance(child, ast.Call):
                for kw in child.keywords:
                    if kw.arg == "name":
                        for tpl in sym(kw.value):
                            for fr in fragments_of(tpl):
                                out.append({"re": fr, "file": rel, "line": child.lineno, "callee": callee_name(child),
                                            "scope": ".".join(sc), "on_path": any(s in on_path for s in sc)})
                if callee_name(child) == "add_weight" and child.args and isinstance(child.args[0], ast.Constant):
                    out.append({"re": re.escape(str(child.args[0].value)), "file": rel, "line": child.lineno, "callee": "add_weight",
                                "scope": ".".join(sc), "on_path": any(s in on_path for s in sc)})
            if isinstance(child, ast.Assign) and len(child.targets) == 1 and isinstance(child.targets[0], ast.Name) \
                    and "name" in child.targets[0].id.lower():
                for tpl in sym(child.value):
                    for fr in fragments_of(tpl):
                        out.append({"re": fr, "file": rel, "line": child.lineno, "callee": "=" + child.targets[0].id,
                                    "scope": ".".join(sc), "on_path": any(s in on_path for s in sc)})
            visit(child, sc)
    visit(tree, [])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                  "ref_varnames.json"))
    a = ap.parse_args()
    result = {}
    for fam, (files, on_path) in FAMILIES.items():
        frs = []
        for rel in files:
            p = os.path.join(a.ref, rel)
            if os.path.exists(p):
                frs += scan(p, rel, set(on_path))
        seen, uniq = set(), []
        for f in frs:
            k = (f["re"], f["callee"], f["on_path"])
            if k not in seen:
                seen.add(k)
                uniq.append(f)
        result[fam] = {"fragments": uniq, "weight_callees": list(WEIGHT_CALLEES)}
        print(f"{fam}: {len(uniq)} fragments from {len(files)} files")
    json.dump(result, open(a.out, "w"), indent=1, sort_keys=True)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
