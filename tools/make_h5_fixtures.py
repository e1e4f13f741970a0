"""Writes the HDF5 fixtures of tests/test_h5lite_cpu.py with h5py (real libhdf5 output) in the layout Keras uses for ``.h5`` weight and
model files (keras/saving/hdf5_format.py: root attribute ``layer_names``, one group per layer with attribute ``weight_names`` and one
dataset per variable, nested by the '/' in the variable name; a full-model file keeps that tree under ``model_weights``).
The product's interpreter has no h5py; this image ships one for another interpreter:
    /opt/conda/bin/python3.9 tools/make_h5_fixtures.py
Outputs: tests/golden/h5/*.h5 and, next to each, the expected arrays (.npz, keys = variable names) and attributes (.json)."""
import json, os
import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "h5")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(2022)


def save_attr_chunked(g, name, vals, limit):
    """keras save_attributes_to_hdf5_group: split when the attribute would exceed the header limit"""
    arr = np.asarray([v.encode() for v in vals])
    if arr.nbytes <= limit:
        g.attrs[name] = arr
        return
    n = 2
    while any(c.nbytes > limit for c in np.array_split(arr, n)):
        n += 1
    for i, c in enumerate(np.array_split(arr, n)):
        g.attrs[f"{name}{i}"] = c


def write(fname, layers, full_model=False, libver=None, compress=(), attr_limit=64512):
    path = os.path.join(OUT, fname)
    expect = {}
    with h5py.File(path, "w", libver=libver) as f:
        root = f.create_group("model_weights") if full_model else f
        if full_model:
            f.attrs["keras_version"] = "2.8.0"                      # str -> variable-length UTF-8 (global heap)
            f.attrs["backend"] = "tensorflow"
            f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {"name": "m"}})
        save_attr_chunked(root, "layer_names", [l for l, _ in layers], attr_limit)
        root.attrs["backend"] = np.bytes_("tensorflow")             # fixed-length scalar string
        for lname, weights in layers:
            g = root.create_group(lname)
            save_attr_chunked(g, "weight_names", [w for w, _ in weights], attr_limit)
            for wname, val in weights:
                kw = {}
                if wname in compress:
                    kw = dict(chunks=tuple(max(1, s // 2 + 1) for s in val.shape), compression="gzip", shuffle=True, fletcher32=True)
                g.create_dataset(wname, data=val, **kw)
                expect[wname.rsplit(":", 1)[0]] = val
    np.savez(path[:-3] + ".npz", **expect)
    print(fname, os.path.getsize(path), "bytes,", len(expect), "variables")


def conv(name, kh, cin, cout, bias=True, dtype=np.float32):
    w = [(f"{name}/kernel:0", rng.standard_normal((kh, kh, cin, cout)).astype(dtype))]
    if bias:
        w.append((f"{name}/bias:0", rng.standard_normal(cout).astype(dtype)))
    return name, w


def bn(name, c):
    return name, [(f"{name}/{k}:0", rng.standard_normal(c).astype(np.float32)) for k in ("gamma", "beta", "moving_mean", "moving_variance")]


# 1. save_weights layout, a handful of layers, one layer without weights, nested variable scopes
write("keras_weights_small.h5", [conv("stem_conv_1", 3, 3, 8), bn("stem_bn_1", 8), ("activation", []),
                                 ("stack1/block1", [("stack1/block1/se/fc1/kernel:0", rng.standard_normal((8, 2)).astype(np.float32)),
                                                    ("stack1/block1/se/fc1/bias:0", np.zeros(2, np.float32))]),
                                 conv("predictions", 1, 8, 1)])
# 2. full-model layout, 40 layers (the group B-tree splits), gzip + shuffle + fletcher32 chunked datasets, fp16 and int64 variables,
#    weight_names / layer_names split into chunks (a tiny limit forces what 64 KB forces on a real model)
layers = []
for i in range(20):
    layers += [conv(f"block{i}_conv", 3, 4, 6, bias=(i % 2 == 0)), bn(f"block{i}_bn", 6)]
layers.append(("half_layer", [("half_layer/kernel:0", rng.standard_normal((5, 7)).astype(np.float16))]))
layers.append(("counter", [("counter/iterations:0", np.arange(6, dtype=np.int64).reshape(2, 3))]))
layers.append(("big", [("big/kernel:0", rng.standard_normal((3, 3, 40, 50)).astype(np.float32)),
                       ("big/ragged:0", rng.standard_normal((7, 11, 5)).astype(np.float32))]))
write("keras_model_many_layers.h5", layers, full_model=True, compress=("big/kernel:0", "big/ragged:0"), attr_limit=256)
# 3. libver='latest': superblock 3, version-2 object headers, link messages (compact new-style groups)
write("keras_weights_latest_format.h5", [conv("a", 1, 2, 3), bn("b", 3)], libver="latest")
