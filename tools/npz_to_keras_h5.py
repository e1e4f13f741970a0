"""Flat ``.npz`` of Keras-named arrays -> Keras-layout ``.h5`` weight file (root ``layer_names``, per-layer ``weight_names``, one
dataset per variable; keras/saving/hdf5_format.py ``save_weights_to_hdf5_group``), or with a third argument - a JSON file holding a
Keras ``model_config`` - a FULL-MODEL file (``save_model_to_hdf5``: the same tree under ``model_weights`` plus the root attributes
``model_config`` / ``keras_version`` / ``backend``).  Needs h5py:
    /opt/conda/bin/python3.9 tools/npz_to_keras_h5.py in.npz out.h5 [model_config.json]
Used by tests/test_h5lite_cpu.py to round-trip a whole member checkpoint through real libhdf5 output."""
import sys
import h5py
import numpy as np

src, dst = sys.argv[1], sys.argv[2]
arrays = np.load(src)
layers = {}
for k in arrays.files:
    layers.setdefault(k.rsplit("/", 1)[0], []).append(k)


def put_attr(g, name, vals, limit=64512):
    arr = np.asarray([v.encode() for v in vals])
    if arr.nbytes <= limit:
        g.attrs[name] = arr
        return
    n = 2
    while any(c.nbytes > limit for c in np.array_split(arr, n)):
        n += 1
    for i, c in enumerate(np.array_split(arr, n)):
        g.attrs[f"{name}{i}"] = c


with h5py.File(dst, "w") as top:
    f = top
    if len(sys.argv) > 3:
        top.attrs["model_config"] = open(sys.argv[3]).read()          # str -> variable-length UTF-8, as Keras writes it
        top.attrs["keras_version"] = "2.8.0"
        top.attrs["backend"] = "tensorflow"
        f = top.create_group("model_weights")
    put_attr(f, "layer_names", list(layers))
    f.attrs["backend"] = "tensorflow"
    f.attrs["keras_version"] = "2.8.0"
    for layer, names in layers.items():
        g = f.require_group(layer)          # a layer whose name is the prefix of another (gcvit: levels/0/blocks/0/attn[/qkv]) already exists
        put_attr(g, "weight_names", [n + ":0" for n in names])
        for n in names:
            g.create_dataset(n + ":0", data=arrays[n])
print(dst, len(arrays.files), "variables in", len(layers), "layers")
