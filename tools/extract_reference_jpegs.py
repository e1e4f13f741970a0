"""Generating script for tests/golden/ref_{dog_cat,cat,dog}.jpg: the three baseline JPEG images that the
reference embeds as byte strings in models/keras_cv_attention_models/test_images.py:6-15 (the only
fixture data the reference ships).  Runs in the build container only (/root/reference is absent on the GPU
box); the extracted files are DATA (images), not reference source text.  Also records, next to each file, the
Pillow (libjpeg-turbo) decode statistics the oracle is pinned to.
"""
import importlib.util
import json
import os

import numpy as np

REF = "/root/reference/models/keras_cv_attention_models/test_images.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    spec = importlib.util.spec_from_file_location("ref_test_images", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # numpy + PIL only
    os.makedirs(OUT, exist_ok=True)
    meta = {}
    for name in ("dog_cat", "cat", "dog"):
        raw = getattr(mod, f"__{name}__")
        with open(os.path.join(OUT, f"ref_{name}.jpg"), "wb") as f:
            f.write(raw)
        px = getattr(mod, name)()
        meta[name] = {"bytes": len(raw), "shape": list(px.shape), "mean": float(px.mean()),
                      "sum": int(px.astype(np.int64).sum()),
                      "crc_rows": [int(px[r].astype(np.int64).sum()) for r in (0, 100, 255, 511)]}
    with open(os.path.join(OUT, "ref_jpeg_pillow_stats.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
