"""Deterministic synthetic 200x200 JPEG set — the stand-in for the unreleased VIP-Cup
test images.  Image i = clip(low-frequency field + noise) saved as baseline JPEG with PIL
(libjpeg-turbo): quality cycles 50/60/70/80/90/95, 4:2:0 except every 4th image 4:4:4, every 50th
image is 256x192 (exercises the resize branch of dataset/dataset.py:33-34).

    python tools/make_synth.py --n 32 --out /tmp/synth   -> img_00000.jpg ... + test.csv (column `filename`)
"""
import argparse
import io
import os

import numpy as np
from PIL import Image

QUALITIES = (50, 60, 70, 80, 90, 95)


def synth_pixels(i: int, seed: int = 2022) -> np.ndarray:
    rng = np.random.default_rng(seed + i)
    if i % 50 == 49:
        h, w = 192, 256
    else:
        h, w = 200, 200
    low = rng.normal(128.0, 48.0, size=(8, 8, 3)).astype(np.float32)
    field = np.asarray(Image.fromarray(np.clip(low, 0, 255).astype(np.uint8)).resize((w, h), Image.BICUBIC),
                       dtype=np.float32)
    noise = rng.normal(0.0, 12.0, size=(h, w, 3)).astype(np.float32)
    return np.clip(field + noise, 0, 255).astype(np.uint8)


def synth_jpeg(i: int, seed: int = 2022) -> bytes:
    px = synth_pixels(i, seed)
    buf = io.BytesIO()
    sub = 0 if i % 4 == 0 else 2  # PIL: 0 = 4:4:4, 2 = 4:2:0
    Image.fromarray(px).save(buf, format="JPEG", quality=QUALITIES[i % 6], subsampling=sub)
    return buf.getvalue()


def synth_set(n: int, seed: int = 2022):
    return [synth_jpeg(i, seed) for i in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    names = []
    for i in range(a.n):
        name = f"img_{i:05d}.jpg"
        with open(os.path.join(a.out, name), "wb") as f:
            f.write(synth_jpeg(i, a.seed))
        names.append(name)
    with open(os.path.join(a.out, "test.csv"), "w") as f:
        f.write("filename\n" + "\n".join(names) + "\n")
    print(f"wrote {a.n} images + test.csv to {a.out}")


if __name__ == "__main__":
    main()
