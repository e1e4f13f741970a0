"""End-to-end run of the drop-in CLI on a synthetic 5 000-image 200x200 JPEG set (BASELINE.json configs[4]):
file read -> host Huffman decode -> PCIe -> GPU IDCT/colour/resize -> every ensemble member -> CSV.
    python tools/e2e_5000.py [n_images] [batch]
Writes the set under $TMPDIR, runs vip-cup-2022_amd/main.py --synthetic twice (first run pays model build + page-in)
and prints the CLI's own "TIME TO INFER" lines."""
import io, os, subprocess, sys, tempfile
import numpy as np
from PIL import Image

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
batch = sys.argv[2] if len(sys.argv) > 2 else "256"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(prefix="vip5000_")
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:200, 0:200].astype(np.float32)
names = []
for i in range(n):
    f = rng.uniform(0.01, 0.08, size=(3, 2))
    ph = rng.uniform(0, 6.28, size=3)
    img = np.stack([127 + 90 * np.sin(f[c, 0] * xx + f[c, 1] * yy + ph[c]) for c in range(3)], -1)
    img += rng.normal(0, 12, img.shape)
    name = f"img_{i:05d}.jpg"
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(d, name), quality=int(rng.integers(75, 96)),
                                                                 subsampling=int(rng.integers(0, 3)))
    names.append(name)
with open(os.path.join(d, "input.csv"), "w") as fh:
    fh.write("filename\n" + "\n".join(names) + "\n")
print(f"wrote {n} JPEGs to {d}", flush=True)
for run in range(2):
    r = subprocess.run([sys.executable, os.path.join(root, "vip-cup-2022_amd", "main.py"), os.path.join(d, "input.csv"),
                        os.path.join(d, "out.csv"), "--synthetic", "--batch-size", batch], capture_output=True, text=True)
    tail = [l for l in r.stdout.splitlines() if "TIME TO INFER" in l or "FINAL" in l]
    print(f"run {run}: rc={r.returncode}", *tail, sep="\n  ", flush=True)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        break
print(open(os.path.join(d, "out.csv")).read()[:200])
