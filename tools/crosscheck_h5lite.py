"""Cross-check of vipcup_amd/h5lite.py against h5py on HDF5 files it was not written around (the PyTables test-suite files of this
image).  Run with the interpreter that has h5py:   /opt/conda/bin/python3.9 tools/crosscheck_h5lite.py [dir]
Every numeric / fixed-string dataset both can open must be bit-identical; anything h5lite refuses must be refused loudly (H5Unsupported)."""
import glob, importlib.util, os, sys
import h5py
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("h5lite", os.path.join(root, "vip-cup-2022_amd", "h5lite.py"))
h5lite = importlib.util.module_from_spec(spec)
spec.loader.exec_module(h5lite)
d = sys.argv[1] if len(sys.argv) > 1 else "/opt/conda/lib/python3.9/site-packages/tables/tests"
same = refused = skipped = 0
bad = []
for path in sorted(glob.glob(os.path.join(d, "*.h5"))):
    try:
        hf = h5py.File(path, "r")
    except Exception:
        continue
    items = []
    hf.visititems(lambda n, o: items.append((n, o)) if isinstance(o, h5py.Dataset) else None)
    try:
        lf = h5lite.File(path)
    except h5lite.H5Unsupported:
        refused += len(items)
        continue
    for n, o in items:
        try:
            simple = o.dtype.kind in "iufS" and o.dtype.fields is None and o.shape is not None
        except Exception:                 # types h5py itself cannot map (float128, ...)
            simple = False
        if not simple:
            skipped += 1
            continue
        try:
            got = lf[n].read()
        except (h5lite.H5Unsupported, KeyError) as e:
            refused += 1
            continue
        try:
            want = o[()]
        except Exception:                 # filters this h5py build lacks (lzo, bzip2)
            skipped += 1
            continue
        if got.shape == np.shape(want) and got.tobytes() == np.asarray(want).tobytes():
            same += 1
        else:
            bad.append((os.path.basename(path), n, got.dtype, o.dtype, got.shape, o.shape))
print(f"identical: {same}   refused loudly: {refused}   skipped (compound / vlen / object types): {skipped}   MISMATCHES: {len(bad)}")
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
