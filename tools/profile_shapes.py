"""Per-shape time breakdown of the GEMM-like launches of a workload step: kernel (as the C dispatcher names it), shape, launches, time,
TFLOP/s and algorithmic GB/s - from HIP events around each launch on a single stream (workloads.KernelProfile).
    python tools/profile_shapes.py [workload] [batch] [top]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import ops, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ensemble8"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 60
wl = workloads.build(name, B, resident=True)
for _ in range(2):
    wl.step(serial=True)
torch.cuda.synchronize()
prof = workloads.KernelProfile()
ops.set_profiler(prof)
wl.step(serial=True)
ops.set_profiler(None)
rows = prof.by_shape()
tot = sum(r[3] for r in rows)
print(f"total instrumented time {tot:.2f} ms over {sum(r[2] for r in rows)} launches")
fam = {}
for k, tag, n, ms, fl, by in rows:
    fam[k] = fam.get(k, 0.0) + ms
print("  ".join(f"{k}: {v:.2f} ms" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
# "floor": the larger of the algorithmic bytes at HBM_GBS and the MFMA work (three matrix instructions per logical one on the packed
# storage) at MFMA_TF - rates the best kernels of this library sustain, not the vendor peaks; "gap" = time above that floor
HBM_GBS = float(os.environ.get("VIP_FLOOR_GBS", "5500"))
MFMA_TF = float(os.environ.get("VIP_FLOOR_TF", "1150"))
gap_tot = 0.0
for k, tag, n, ms, fl, by in rows[:TOP]:
    mult = 3.0 if k.startswith("h2:") else 1.0
    floor = max(by / HBM_GBS / 1e6, mult * fl / MFMA_TF / 1e9)
    gap_tot += ms - floor
    print(f"{ms:7.3f} ms {100 * ms / tot:5.1f}% n={n:3d} {fl / ms / 1e9:7.1f} TF {by / ms / 1e6:7.0f} GB/s  floor {floor:6.3f} gap {ms - floor:6.3f}  {k:26s} {tag}")
print(f"time above the floor ({HBM_GBS:.0f} GB/s, {MFMA_TF:.0f} TF): {gap_tot:.2f} ms of {tot:.2f} ms")
