"""Serial per-member time of the ensemble workload (B=256), HIP events around each member's predict().
    python tools/member_times.py [8]        (VIP_PRECISION=strict: the strict-mode members on fp32 inputs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops, workloads
wl = workloads.build("ensemble8" if len(sys.argv) > 1 and sys.argv[1] == "8" else "ensemble", 256, resident=True)
for _ in range(2):
    wl.step(serial=True)
cache = {}
for spec, _ in wl.models:
    if spec.input_hw not in cache:
        cache[spec.input_hw] = wl._resident_batch.resized(spec.input_hw, spec.input_hw, dtype=ops.act_dtype())
torch.cuda.synchronize()
tot = 0.0
for spec, model in wl.models:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        model.predict(cache[spec.input_hw])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    tot += ms
    print(f"{spec.name:24s} {spec.input_hw:4d}px {ms:8.2f} ms  {256/ms*1e3:9.0f} img/s")
print(f"sum {tot:.2f} ms")
