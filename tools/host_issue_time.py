"""Host-side enqueue time of one workload step vs its GPU time (is the issue thread the bottleneck?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import workloads
wl = workloads.build(sys.argv[1] if len(sys.argv) > 1 else "ensemble", int(sys.argv[2]) if len(sys.argv) > 2 else 256)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
