"""Which Python lines of the predict path cause torch copy kernels / D2D memcpys (VERDICT r02 housekeeping: 164 copyBuffer + 11
at::native direct_copy<Half> per step)?  One serial step of the bench workload under torch.profiler with stacks; prints, per
(op, innermost vipcup_amd frame), the number of calls.

    python tools/find_copies.py [--workload ensemble8] [--batch 256]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ensemble8")
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    wl = workloads.build(a.workload, batch=a.batch)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        wl.step(serial=True)
        torch.cuda.synchronize()
    agg = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        n = ev.name
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            names[n[:60]] += 1
        if not n.startswith("aten::"):
            continue
        if n not in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::stack", "aten::fill_", "aten::zero_",
                     "aten::to", "aten::_to_copy", "aten::zeros", "aten::add", "aten::div", "aten::mul", "aten::sum", "aten::mean",
                     "aten::index", "aten::select", "aten::slice"):
            if not any(k in n for k in ("copy", "cat", "clone", "fill")):
                continue
        frame = "?"
        for fr in ev.stack:
            if "vip-cup-2022_amd" in fr or "vipcup" in fr:
                frame = fr.split("vip-cup-2022_amd/")[-1]
                break
        if n in ("aten::select", "aten::slice", "aten::to", "aten::contiguous"):
            continue
        agg[(n, frame)] += 1
    print("device kernels / memcpys in one serial step (top 25 by count):")
    for n, c in names.most_common(25):
        print(f"  {c:5d}  {n}")
    print("aten ops that launch copies / fills, by innermost product frame:")
    for (n, frame), c in agg.most_common(60):
        print(f"  {c:5d}  {n:18s} {frame}")
    wl.close()


if __name__ == "__main__":
    main()
