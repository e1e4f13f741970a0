"""Experiment: does replaying a member's forward pass as a HIP graph (torch.cuda.CUDAGraph capture of the ctypes launches) beat eager
enqueueing?  Per member at B = 256: eager ms vs graph-replay ms.   python tools/try_graph.py [member ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import pipeline, zoo  # noqa: E402

names = sys.argv[1:] or zoo.ENSEMBLE8
raw = pipeline.calibration_batch(16)
rgb = raw.rgb.repeat(16, 1, 1, 1)
batch = pipeline.DecodedBatch(rgb, raw.sizes.repeat(16, 1), raw.sizes_host * 16)
for name in names:
    spec, model = zoo.build_member(name)
    x = batch.resized(spec.input_hw, spec.input_hw)
    for _ in range(3):
        y = model.predict(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = model.predict(x)
    e1.record()
    torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / 10 * 1e3
    eager = e0.elapsed_time(e1) / 10
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model.predict(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            yg = model.predict(x)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        graph = e0.elapsed_time(e1) / 10
        same = torch.equal(y, yg)
        print(f"{name:24s} eager {eager:7.2f} ms (host wall {host:6.2f} ms/iter)  graph {graph:7.2f} ms  ({100 * (eager - graph) / eager:+.1f} %)  identical={same}", flush=True)
    except Exception as ex:   # noqa: BLE001
        print(f"{name:24s} eager {eager:7.2f} ms  graph capture failed: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
