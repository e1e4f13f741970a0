"""Benchmark workloads: which ensemble members run, on what resident input, and how a step is timed.

Used by bench.py and __graft_entry__.smoke().  No oracle imports here (the CPU baseline lives in
bench.py).
"""
from typing import Dict, List, Optional

import torch

from . import ensemble, ops, pipeline
from . import zoo

DEFAULT = "ensemble"


class KernelProfile:
    """HIP-event bracketing of individual launches on the launch stream (torch's current stream)."""

    def __init__(self):
        self.rec: List = []

    def start(self, family: str, flops: float, nbytes: float):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (family, flops, nbytes, e0, e1)

    def stop(self, tok):
        tok[4].record()
        self.rec.append(tok)

    def summary(self) -> Dict[str, Dict[str, float]]:
        torch.cuda.synchronize()
        out: Dict[str, Dict[str, float]] = {}
        for fam, fl, nb, e0, e1 in self.rec:
            d = out.setdefault(fam, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


class Workload:
    def __init__(self, name: str, members: List[str], batch: int, rank: int, world: int):
        self.name = name
        self.batch = batch
        self.rank = rank
        self.world = world
        self.members = members
        self.models = [zoo.build_member(m) for m in members]
        # resident input: decoded 200x200 RGB uint8 pixels (what tf.image.decode_jpeg yields, dataset.py:28)
        g = torch.Generator().manual_seed(1234 + rank)
        rgb = torch.randint(0, 256, (batch, 200, 200, 3), generator=g, dtype=torch.uint8).cuda()
        sizes = torch.tensor([[200, 200]] * batch, dtype=torch.int32, device="cuda")
        self.batch_rgb = pipeline.DecodedBatch(rgb, sizes, [(200, 200)] * batch)
        self.scores = None
        self._gather = None
        self.member_streams = ensemble.MemberStreams(ensemble.default_streams())
        self._serial = ensemble.MemberStreams(1)

    def step(self, dist=None, serial: bool = False):
        """cast+resize+/255 per member resolution (dataset.py:31-38), score the resident batch with every
        member, mean over members (main.py:142-143), all-gather across ranks."""
        cache = {}
        for spec, _ in self.models:
            hw = spec.input_hw
            if hw not in cache:
                cache[hw] = self.batch_rgb.resized(hw, hw)
        probs = (self._serial if serial else self.member_streams).predict_all(self.models, cache)
        s = torch.stack(probs, 0).mean(0).reshape(-1)
        if dist is not None and self.world > 1:
            if self._gather is None:       # flat: the concatenation form every backend accepts (gloo rejects the stacked one)
                self._gather = torch.empty((self.world * s.numel(),), dtype=s.dtype, device=s.device)
            dist.all_gather_into_tensor(self._gather, s)
            s = self._gather.view(self.world, -1)
        self.scores = s
        return s

    def config(self):
        return {"workload": self.name, "members": self.members, "batch_per_gpu": self.batch,
                "global_batch": self.batch * self.world,
                "input": "decoded 200x200 RGB u8 resident in HBM -> bicubic resize per member resolution -> fp16 NHWC",
                "parallelism": f"image-parallel dp{self.world}, all-gather of scores",
                "member_streams": self.member_streams.n}

    def roofline(self, peak_tflops: float, peak_gbs: float):
        """One instrumented step: per-kernel-family time from HIP events around each launch."""
        prof = KernelProfile()
        ops.set_profiler(prof)
        try:
            self.step(None, serial=True)      # one stream: per-launch HIP events see only their own kernel
        finally:
            ops.set_profiler(None)
        summ = prof.summary()
        self._summ = summ
        if not summ:
            return None
        fam = max(summ, key=lambda k: summ[k]["ms"])
        d = summ[fam]
        sec = d["ms"] * 1e-3
        tf = d["flops"] / sec / 1e12
        gbs = d["bytes"] / sec / 1e9
        intensity = d["flops"] / max(d["bytes"], 1.0)
        common = {"kernel": fam, "traffic": _pmc_traffic(fam), "traffic_unit": "HBM bytes/launch (rocprofv3 PMC, profiles/)",
                  "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], "launches": d["launches"],
                  "avg_launch_ms": d["ms"] / d["launches"]}
        if intensity * peak_gbs * 1e9 >= peak_tflops * 1e12:
            return {"bound": "mfma", "achieved": tf, "peak": peak_tflops, "unit": "TFLOP/s", "frac": tf / peak_tflops, **common}
        return {"bound": "hbm", "achieved": gbs, "peak": peak_gbs, "unit": "GB/s", "frac": gbs / peak_gbs, **common}

    def extra(self):
        summ = getattr(self, "_summ", None)
        if not summ:
            return None
        return {k: {"launches": v["launches"], "ms_per_step": round(v["ms"], 4),
                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                    "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}


def _pmc_traffic(fam: str):
    """HBM bytes per launch of a kernel family from the committed PMC summary of this bench command
    (tools/pmc_traffic.py over `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes), or None."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_hbm_traffic_pmc.json")
    try:
        return json.load(open(path))[fam]["hbm_bytes_per_launch"]
    except Exception:
        return None


def build(name: str, batch: int, rank: int = 0, world: int = 1) -> Workload:
    if name == "ensemble":
        members = zoo.ENSEMBLE
    elif name == "ensemble8":
        members = zoo.ENSEMBLE8
    elif name == "ensemble4":
        members = zoo.ENSEMBLE4
    else:
        members = [name]
    return Workload(name, members, batch, rank, world)
