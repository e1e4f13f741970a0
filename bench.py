#!/usr/bin/env python3
"""bench.py — images/sec of the MI355X scoring path (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: decoded 200x200 RGB u8 images (what tf.image.decode_jpeg yields) -> bicubic resize + /255 per member resolution ->
forward of every ensemble member -> ensemble-mean scores (-> RCCL all-gather of the scores when N > 1).  Scaling is weak: every rank scores its own batch of
`--batch` images with all members (image-parallel sharding, SURVEY.md §8e second form), so
value = N * batch * K / t.

Extra objects on the JSON line:
  roofline     — the dominant kernel family, timed live with HIP events on the launch stream;
  cpu_baseline — the fp32 CPU oracle ("port": the reference's TF path cannot run anywhere here) on a
                 bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workload", default="auto", help="auto | resnet_rs50 | gcvit_tiny | ensemble")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=32)
    return ap.parse_args()


def cpu_baseline(wl, n_images):
    """The fp32 CPU oracle (oracle/*, a port — the reference's TF/Keras path cannot run here) timed on the
    host cores over a bounded sample of the same workload: n_images images through every member,
    batch 16, first batch discarded (protocol of tfimm/utils/profile.py:30-42)."""
    import importlib
    from vipcup_amd import zoo
    threads = min(16, os.cpu_count() or 1)  # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(threads)
    bs = 16
    total_s = 0.0
    g = torch.Generator().manual_seed(99)
    for name in wl.members:
        spec = zoo.MEMBERS[name]
        ref = importlib.import_module(f"oracle.{spec.oracle}")
        params = zoo.build_params(name)
        x = torch.rand((bs, spec.input_hw, spec.input_hw, 3), generator=g)
        with torch.no_grad():
            ref.predict_logits(name, params, x)  # discarded
            t0 = time.perf_counter()
            for _ in range(max(1, n_images // bs)):
                ref.predict_logits(name, params, x)
            total_s += time.perf_counter() - t0
    n = max(1, n_images // bs) * bs
    return {"value": n / total_s, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic images x {len(wl.members)} member(s), batch {bs}, fp32 torch-CPU oracle"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; VIP_DIST_BACKEND=gloo lets several ranks share a card (a rehearsal of the N > 1 control flow on a
    # one-GPU box - RCCL itself refuses two ranks on one device)
    backend = os.environ.get("VIP_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    import vipcup_amd  # noqa: F401
    from vipcup_amd import workloads

    name = a.workload if a.workload != "auto" else workloads.DEFAULT
    wl = workloads.build(name, batch=a.batch, rank=rank, world=world)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.step(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step(dist)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roof = wl.roofline(PEAK_MFMA_F16_TFLOPS, PEAK_HBM_GBS) if rank == 0 else None
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(wl, a.cpu_images)

    if rank == 0:
        images = a.batch * world * a.steps
        line = {
            "metric": "images/sec (200x200, full ensemble)",
            "value": images / dt,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": wl.config(),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        extra = wl.extra()
        if extra:
            line["detail"] = extra
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()                      # rank 0 is still in its instrumented step / JSON line: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
