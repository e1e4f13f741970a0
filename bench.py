#!/usr/bin/env python3
"""bench.py — images/sec of the MI355X scoring path (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--shard images|members|hybrid]

Default workload = BASELINE.json config 5 ("ensemble8": the seven manifest members + tfimm ViT-S/16).  A "step" is one pass of the
hot path over one batch of synthetic input per rank, starting at the JPEG BYTES ("bytes in RAM -> scores"):
256 synthetic 200x200 JPEG byte strings resident in host RAM -> host Huffman decode into a page-locked buffer (C++ threads; one batch
of read-ahead on a worker thread, the overlap tf.data's prefetch gives the reference, dataset/dataset.py:101) -> H2D -> GPU dequant /
IDCT / upsample / YCbCr->RGB -> bicubic resize + /255 per member resolution -> forward of every ensemble member -> ONE RCCL all-gather
of the scores when N > 1 -> ensemble mean.  Scaling is weak: every rank brings its own batch of `--batch` images (one image-shard
per rank); `--shard` picks how the (member, image-shard) grid is dealt to the ranks (vipcup_amd/ensemble.py ShardPlan), the default
`images` keeps every member on every rank.  value = N * batch * K / t.

The round-1 variant (the step starts from decoded RGB u8 pixels already resident in HBM: the metric's "inputs resident" form) is
`--workload ensemble8-resident`; the default run times it as well and reports it under "detail".

Extra objects on the JSON line:
  roofline     — the dominant kernel family, timed live with HIP events on the launch stream; `peak` is the vendor figure of
                 MI355X_MICROARCH.md, `peak_measured` the on-box probe (vip_microbench_*), both fractions are given;
  cpu_baseline — the fp32 CPU oracle ("port": the reference's TF path cannot run anywhere here) on a bounded sample of the same
                 workload with the protocol of SURVEY.md section 8(d) (batch 128, first batch discarded, >= 3 timed batches, all host
                 cores of the box's share, Pillow decode + oracle resize + every member), rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (vendor)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (vendor)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workload", default="auto",
                    help="auto (= ensemble8) | ensemble8 | ensemble (7 manifest members) | ensemble4 | <member name>; "
                         "suffix -resident: start from decoded pixels in HBM")
    ap.add_argument("--shard", default="images", choices=["images", "members", "hybrid"])
    ap.add_argument("--precision", default=None, choices=["fast", "strict", "f32"],
                    help="fast (default): fp16 storage; strict: packed fp16 (hi, lo) pair storage, three fp16 MFMAs per product, fp32 "
                         "accumulate (every member logit within 1e-3 of the fp32 oracle, measured <= 4e-5); f32: fp32 storage (round 3's "
                         "strict mode).  The default run times the strict mode too and reports it under detail.strict_precision")
    ap.add_argument("--distinct-batches", type=int, default=20,
                    help="the steps cycle through this many distinct batches of synthetic JPEGs (20 x 256 = 5 120 distinct files: "
                         "BASELINE config 5's 5 000-image set)")
    ap.add_argument("--no-strict-leg", action="store_true")
    ap.add_argument("--no-batch-sweep", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batches", type=int, default=1, help="timed CPU batches (after the discarded warm-up images)")
    ap.add_argument("--cpu-batch", type=int, default=48, help="images per timed CPU batch (SURVEY 8(d) protocol: 128)")
    ap.add_argument("--cpu-warm", type=int, default=8, help="images of the discarded CPU warm-up batch (SURVEY 8(d) protocol: 128)")
    ap.add_argument("--no-resident-leg", action="store_true")
    return ap.parse_args()


def note(msg: str):
    """progress on stderr (stdout carries only the JSON line)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """Host cores this process may really use: the affinity mask, cut down to the cgroup CPU quota where one is set, and to the
    16-core share a one-GPU box of this pool gives its GPU (its affinity mask shows every core of the host; 256 torch threads on
    a 16-core share make the CPU leg crawl)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("VIP_CPU_THREADS", "16")))


def cpu_baseline(wl, timed_batches: int, bs: int = 48, warm: int = 8):
    """The fp32 CPU oracle (oracle/*: a port - the reference's TF/Keras path cannot run here) on a BOUNDED sample of the same
    workload: `warm` images discarded (tfimm/utils/profile.py:30-42 discards its first batch), then `timed_batches` batches of `bs`
    of the same synthetic JPEGs - about 30 s of CPU work at the defaults (SURVEY.md section 8(d)'s full protocol is batch 128, one
    discarded + three timed batches: `--cpu-batch 128 --cpu-warm 128 --cpu-batches 3`, ~4 minutes).  torch threads = every host
    core this process may use; per batch: Pillow (libjpeg-turbo) decode -> oracle bicubic resize + /255 per member resolution ->
    every member's oracle forward (the reference re-decodes per member, main.py:67,89 - decode is counted once per member
    resolution here, in its favour)."""
    import importlib
    import io
    import numpy as np
    from PIL import Image
    from oracle import ops_ref as R
    from vipcup_amd import zoo
    threads = host_cores()
    torch.set_num_threads(threads)
    pool = wl.jpegs * (max(bs, warm) // len(wl.jpegs) + 1)
    params = {m: zoo.build_params(m) for m in wl.members}
    refs = {m: importlib.import_module(f"oracle.{zoo.MEMBERS[m].oracle}") for m in wl.members}

    def one_batch(raws):
        pix = [np.asarray(Image.open(io.BytesIO(r)).convert("RGB")) for r in raws]
        inputs = {}
        for m in wl.members:
            hw = zoo.MEMBERS[m].input_hw
            if hw not in inputs:
                inputs[hw] = torch.stack([R.decode_resize_normalize(p, hw, hw) for p in pix])
        with torch.no_grad():
            for m in wl.members:
                refs[m].predict_logits(m, params[m], inputs[zoo.MEMBERS[m].input_hw])

    note(f"cpu_baseline: {threads} threads, {warm} images discarded + {timed_batches} timed batch(es) of {bs} x {len(wl.members)} members")
    one_batch(pool[:warm])                       # discarded
    t0 = time.perf_counter()
    for i in range(timed_batches):
        one_batch(pool[:bs])
        note(f"cpu_baseline: batch {i + 1}/{timed_batches} done, {time.perf_counter() - t0:.0f} s")
    dt = time.perf_counter() - t0
    return {"value": timed_batches * bs / dt, "unit": "images/sec", "cores": threads, "kind": "port", "protocol": "bounded sample",
            "sample": f"{timed_batches} timed batch(es) of {bs} synthetic JPEGs ({warm} images discarded first) x {len(wl.members)} members: "
                      f"Pillow decode + oracle resize + fp32 torch-CPU oracle forward, {threads} threads",
            "seconds": dt}


def timed_steps(wl, dist, steps, warmup):
    on_gpu = torch.cuda.is_available()

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    # pipelined steps (VIP_STEP_PIPELINE=0: join every step): step i is joined and scored after step i+1 has been forked; the step
    # left in flight is flushed INSIDE the bracket it was forked in, so the timed region holds exactly `steps` forks and `steps` joins
    pipe = os.environ.get("VIP_STEP_PIPELINE", "1") != "0"
    for _ in range(warmup):
        wl.step(dist, pipelined=pipe)
    wl.flush(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(dist, pipelined=pipe)
    wl.flush(dist)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this script (one per GPU) as CHILD processes through
    torch.distributed.run and relay their output - rank 0's JSON line goes to stdout as it is.  Runs before anything in this process
    has touched the GPU (no torch.cuda call yet), and never replaces the process (no exec)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"--gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # `--workload fake`: the launcher / rendezvous / timing / JSON plumbing of this file on CPU tensors over gloo, no kernels
    # (tests/test_bench_launcher_cpu.py) - never a measurement
    fake = a.workload == "fake"
    if not torch.cuda.is_available() and not fake:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; VIP_DIST_BACKEND=gloo lets several ranks share a card (a rehearsal of the N > 1 control flow on a
    # one-GPU box - RCCL itself refuses two ranks on one device)
    backend = "gloo" if fake else os.environ.get("VIP_DIST_BACKEND", "nccl")
    dev_index = 0
    if not fake:
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

    if fake:
        wl = workloads.FakeWorkload(a.batch, rank, world)
        dt = timed_steps(wl, dist, a.steps, a.warmup)
        if rank == 0:
            print(json.dumps({"metric": "images/sec (200x200, full ensemble)", "value": a.batch * world * a.steps / dt, "unit": "images/sec",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "fake",
                              "config": wl.config(), "roofline": None, "cpu_baseline": None,
                              "detail": {"checksum": float(wl.scores.sum()), "steps_seen": wl.steps_seen}}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    from vipcup_amd import ensemble, ops, zoo
    name = a.workload if a.workload != "auto" else workloads.DEFAULT
    mode = a.precision or ops.PRECISION
    if rank == 0:
        note(f"building workload {name} ({mode}: members" + (" + bias calibration)" if mode == "fast" else ")"))
    costs, models = None, None
    if a.shard == "hybrid" and world > 1:
        # the hybrid plan needs ms / image per member: measured HERE (rank 0, one serial pass, broadcast), every rank resident with
        # every member for the measurement; the plan then says which of them a rank keeps
        from tools.make_synth import synth_jpeg
        keys = workloads.member_list(name[:-len("-resident")] if name.endswith("-resident") else name)
        models = [zoo.build_member(k, precision=mode) for k in keys]
        costs = ensemble.measure_costs(models, [synth_jpeg(i) for i in range(min(a.batch, 64))], dist, rank)
        if rank == 0:
            note("hybrid plan from measured ms/image: " + ", ".join(f"{k} {c:.4f}" for k, c in zip(keys, costs)))
    wl = workloads.build(name, batch=a.batch, rank=rank, world=world, shard=a.shard, precision=mode, costs=costs,
                         distinct_batches=a.distinct_batches, models=models)
    if rank == 0:
        note(f"timing {a.warmup} + {a.steps} steps over {len(wl.jpeg_batches)} distinct batches")
    dt = timed_steps(wl, dist, a.steps, a.warmup)
    if rank == 0:
        note(f"{a.batch * world * a.steps / dt:.0f} images/s, {dt / a.steps * 1e3:.2f} ms/step")

    # the metric's "inputs already resident in HBM" form of the same step, for the record (never `value` here)
    resident = None
    if not wl.resident and not a.no_resident_leg:
        wr = workloads.Workload(wl.name, wl.members, a.batch, rank, world, a.shard, resident=True, jpegs=wl.jpegs, models=wl.models,
                                precision=mode, costs=costs)
        wr.member_streams = wl.member_streams                                # same resident members, same stream assignment
        dtr = timed_steps(wr, dist, a.steps, 1)
        resident = {"images_per_sec": a.batch * world * a.steps / dtr, "ms_per_step": dtr / a.steps * 1e3,
                    "input": "decoded RGB u8 resident in HBM (no Huffman / H2D / IDCT in the step)"}

    # the price of the stated tolerance, driver-timed: the same step in STRICT precision (packed fp16-pair storage, three MFMAs per
    # product) - checked IN THE RUN against the fast leg on the same batch (a strict step that mis-dispatched would still print img/s)
    strict = None
    if mode == "fast" and not a.no_strict_leg and a.shard == "images":
        if rank == 0:
            note("strict-precision leg: building packed-strict members")
        one = [wl.jpeg_batches[0]]
        wf = workloads.Workload(wl.name, wl.members, a.batch, rank, world, a.shard, jpegs=one, models=wl.models, precision=mode, costs=costs)
        fast_scores = wf.step(dist).float().cpu()
        wf.close()
        ws = workloads.build(name, batch=a.batch, rank=rank, world=world, shard=a.shard, precision="strict", jpegs=wl.jpeg_batches)
        wc = workloads.Workload(wl.name, wl.members, a.batch, rank, world, a.shard, jpegs=one, models=ws.models, precision="strict")
        strict_scores = wc.step(dist).float().cpu()
        wc.close()
        ops.h2_check("bench strict leg")                     # no activation left the fp16 range of the packed storage
        assert bool(torch.isfinite(strict_scores).all()), "strict leg: non-finite scores"
        thr = 0.487
        near = (strict_scores - thr).abs() < 5e-3             # the fast path may differ from fp32 by its storage floor near the threshold
        flips = int((((strict_scores > thr) != (fast_scores > thr)) & ~near).sum())
        dmax = float((strict_scores - fast_scores).abs().max())
        assert flips == 0 and dmax < 2e-2, f"strict leg disagrees with the fast leg: {flips} decision flips, max |dp| {dmax:.3e}"
        k_strict = max(3, min(a.steps, 5))
        dts = timed_steps(ws, dist, k_strict, 1)
        sroof = None
        if rank == 0:
            speaks = {"mfma_tflops": PEAK_MFMA_F16_TFLOPS / 3.0, "hbm_gbs": PEAK_HBM_GBS}      # three MFMAs per product
            sroof = ws.roofline(speaks)
            if sroof is not None:
                sroof["note"] = ("dominant strict kernel family; mfma peak = dense fp16 peak / 3 (three v_mfma_f32_16x16x32_f16 per "
                                 "fragment pair), bytes = 4 per element (an fp16 pair)")
        strict = {"images_per_sec": a.batch * world * k_strict / dts, "ms_per_step": dts / k_strict * 1e3, "steps": k_strict,
                  "arithmetic": "packed storage: every activation / weight an fp16 (hi, lo) pair (22 bits, 4 bytes); GEMMs = the fast path's "
                                "kernels with three v_mfma_f32_16x16x32_f16 per fragment pair, fp32 accumulate; fp32 LayerNorm / softmax / "
                                "activations / depthwise on the joined values",
                  "gemm": "h2",
                  "in_run_check": {"vs": "fast leg, same 256 images", "max_abs_dp": dmax, "decision_flips_outside_5e-3_of_thr": flips,
                                   "finite": True, "fp16_range_guard": "clear"},
                  "roofline": sroof,
                  "kernel_families": ws.extra() if rank == 0 else None,
                  "parity": "every member's calibrated logit and logit(ensemble mean) within 1e-3 of the fp32 oracle "
                            "(tests/test_gpu_strict.py: measured <= 4e-5 / <= 8e-6)"}
        if rank == 0:
            note(f"strict: {strict['images_per_sec']:.0f} images/s, {strict['ms_per_step']:.2f} ms/step")
        ws.close()
        del ws
        torch.cuda.empty_cache()

    # batch sweep for the default step (the deep stages run at M = B x 49 rows: larger batches fill the chip better)
    sweep = None
    if not a.no_batch_sweep and world == 1 and not wl.resident:
        sweep = {str(a.batch): {"images_per_sec": a.batch * a.steps / dt, "ms_per_step": dt / a.steps * 1e3}}
        for bsz in (512, 1024):
            if bsz == a.batch:
                continue
            flat = [j for b in wl.jpeg_batches for j in b]
            if len(flat) < bsz:
                continue
            batches = [flat[i:i + bsz] for i in range(0, len(flat) - bsz + 1, bsz)]
            wb = workloads.Workload(wl.name, wl.members, bsz, rank, world, a.shard, jpegs=batches, models=wl.models, precision=mode)
            dtb = timed_steps(wb, dist, 5, 2)
            sweep[str(bsz)] = {"images_per_sec": bsz * 5 / dtb, "ms_per_step": dtb / 5 * 1e3}
            note(f"batch {bsz}: {sweep[str(bsz)]['images_per_sec']:.0f} images/s")
            wb.close()
            del wb
            torch.cuda.empty_cache()

    roof, peaks = None, None
    if rank == 0:
        note("peak probes + instrumented step")
        peaks = {"mfma_tflops": PEAK_MFMA_F16_TFLOPS, "hbm_gbs": PEAK_HBM_GBS}
        peaks.update(workloads.measure_peaks())
        if mode == "strict":      # a logical product costs three matrix instructions on the packed storage: price the MFMA side for that
            roof = wl.roofline(dict(peaks, mfma_tflops=PEAK_MFMA_F16_TFLOPS / 3.0))
            if roof is not None:
                roof["note"] = ("mfma_frac against the dense fp16 peak / 3 (three v_mfma_f32_16x16x32_f16 per fragment pair); bytes = 4 per "
                                "element (an fp16 pair)")
        else:
            roof = wl.roofline(peaks)
        if roof is not None and any(m.startswith("gcvit") for m in wl.members) and len(wl.members) == 1:
            # BASELINE config 3 / SURVEY 8(d): the north-star kernel's roofline per level and FLOP-weighted
            lv = wl.attention_levels(peaks)
            if lv is not None:
                roof["levels"] = lv["levels"]
                roof["flop_weighted"] = lv["flop_weighted"]
                roof["levels_definition"] = lv["definition"]
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(wl, max(1, a.cpu_batches), max(1, a.cpu_batch), max(1, a.cpu_warm))

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
            "dtype": {"fast": "f16", "strict": "f16x2 (hi, lo) pairs, f32 accumulate", "f32": "f32"}[mode],
            "data": "synthetic",
            "config": wl.config(),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        detail = {"kernel_families": wl.extra(), "resident_input_variant": resident, "strict_precision": strict,
                  "batch_sweep": sweep,
                  "peaks": {"vendor": {"mfma_f16_tflops": PEAK_MFMA_F16_TFLOPS, "hbm_gbs": PEAK_HBM_GBS},
                            "measured_on_box": {k: (round(v, 1) if isinstance(v, float) else v) for k, v in peaks.items() if k.endswith("_measured")},
                            "guide_achievable_hbm_gbs": 6290.0,
                            "note": "peak_measured (HBM) = the best of three copy probes on THIS box (vip_microbench_copy_variant); "
                                    "MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy"}}
        if world > 1:
            detail["shard_plan"] = wl.plan.describe()
        line["detail"] = detail
        print(json.dumps(line))
    wl.close()
    if dist is not None:
        dist.barrier()                      # rank 0 is still in its instrumented step / JSON line: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
