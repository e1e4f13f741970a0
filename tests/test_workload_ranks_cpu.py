"""bench.py's workload object on 2 ranks over gloo, with fake members and a fake device stage (no GPU): every ShardPlan mode, joined and
pipelined steps - the score payloads of the (member, image-shard) units each rank owns are packed, exchanged with ONE all-gather per
step and unpacked into the same ensemble scores on every rank, in the order the steps were forked (the path the driver's 2/4/8-GPU bench
runs and no single-GPU test can)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAINS = [1.0, 3.0, 5.0, 11.0, 2.0]


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import contextlib
    import torch.distributed as dist
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble, ops, pipeline, workloads

    class Stream:
        def wait_event(self, ev):
            pass

    class Event:
        def __init__(self, *a, **k):
            pass

        def record(self, *a):
            pass

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return 1.0

    main = Stream()
    torch.cuda.Stream = lambda *a, **k: Stream()
    torch.cuda.Event = Event
    torch.cuda.current_stream = lambda *a, **k: main
    torch.cuda.stream = lambda st: contextlib.nullcontext()
    torch.cuda.is_available = lambda: False
    n_dec = [0]

    class Batch:
        def __init__(self, k):
            self.k = k

        def resized(self, h, w, dtype=None):
            return torch.full((4, 1), float(self.k))

    def decode(staged):
        n_dec[0] += 1
        return Batch(n_dec[0])

    pipeline.entropy_decode = lambda jpegs, pinned=False: ("staged",)
    pipeline.decode_entropy = decode
    ops.binary_score = lambda p, out=None: p[:, 0].float() if out is None else out.copy_(p[:, 0].float())
    ops.ensemble_mean = lambda full: full.mean(0)
    real_gather = ensemble.gather_plan_scores
    ensemble.gather_plan_scores = lambda plan, r, n, local, d, dev: real_gather(plan, r, n, local, d, torch.device("cpu"))

    class Model:
        def __init__(self, g):
            self.g = g

        def predict(self, x):
            return x * self.g

    class Spec:
        def __init__(self, name):
            self.name, self.input_hw = name, 224

    class Reg:
        gmac_per_image = 1.0

    names = [f"m{i}" for i in range(len(GAINS))]
    workloads.zoo.MEMBERS = {n: Reg() for n in names}
    workloads.MEMBER_MS_256 = dict(zip(names, [16.8, 9.3, 8.0, 6.5, 4.9]))
    models = [(Spec(n), Model(g)) for n, g in zip(names, GAINS)]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)
    dist.all_gather_into_tensor = counting
    wl = workloads.Workload("fake", names, batch=4, rank=rank, world=world, shard=mode, jpegs=[b"x"] * 4, models=models)
    per_step = len(wl.plan.units[rank])                     # image-shards this rank decodes per step
    out = [wl.step(dist).clone()]                           # joined
    out.append(wl.step(dist).clone())
    first = wl.step(dist, pipelined=True)                   # pipelined: nothing yet
    out.append(wl.step(dist, pipelined=True).clone())
    out.append(wl.flush(dist).clone())
    q.put((rank, [o.tolist() for o in out], first is None, len(calls), per_step, wl.plan.describe() if hasattr(wl.plan, "describe") else ""))
    wl.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["images", "members", "hybrid"])
def test_workload_two_ranks_gloo(mode):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() * 13 + len(mode) * 7) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, out, first_none, ncalls, per_step, desc = q.get(timeout=180)
        res[r] = (out, per_step)
        assert first_none
        assert ncalls == 4, f"rank {r}: {ncalls} all-gathers for 4 scored steps ({desc})"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean_gain = sum(GAINS) / len(GAINS)
    for r in range(world):
        out, per_step = res[r]
        assert [len(o) for o in out] == [8] * 4                        # 2 shards x 4 images, identical on both ranks
        assert out == res[0][0]
    # every (member, shard) unit was scored exactly once per step: the ensemble mean of shard s is mean(gains) x the value of the batch
    # that shard's owner(s) decoded; with one decode per shard and step the batch counter of a rank's k-th decode is k
    out, per_step0 = res[0]
    for step, scores in enumerate(out):
        for s in range(world):
            vals = set(round(v / mean_gain, 6) for v in scores[4 * s:4 * s + 4])
            if mode == "images":
                assert vals == {float(step + 1)}, (mode, step, s, vals)     # each rank decodes one shard per step
            elif mode == "members":
                assert vals == {float(2 * step + 1 + s)}, (mode, step, s, vals)   # every rank decodes both shards, in shard order
            else:
                assert len(vals) == 1 and min(vals) > 0, (mode, step, s, vals)    # hybrid: owners differ per unit, one value per shard
