"""CPU: ensemble aggregation semantics (main.py:111-145) and the world_size-2 exchange step over gloo."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_aggregate_matches_reference_semantics():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble
    names = ["b.jpg", "a.jpg", "c.jpg", "a.jpg"]                     # unsorted + duplicate
    per_model = np.array([[0.2, 0.9, 0.487, 0.5], [0.4, 0.7, 0.487, 0.1]], np.float32)
    uniq, score, dec = ensemble.aggregate(names, per_model)
    assert uniq == ["a.jpg", "b.jpg", "c.jpg"]                       # groupby sorts (main.py:143)
    # pandas semantics: concat of per-model frames then groupby-mean == mean over (model, duplicate) rows
    assert np.allclose(score, [np.mean([0.9, 0.7, 0.5, 0.1]), 0.3, 0.487])
    assert dec.tolist() == [1.0, 0.0, 0.0]                           # strict > 0.487 (main.py:144)
    assert np.allclose(ensemble.to_binary(np.array([[0.8, 0.2], [0.1, 0.9]])), [[0.2], [0.9]])  # main.py:113-114


def test_shard_bounds_cover_everything():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble
    for n in (0, 1, 7, 5000):
        for world in (1, 2, 3, 8):
            b = [ensemble.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_tta_flags_follow_apply_augment_and_do_not_depend_on_sharding():
    """dataset/augment.py:153-182: 20 % of the draws leave the image alone; otherwise hflip .5, vflip .5, gray .3"""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble
    f = ensemble.tta_flags(20000, 3, seed=5)
    assert f.shape == (3, 20000, 3) and f.dtype == bool
    assert np.allclose(f.reshape(-1, 3).mean(0), [0.8 * 0.5, 0.8 * 0.5, 0.8 * 0.3], atol=0.01)
    assert abs((~f.any(-1)).mean() - (0.2 + 0.8 * 0.25 * 0.7)) < 0.01      # untouched: gate off, or all three draws miss
    assert np.array_equal(f, ensemble.tta_flags(20000, 3, seed=5)) and not np.array_equal(f[0], f[1])
    assert not np.array_equal(f, ensemble.tta_flags(20000, 3, seed=6))

    # score_files: mean over passes, the same numbers whatever the batch size / shard
    def jpegs_for(lo, hi):
        return [bytes([i % 251]) for i in range(lo, hi)]

    def scorer(raws, members, flags=None):
        ids = torch.tensor([r[0] for r in raws], dtype=torch.float32)
        base = torch.stack([(ids * (m + 1) % 97) / 97.0 for m in range(len(members))], 0)
        if flags is None:
            return base
        w = torch.tensor(flags.astype(np.float32)) @ torch.tensor([0.1, 0.01, 0.001])      # [tta, n]
        return base + w.mean(0)[None, :]

    n = 53
    ref = ensemble.score_files(jpegs_for, n, [(None, None)] * 2, batch_size=64, scorer=scorer, tta=4, tta_seed=9)
    fl = ensemble.tta_flags(n, 4, 9).astype(np.float32) @ np.array([0.1, 0.01, 0.001], np.float32)
    plain = ensemble.score_files(jpegs_for, n, [(None, None)] * 2, batch_size=64, scorer=scorer)
    assert np.allclose(ref, plain + fl.mean(0)[None, :], atol=1e-6)
    small = ensemble.score_files(jpegs_for, n, [(None, None)] * 2, batch_size=7, scorer=scorer, tta=4, tta_seed=9)
    assert np.allclose(ref, small, atol=1e-6)
    lo, hi = ensemble.shard_bounds(n, 1, 2)
    part = ensemble.score_files(lambda a, b: jpegs_for(a, b), n, [(None, None)] * 2, batch_size=16, rank=1, world=2,
                                dist=None, scorer=scorer, tta=4, tta_seed=9)      # no exchange: only rank 1's shard is filled in
    assert np.allclose(part[:, lo:hi], ref[:, lo:hi], atol=1e-6) and not part[:, :lo].any()


def _id_scorer(raws, members, flags=None):
    """stand-in for the GPU path: score = f(member id, image id); members = [(member id, None)]"""
    ids = torch.tensor([r[0] for r in raws], dtype=torch.float32)
    return torch.stack([(ids * (mid + 1) % 97) / 97.0 for mid, _ in members], 0)


def _worker(rank, world, port, n, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)
    dist.all_gather_into_tensor = counting

    def jpegs_for(lo, hi):
        return [bytes([i % 251]) for i in range(lo, hi)]

    out = ensemble.score_files(jpegs_for, n, [(m, None) for m in range(5)], batch_size=16, rank=rank, world=world, dist=dist,
                               scorer=_id_scorer, shard=mode, costs=[16.8, 9.3, 8.0, 6.5, 4.9])
    q.put((rank, out, len(calls)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,mode", [(2, 37, "images"), (2, 2, "images"), (2, 37, "members"), (2, 37, "hybrid"),
                                          (4, 37, "images"), (4, 37, "members"), (4, 37, "hybrid"), (4, 3, "hybrid")])
def test_multi_rank_gloo_exchange(world, n, mode):
    """world_size 2 and 4 over gloo, every ShardPlan mode: scores identical to the single-process result, ONE all-gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 31 + n + len(mode) * 3) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, out, ncalls = q.get(timeout=120)
        res[r] = out
        assert ncalls == 1, f"rank {r}: {ncalls} all-gathers on the data path"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = np.arange(n, dtype=np.float32) % 251
    want = np.stack([(ids * (m + 1) % 97) / 97.0 for m in range(5)], 0)
    for r in range(world):
        assert res[r].shape == (5, n)
        assert np.allclose(res[r], want), (mode, r)


def test_shard_plans_partition_the_work():
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble
    costs = [16.8, 9.3, 8.0, 6.5, 6.2, 5.6, 4.9, 7.0]          # ms per 256 images, the eight config-5 members
    for world in (1, 2, 4, 8):
        for mode in ensemble.SHARD_MODES:
            plan = ensemble.ShardPlan(mode, len(costs), world, costs)
            owned = sorted((m, s) for u in plan.units for s, ms in u.items() for m in ms)
            assert owned == sorted((m, s) for m in range(len(costs)) for s in range(world)), (mode, world)
            assert ensemble.ShardPlan(mode, len(costs), world, costs).units == plan.units      # deterministic
    members8 = ensemble.ShardPlan("members", 8, 8, costs)
    assert all(list(u.values())[0] == [r] and sorted(u) == list(range(8)) for r, u in enumerate(members8.units))
    # LPT: the critical path of hybrid is within one unit of the mean load; member-parallel is capped by the slowest member
    hyb = ensemble.ShardPlan("hybrid", 8, 8, costs)
    mean = sum(costs)
    assert max(hyb.load) <= mean + min(costs) and max(hyb.load) < 8 * max(costs)
    with pytest.raises(ValueError):
        ensemble.ShardPlan("rows", 2, 2)
