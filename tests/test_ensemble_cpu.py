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


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def jpegs_for(lo, hi):
        return [bytes([i % 251]) for i in range(lo, hi)]

    def scorer(raws, members):  # stand-in for the GPU path: score = f(member, image id)
        ids = torch.tensor([r[0] for r in raws], dtype=torch.float32)
        return torch.stack([(ids * (m + 1) % 97) / 97.0 for m in range(len(members))], 0)

    out = ensemble.score_files(jpegs_for, n, [(None, None)] * 3, batch_size=16, rank=rank, world=world, dist=dist,
                               scorer=scorer)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 2])
def test_two_rank_gloo_exchange(n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = np.arange(n, dtype=np.float32) % 251
    want = np.stack([(ids * (m + 1) % 97) / 97.0 for m in range(3)], 0)
    for r in (0, 1):
        assert res[r].shape == (3, n)
        assert np.allclose(res[r], want)
