"""The fork / join contract of ensemble.MemberStreams.predict_all on fake streams (no GPU): where after_fork runs, what defer_join
returns, that every member is predicted exactly once on the stream its group was packed onto, and that the launching stream waits
for every side stream - the scheduling the bench step and the CLI loop rely on (input prefetch between fork and join)."""
import contextlib

import pytest
import torch

import vipcup_amd  # noqa: F401
from vipcup_amd import ensemble


class FakeEvent:
    def __init__(self, log, enable_timing=False):
        self.log, self.stream = log, None

    def record(self, stream=None):
        self.stream = stream if stream is not None else FakeCuda.current
        self.log.append(("record", self.stream.name))

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return FakeCuda.costs.pop(0)


class FakeStream:
    n = 0

    def __init__(self, log, name=None):
        self.log = log
        if name is None:
            FakeStream.n += 1
            name = f"side{FakeStream.n}"
        self.name = name

    def wait_event(self, ev):
        self.log.append(("wait", self.name, ev.stream.name))


class FakeCuda:
    current = None
    costs = []


class Member:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def predict(self, x):
        self.log.append(("predict", self.name, FakeCuda.current.name))
        return (self.name, x)


class Spec:
    def __init__(self, name, hw):
        self.name, self.input_hw = name, hw


@pytest.fixture
def fake_cuda(monkeypatch):
    log = []
    FakeStream.n = 0
    main = FakeStream(log, "main")
    FakeCuda.current = main

    @contextlib.contextmanager
    def stream_ctx(st):
        old, FakeCuda.current = FakeCuda.current, st
        try:
            yield
        finally:
            FakeCuda.current = old

    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: FakeStream(log))
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: FakeEvent(log, *a, **k))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: FakeCuda.current)
    monkeypatch.setattr(torch.cuda, "stream", stream_ctx)
    return log


def test_fork_after_fork_join_order(fake_cuda):
    log = fake_cuda
    members = [(Spec(n, hw), Member(n, log)) for n, hw in [("a", 224), ("b", 200), ("c", 224), ("d", 384), ("e", 200)]]
    inputs = {224: "x224", 200: "x200", 384: "x384"}
    ms = ensemble.MemberStreams(3)
    FakeCuda.costs = [9.0, 5.0, 4.0, 3.0, 1.0]                      # the one-off serial timing pass
    hook = []
    first = ms.predict_all(members, inputs, after_fork=lambda: hook.append(len(log)))
    assert [p[0] for p in first] == ["a", "b", "c", "d", "e"] and len(hook) == 1
    assert all(e[2] == "main" for e in log if e[0] == "predict")    # calibration pass: serial, on the launching stream
    assert ms.cost_ms == {"a": 9.0, "b": 5.0, "c": 4.0, "d": 3.0, "e": 1.0}

    del log[:]
    out = ms.predict_all(members, inputs, after_fork=lambda: log.append(("after_fork",)))
    assert out == [(n, inputs[hw]) for n, hw in [("a", 224), ("b", 200), ("c", 224), ("d", 384), ("e", 200)]]
    on = {e[1]: e[2] for e in log if e[0] == "predict"}
    assert sorted(on) == ["a", "b", "c", "d", "e"] and "main" not in on.values()
    # longest-first packing of 9, 5, 4, 3, 1 onto three streams: a | b | c, then d (3) joins the lightest (c: 4 -> 7), e (1) joins b (5 -> 6)
    groups = {}
    for m, st in on.items():
        groups.setdefault(st, []).append(m)
    assert sorted(sorted(g) for g in groups.values()) == [["a"], ["b", "e"], ["c", "d"]]
    i_fork = log.index(("after_fork",))
    assert all(log.index(e) < i_fork for e in log if e[0] == "predict")                      # everything is enqueued before the hook
    joins = [e for e in log if e[0] == "wait" and e[1] == "main"]
    assert len(joins) == 3 and all(log.index(e) > i_fork for e in joins)                       # the launching stream joins AFTER the hook
    assert {e[2] for e in joins} == set(groups)                                                # ... every side stream
    assert [e for e in log if e[0] == "wait" and e[1] != "main"] == [("wait", s, "main") for s in ("side1", "side2", "side3")]

    del log[:]
    out2, events = ms.predict_all(members, inputs, defer_join=True)
    assert out2 == out and len(events) == 3
    assert not [e for e in log if e[0] == "wait" and e[1] == "main"]                           # nobody waited yet
    ensemble.MemberStreams.join(events)
    assert len([e for e in log if e[0] == "wait" and e[1] == "main"]) == 3


def test_single_stream_and_single_member(fake_cuda):
    log = fake_cuda
    members = [(Spec("a", 224), Member("a", log)), (Spec("b", 224), Member("b", log))]
    called = []
    ms = ensemble.MemberStreams(1)
    out, ev = ms.predict_all(members, {224: "x"}, after_fork=lambda: called.append(1), defer_join=True)
    assert [o[0] for o in out] == ["a", "b"] and ev == [] and called == [1]
    assert not [e for e in log if e[0] in ("wait", "record")]
    ms3 = ensemble.MemberStreams(3)
    out = ms3.predict_all(members[:1], {224: "x"}, after_fork=lambda: called.append(2))
    assert out == [("a", "x")] and called == [1, 2]


def test_workload_pipelined_steps_and_input_prefetch(fake_cuda, monkeypatch):
    """workloads.Workload on fakes: a pipelined step returns the previous step's scores (flush() the last), the next batch's device
    decode is enqueued between the fork and the join of the current one, and every batch is decoded exactly once."""
    import numpy as np
    from vipcup_amd import ops, pipeline, workloads
    log = fake_cuda
    decoded = []

    class Batch:
        def __init__(self, k):
            self.k = k

        def resized(self, h, w, dtype=None):
            return torch.full((4, 1), float(self.k))

    def fake_entropy(jpegs, pinned=False):
        return ("staged", len(decoded))

    def fake_decode(staged):
        decoded.append(len(log))
        log.append(("decode", len(decoded)))
        return Batch(len(decoded))

    monkeypatch.setattr(pipeline, "entropy_decode", fake_entropy)
    monkeypatch.setattr(pipeline, "decode_entropy", fake_decode)
    def fake_score(p, out=None):
        v = p[:, 0].float()
        return v if out is None else out.copy_(v)
    monkeypatch.setattr(ops, "binary_score", fake_score)
    monkeypatch.setattr(ops, "ensemble_mean", lambda full: full.mean(0))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)

    class Model:
        def __init__(self, name, gain):
            self.name, self.gain = name, gain

        def predict(self, x):
            log.append(("predict", self.name, FakeCuda.current.name))
            return x * self.gain

    models = [(Spec("a", 224), Model("a", 1.0)), (Spec("b", 200), Model("b", 3.0))]
    class Reg:
        gmac_per_image = 1.0
    monkeypatch.setattr(workloads.zoo, "MEMBERS", {"a": Reg(), "b": Reg()})
    monkeypatch.setattr(workloads, "MEMBER_MS_256", {"a": 2.0, "b": 1.0})
    FakeCuda.costs = [2.0, 1.0]
    wl = workloads.Workload("fake", ["a", "b"], batch=4, rank=0, world=1, jpegs=[b"x"] * 4, models=models)
    s1 = wl.step()                                                  # joined step (and the one-off calibration pass): batch 1
    assert torch.equal(s1, torch.full((4,), 2.0))                   # mean of 1 * 1 and 1 * 3
    assert [e for e in log if e[0] == "decode"] == [("decode", 1), ("decode", 2)]      # batch 2 was prefetched during step 1
    del log[:]
    assert wl.step(pipelined=True) is None                          # batch 2 forked, nothing to return yet
    i_dec = log.index(("decode", 3))
    assert all(log.index(e) < i_dec for e in log if e[0] == "predict")                  # the prefetch sits behind the fork ...
    assert not [e for e in log if e[0] == "wait" and e[1] == "main"]                    # ... and nobody has joined
    s2 = wl.step(pipelined=True)                                    # forks batch 3, joins and scores batch 2
    assert torch.equal(s2, torch.full((4,), 4.0))
    s3 = wl.flush()
    assert torch.equal(s3, torch.full((4,), 6.0)) and torch.equal(wl.flush(), s3)
    assert len(decoded) == 4                                        # batches 1-3 scored, batch 4 prefetched, none decoded twice
    wl.close()
