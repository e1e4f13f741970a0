"""Ensemble scoring — the MI355X counterpart of ``predict_soln`` (main.py:58-149).

Reference flow per model: rebuild the dataset (re-decoding every JPEG), ``model.predict`` in batches of 128,
mean over TTA (:111), multi-class -> binary ``1 - p[:,0]`` (:113-114), mean over folds (:121); then across
models: concat, ``groupby(filename).mean()`` (:142-143), ``(mean > thr) * 1.0`` (:144), CSV (:145).

Here: each batch of files is decoded ONCE, every member consumes the resident pixels at its own resolution.
With N > 1 processes the work is a grid of (member, image-shard) units dealt to the ranks by a ``ShardPlan``:
``images`` (every rank holds all members and scores its own image shard - what MirroredStrategy does, utils/device.py:7),
``members`` (rank r owns members m = r mod world and sees every image - the one-model-per-GPU split of BASELINE.json) or
``hybrid`` (longest-processing-time-first packing of the units by measured ms/image).  In every mode the only exchange step
is ONE all-gather of the ranks' score payloads at the end.
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

THR = 0.487          # main.py:225
REF_BATCH = 128      # 8 * NAME2BS.get(name, 16) for every shipped member (main.py:43-56,85)
NAME2BS = {          # main.py:43-56: per-replica batch of the larger members of earlier ensembles (x 8 replicas, :85)
    "convnext_large_384_in22ft1k-200x200": 16, "convnext_large_in22ft1k-200x200": 16, "convnext_base_384_in22ft1k-200x200": 32,
    "HorNetBase-200x200": 32, "EfficientNetV2M-200x200": 64, "convnext_base_in22k-200x200": 32, "ECA_NFNetL2-200x200": 32,
    "GCViTBase-224x224": 48, "ResNest200-200x200": 64, "EfficientNetV2L-200x200": 32, "ResNetRS200-200x200": 32,
    "ResNet200D-200x200": 32,
}


def ref_batch(ckpt_name: str) -> int:
    """the reference's batch size for a checkpoint directory name (main.py:85)"""
    return 8 * NAME2BS.get(ckpt_name, 16)


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced image shard of rank ``rank``: sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


SHARD_MODES = ("images", "members", "hybrid")


class ShardPlan:
    """Who scores what: ``units[r]`` = {image-shard s: [member index, ...]} for rank r; image-shard s = ``shard_bounds(n, s, world)``.
    Every (member, shard) pair is owned by exactly one rank.  The plan is a pure function of (mode, n_members, world, costs), so
    every rank derives the same one without talking to the others."""

    def __init__(self, mode: str, n_members: int, world: int, costs: Optional[Sequence[float]] = None):
        if mode not in SHARD_MODES:
            raise ValueError(f"shard mode {mode!r}: expected one of {SHARD_MODES}")
        self.mode, self.n_members, self.world = mode, n_members, world
        units: List[Dict[int, List[int]]] = [dict() for _ in range(world)]
        if mode == "images" or world == 1:
            for r in range(world):
                units[r][r] = list(range(n_members))
        elif mode == "members":
            for r in range(world):
                mine = [m for m in range(n_members) if m % world == r]
                if mine:
                    for s in range(world):
                        units[r][s] = list(mine)
        else:       # hybrid: LPT over the n_members x world units; a unit costs the member's ms/image (shards are equal-sized)
            c = [1.0] * n_members if costs is None else [float(v) for v in costs]
            assert len(c) == n_members
            order = sorted(((c[m], m, s) for m in range(n_members) for s in range(world)), key=lambda t: (-t[0], t[1], t[2]))
            load = [0.0] * world
            for cost, m, s in order:
                # least-loaded rank; ties go to the rank that already decodes shard s (no extra decode), then to the lowest rank
                r = min(range(world), key=lambda k: (round(load[k], 9), 0 if s in units[k] else 1, k))
                units[r].setdefault(s, []).append(m)
                load[r] += cost
            for r in range(world):
                for s in units[r]:
                    units[r][s].sort()
            self.load = load
        self.units = units

    def payload_len(self, rank: int, n_images: int) -> int:
        return sum(len(ms) * (shard_bounds(n_images, s, self.world)[1] - shard_bounds(n_images, s, self.world)[0])
                   for s, ms in self.units[rank].items())

    def describe(self) -> str:
        return "; ".join(f"rank {r}: " + ", ".join(f"shard {s} x members {ms}" for s, ms in sorted(u.items()))
                         for r, u in enumerate(self.units))


def plan_payload(plan: ShardPlan, rank: int, n_images: int, device):
    """This rank's exchange payload and where each of its units' scores go in it: ``(mine [width] fp32, {(s, m): 1-D view of mine})``.
    Producers write their scores STRAIGHT into the views (``ops.binary_score(p, out=view)``) - no staging copies."""
    world = plan.world
    width = max(max(plan.payload_len(r, n_images) for r in range(world)), 1)
    mine = torch.zeros((width,), dtype=torch.float32, device=device)
    views, off = {}, 0
    for s in sorted(plan.units[rank]):
        lo, hi = shard_bounds(n_images, s, world)
        for m in plan.units[rank][s]:
            views[(s, m)] = mine[off:off + (hi - lo)]
            off += hi - lo
    assert off == plan.payload_len(rank, n_images)
    return mine, views


def exchange_payload(plan: ShardPlan, rank: int, n_images: int, mine: torch.Tensor, dist=None) -> torch.Tensor:
    """The exchange step: ONE ``all_gather_into_tensor`` of the ranks' payloads (RCCL over xGMI on GPUs, gloo in the CPU tests);
    returns ``[n_members, n_images]`` on every rank.  One rank with the ``images`` plan: the payload already IS that matrix."""
    world, width = plan.world, mine.numel()
    gathered = dist is not None and world > 1
    if not gathered and world == 1 and width == plan.n_members * n_images and n_images > 0:
        return mine.view(plan.n_members, n_images)
    if gathered:
        allp = torch.empty((world * width,), dtype=torch.float32, device=mine.device)     # flat: every backend accepts this form
        dist.all_gather_into_tensor(allp, mine)
        allp = allp.view(world, width)
    else:
        allp = mine.view(1, width)
    full = torch.zeros((plan.n_members, n_images), dtype=torch.float32, device=mine.device)
    for r in (range(world) if gathered else [rank]):     # without an exchange only this rank's units are known
        off = 0
        for s in sorted(plan.units[r]):
            lo, hi = shard_bounds(n_images, s, world)
            for m in plan.units[r][s]:
                full[m, lo:hi] = allp[r if gathered else 0, off:off + (hi - lo)]
                off += hi - lo
    return full


def gather_plan_scores(plan: ShardPlan, rank: int, n_images: int, local: Dict[Tuple[int, int], torch.Tensor], dist=None,
                       device=None) -> torch.Tensor:
    """``plan_payload`` + ``exchange_payload`` for callers that hold their units' scores as separate tensors: ``local[(s, m)]`` =
    scores of member m on image-shard s (1-D, this rank's units).  Each rank's payload (its units back to back, in (shard, member)
    order, padded to the longest payload) goes through ONE ``all_gather_into_tensor``; returns ``[n_members, n_images]`` on every rank."""
    if device is None:
        device = next(iter(local.values())).device if local else torch.device("cpu")
    mine, views = plan_payload(plan, rank, n_images, device)
    for key, view in views.items():
        view.copy_(local[key].reshape(-1).to(torch.float32))
    return exchange_payload(plan, rank, n_images, mine, dist)


def to_binary(pred: np.ndarray) -> np.ndarray:
    """main.py:113-114: a multi-class head reports P(real) in column 0 -> P(synthetic) = 1 - p[:,0]"""
    pred = np.asarray(pred, dtype=np.float32)
    if pred.ndim == 1:
        pred = pred[:, None]
    return 1.0 - pred[:, 0:1] if pred.shape[1] > 1 else pred


def aggregate(filenames: Sequence[str], per_model: np.ndarray, thr: float = THR):
    """per_model ``[M, N]`` probabilities -> (sorted unique filenames, mean score, decision).
    Mirrors concat + groupby('filename').mean() (rows sorted by filename, duplicates averaged) and the strict
    ``> thr`` of main.py:142-144.  (The reference's own frame has a string-typed 'logit' column and fails on
    current pandas; the evident intent — the arithmetic mean — is what is computed.)"""
    names = np.asarray(filenames)
    uniq, inv = np.unique(names, return_inverse=True)
    mean_per_image = per_model.astype(np.float64).mean(axis=0)            # mean over models per row
    sums = np.zeros(len(uniq), np.float64)
    cnts = np.zeros(len(uniq), np.float64)
    np.add.at(sums, inv, mean_per_image)
    np.add.at(cnts, inv, 1.0)
    score = (sums / cnts).astype(np.float32)
    return uniq.tolist(), score, (score > thr).astype(np.float32)


def tta_flags(n_images: int, tta: int, seed: int = 0) -> np.ndarray:
    """The random draws of ``apply_augment`` (dataset/augment.py:153-182 with RandomFlip :115-120, RandomGray
    :142-146) for every (pass, image): bool ``[tta, n_images, 3]`` = (hflip, vflip, gray).  With probability 0.2 an
    image is left alone (``random_float() > 0.80``), otherwise hflip / vflip with p = 0.5 each and gray with p = 0.3.
    TensorFlow's RNG stream cannot be reproduced, so the draws come from a seeded numpy generator; they are a function
    of (seed, pass, image index) only, so the scores do not depend on batch size or on how images are sharded."""
    return np.stack([tta_flags_pass(n_images, t, seed) for t in range(tta)]) if tta > 0 else np.zeros((0, n_images, 3), dtype=bool)


def tta_flags_pass(n_images: int, t: int, seed: int = 0) -> np.ndarray:
    """the draws of pass ``t`` alone: bool ``[n_images, 3]`` (``tta_flags(n, T, seed)[t]`` for any T > t)"""
    out = np.zeros((n_images, 3), dtype=bool)
    u = np.random.default_rng([int(seed), int(t)]).random((n_images, 4))
    on = ~(u[:, 0] > 0.80)
    out[:, 0] = on & (u[:, 1] < 0.5)
    out[:, 1] = on & (u[:, 2] < 0.5)
    out[:, 2] = on & (u[:, 3] < 0.3)
    return out


def score_files(jpegs_for: Callable[[int, int], List[bytes]], n_images: int, members: List[Tuple[object, object]],
                batch_size: int = REF_BATCH, rank: int = 0, world: int = 1, dist=None,
                scorer: Optional[Callable] = None, tta: int = 1, tta_seed: int = 0,
                shard: str = "images", costs: Optional[Sequence[float]] = None) -> np.ndarray:
    """Score images [0, n_images) with every member; returns ``[M, n_images]`` fp32 probabilities on every rank.

    ``jpegs_for(lo, hi)`` returns the JPEG byte strings of images lo..hi-1 (read lazily, per batch).
    ``members`` = [(spec, model)] with ``spec.input_hw`` and ``model.predict(x) -> [n, C]``.
    ``tta`` > 1: every image is scored ``tta`` times under ``apply_augment`` draws and the predictions are averaged
    (main.py:92,109-111, ``CFG.agg = 'mean'``); the JPEGs are still decoded once.
    ``shard`` / ``costs``: the ShardPlan mode and, for ``hybrid``, the per-member cost (ms/image; identical on every rank).
    A rank only calls ``model.predict`` of the members its plan names, so under ``members`` / ``hybrid`` the others need not
    be resident (``members[i][1]`` may be None there).
    ``scorer(raws, members[, flags]) -> [M, n]`` replaces the GPU path in the CPU (gloo) tests."""
    plan = ShardPlan(shard, len(members), world, costs)
    flags_all = tta_flags(n_images, tta, tta_seed) if tta > 1 else None
    work = []                                   # (shard, member indices, b0, b1)
    for s in sorted(plan.units[rank]):
        lo, hi = shard_bounds(n_images, s, world)
        for b0 in range(lo, hi, batch_size):
            work.append((s, plan.units[rank][s], b0, min(b0 + batch_size, hi)))

    def host_stage(item):
        """file read + Huffman decode of one batch (C++ threads, the GIL is released inside the ctypes call)"""
        raws = jpegs_for(item[2], item[3])
        if scorer is not None:
            return raws
        from . import pipeline
        return pipeline.entropy_decode(raws, pinned=True)

    # read-ahead (the overlap the reference gets from tf.data's prefetch, dataset/dataset.py:101): while the GPU scores batch i the
    # host stage of batch i+2 runs on a worker thread, and the DEVICE half of batch i+1 (H2D of the coefficients, IDCT, colour) is
    # enqueued on the launching stream between the fork of the member streams and their join - that stream idles while the members
    # run, so the work lands under them instead of at the head of the next batch (MemberStreams.predict_all(after_fork=...))
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    chunks: Dict[Tuple[int, int], List[torch.Tensor]] = {}
    with ThreadPoolExecutor(max_workers=1) as pool:
        todo = iter(work)
        futs: deque = deque()

        def submit_next():
            item = next(todo, None)
            if item is not None:
                futs.append(pool.submit(host_stage, item))

        submit_next()
        decoded = [None]

        def after_fork():
            if scorer is None and futs and decoded[0] is None:
                from . import pipeline
                st = futs.popleft().result()
                submit_next()
                decoded[0] = pipeline.decode_entropy(st)

        for i, (s, midx, b0, b1) in enumerate(work):
            if decoded[0] is not None:
                staged, decoded[0] = decoded[0], None
            else:
                staged = futs.popleft().result()
                submit_next()
            sub = [members[m] for m in midx]
            fl = None if flags_all is None else flags_all[:, b0:b1]
            if scorer is not None:
                rows = scorer(staged, sub) if fl is None else scorer(staged, sub, fl)
            else:
                rows = _score_batch(staged, sub, fl, after_fork=after_fork)
            for j, m in enumerate(midx):
                chunks.setdefault((s, m), []).append(rows[j])
    local = {k: torch.cat(v) for k, v in chunks.items()}
    dev = None
    if not local:
        dev = torch.device("cuda" if (scorer is None and torch.cuda.is_available()) else "cpu")
    for s in plan.units[rank]:                  # empty shards (more ranks than images)
        for m in plan.units[rank][s]:
            if (s, m) not in local:
                local[(s, m)] = torch.zeros((0,), dtype=torch.float32, device=dev or next(iter(local.values())).device)
    full = gather_plan_scores(plan, rank, n_images, local, dist, dev)
    return full.detach().float().cpu().numpy()


def member_dtype(model) -> torch.dtype:
    """activation dtype a member was built for (zoo.construct): packed pairs for "strict", fp32 for "f32", fp16 otherwise"""
    from . import ops
    return ops.act_dtype(getattr(model, "precision", "fast"))


def input_key(spec, model):
    """key of a member's input tensor in the ``inputs`` dictionaries below: one resize launch per (resolution, dtype)"""
    return (spec.input_hw, member_dtype(model))


def member_inputs(batch, members) -> Dict:
    """``{input_key: tensor}``: the decoded batch resized (cast -> bicubic -> /255, dataset.py:31-38) once per distinct
    (resolution, dtype) among ``members`` = [(spec, model)] (model None = not resident on this rank: skipped)."""
    out: Dict = {}
    for spec, model in members:
        if model is None:
            continue
        k = input_key(spec, model)
        if k not in out:
            out[k] = batch.resized(spec.input_hw, spec.input_hw, dtype=k[1])
    return out


def _record_stream(t, stream):
    if isinstance(t, torch.Tensor) and t.is_cuda:
        t.record_stream(stream)


def _input_for(inputs, spec, model):
    k = input_key(spec, model)
    return inputs[k] if k in inputs else inputs[spec.input_hw]      # plain {resolution: tensor} dictionaries are accepted too


class MemberStreams:
    """Runs the (independent) ensemble members on several HIP streams.

    A member's deep layers launch grids far smaller than the chip (M = 256*7*7 pixels, SE/ECA layers with M = 256)
    and every launch pays a dispatch gap; with the members spread over a few streams the hardware queues fill
    those holes with another member's kernels.  Members are packed onto streams longest-first from a one-off
    timing of each member.  The reference scores the checkpoints one after another (main.py:199-217); the
    result is order-independent (a mean over members), so only the schedule differs."""

    def __init__(self, n_streams: int):
        self.n = max(1, int(n_streams))
        self.streams = [torch.cuda.Stream() for _ in range(self.n)] if self.n > 1 else []
        self._assign: Dict[Tuple[str, ...], List[List[int]]] = {}    # per member list (a ShardPlan may hand over sub-lists)
        self.cost_ms: Dict[str, float] = {}                              # last one-off timing per member (ms per batch)

    def _calibrate(self, members, inputs):
        """one serial, timed pass; its predictions ARE the first batch's result (nothing is computed twice)"""
        cost, out = [], []
        for i, (spec, model) in enumerate(members):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out.append(model.predict(_input_for(inputs, spec, model)))
            e1.record()
            e1.synchronize()
            cost.append(e0.elapsed_time(e1))
            self.cost_ms[spec.name] = cost[-1]
        order = sorted(range(len(members)), key=lambda i: -cost[i])
        load = [0.0] * self.n
        assign: List[List[int]] = [[] for _ in range(self.n)]
        for i in order:                                   # longest-processing-time-first packing
            j = min(range(self.n), key=lambda k: load[k])
            assign[j].append(i)
            load[j] += cost[i]
        self._assign[tuple(spec.name for spec, _ in members)] = assign
        return out

    def predict_all(self, members, inputs, after_fork=None, defer_join: bool = False):
        """inputs: {input_hw: tensor} produced on the current stream.  Returns member.predict() per member.
        ``after_fork()`` is called once the members are enqueued on their streams and BEFORE the current stream joins them: work it
        enqueues on the current stream (the next batch's H2D + IDCT) runs under the members instead of in front of the next step.
        ``defer_join``: return ``(predictions, join_events)`` without making the current stream wait - the caller joins later
        (``MemberStreams.join``), so the next batch's members can start on the streams that finish first."""
        if self.n <= 1 or len(members) <= 1:
            out = [model.predict(_input_for(inputs, spec, model)) for spec, model in members]
            if after_fork is not None:
                after_fork()
            return (out, []) if defer_join else out
        assign = self._assign.get(tuple(spec.name for spec, _ in members))
        if assign is None:
            out = self._calibrate(members, inputs)
            if after_fork is not None:
                after_fork()
            return (out, []) if defer_join else out
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        out = [None] * len(members)
        joins = []
        for st, idxs in zip(self.streams, assign):
            if not idxs:
                continue
            st.wait_event(ready)
            with torch.cuda.stream(st):
                for i in idxs:
                    spec, model = members[i]
                    x = _input_for(inputs, spec, model)
                    # allocator hygiene across streams: the input was allocated on the launching stream and is read here, the
                    # prediction is allocated here and read on the launching stream - tell the caching allocator, so that neither
                    # block can be handed out again while the other stream still has work queued on it (a caller that drops its
                    # reference early - or a deferred join - would otherwise race with the block's next owner)
                    _record_stream(x, st)
                    out[i] = model.predict(x)
                    _record_stream(out[i], main)
            done = torch.cuda.Event()
            done.record(st)
            joins.append(done)
        if after_fork is not None:
            after_fork()
        if defer_join:
            return out, joins
        self.join(joins)
        return out

    @staticmethod
    def join(joins):
        main = torch.cuda.current_stream()
        for done in joins:
            main.wait_event(done)


def measure_costs(members, raws: Sequence[bytes], dist=None, rank: int = 0) -> List[float]:
    """ms per image of every member on a sample batch (JPEG byte strings), timed on rank 0 (a serial pass, the one
    ``MemberStreams._calibrate`` makes) and broadcast so that every rank derives the SAME hybrid ShardPlan.  Set-up traffic
    (one float per member), not part of the per-image data path."""
    from . import pipeline
    costs = torch.zeros((len(members),), dtype=torch.float64, device="cuda")
    if rank == 0:
        batch = pipeline.decode_jpegs(list(raws))
        inputs = member_inputs(batch, members)
        ms = MemberStreams(2)
        ms._calibrate(members, inputs)          # warm-up: first-launch costs (module load, attribute calls)
        ms._calibrate(members, inputs)
        costs = torch.tensor([ms.cost_ms[spec.name] / max(len(raws), 1) for spec, _ in members], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.broadcast(costs, src=0)
    return [float(v) for v in costs.cpu()]


def default_streams() -> int:
    import os
    return int(os.environ.get("VIP_STREAMS", "3"))


_MEMBER_STREAMS: Optional[MemberStreams] = None


def _score_batch(staged, members, flags: Optional[np.ndarray] = None, after_fork=None) -> torch.Tensor:
    """``staged`` = ``pipeline.entropy_decode(raws)`` (or the raw JPEG byte strings, or an already decoded batch).  ``after_fork``:
    called once, between the fork and the join of the member streams (see ``MemberStreams.predict_all``).  Decode once -> per member: resize
    to its resolution, predict, multi->binary.  Returns [M, n] (device).
    ``flags`` bool [tta, n, 3] (hflip, vflip, gray): one pass per row over augmented copies of the resized batch, mean
    over passes (the mean commutes with the multi->binary map 1 - p0)."""
    from . import pipeline
    if isinstance(staged, pipeline.DecodedBatch):
        batch = staged
    else:
        batch = pipeline.decode_entropy(staged) if isinstance(staged, tuple) else pipeline.decode_jpegs(staged)
    hook = [after_fork]
    global _MEMBER_STREAMS
    if _MEMBER_STREAMS is None:
        _MEMBER_STREAMS = MemberStreams(default_streams())
    cache = member_inputs(batch, members)

    def one_pass(inputs):
        from . import ops
        preds = _MEMBER_STREAMS.predict_all(members, inputs, after_fork=hook[0])     # [n, C] fp32 each
        hook[0] = None
        rows = torch.empty((len(preds), preds[0].shape[0]), dtype=torch.float32, device=preds[0].device)
        for m, p in enumerate(preds):
            ops.binary_score(p, out=rows[m])                     # main.py:113-114
        return rows

    if flags is None:
        return one_pass(cache)
    acc = None
    for fl in flags:                                             # augment AFTER decode+resize, as dataset.py:88-99 maps it
        aug = {key: pipeline.apply_augment(x, fl[:, 0], fl[:, 1], fl[:, 2]) for key, x in cache.items()}
        s = one_pass(aug)
        acc = s if acc is None else acc + s
    return acc / float(len(flags))
