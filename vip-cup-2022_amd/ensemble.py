"""Ensemble scoring — the MI355X counterpart of ``predict_soln`` (main.py:58-149).

Reference flow per model: rebuild the dataset (re-decoding every JPEG), ``model.predict`` in batches of 128,
mean over TTA (:111), multi-class -> binary ``1 - p[:,0]`` (:113-114), mean over folds (:121); then across
models: concat, ``groupby(filename).mean()`` (:142-143), ``(mean > thr) * 1.0`` (:144), CSV (:145).

Here: each batch of files is decoded ONCE, every member consumes the resident pixels at its own resolution,
and with N > 1 processes the IMAGES are sharded across ranks (every rank holds all members — 175 M parameters
are nothing next to 288 GB) with one all-gather of the per-image scores at the end (the only exchange step;
SURVEY.md §8e).
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


def all_gather_scores(local: torch.Tensor, counts: List[int], dist=None) -> torch.Tensor:
    """The exchange step: every rank contributes ``[M, n_local]`` scores; returns ``[M, sum(counts)]`` in rank
    order on every rank.  Shards are padded to the largest count so ONE all_gather (RCCL over xGMI on GPUs,
    gloo in the CPU tests) moves everything."""
    if dist is None or len(counts) == 1:
        return local
    M = local.shape[0]
    width = max(counts)
    pad = torch.zeros((M, width), dtype=local.dtype, device=local.device)
    pad[:, :local.shape[1]] = local
    out = torch.empty((len(counts) * M, width), dtype=local.dtype, device=local.device)  # concatenated along dim 0
    dist.all_gather_into_tensor(out, pad)
    out = out.view(len(counts), M, width)
    return torch.cat([out[r, :, :c] for r, c in enumerate(counts)], dim=1)


def tta_flags(n_images: int, tta: int, seed: int = 0) -> np.ndarray:
    """The random draws of ``apply_augment`` (dataset/augment.py:153-182 with RandomFlip :115-120, RandomGray
    :142-146) for every (pass, image): bool ``[tta, n_images, 3]`` = (hflip, vflip, gray).  With probability 0.2 an
    image is left alone (``random_float() > 0.80``), otherwise hflip / vflip with p = 0.5 each and gray with p = 0.3.
    TensorFlow's RNG stream cannot be reproduced, so the draws come from a seeded numpy generator; they are a function
    of (seed, pass, image index) only, so the scores do not depend on batch size or on how images are sharded."""
    out = np.zeros((tta, n_images, 3), dtype=bool)
    for t in range(tta):
        u = np.random.default_rng([int(seed), t]).random((n_images, 4))
        on = ~(u[:, 0] > 0.80)
        out[t, :, 0] = on & (u[:, 1] < 0.5)
        out[t, :, 1] = on & (u[:, 2] < 0.5)
        out[t, :, 2] = on & (u[:, 3] < 0.3)
    return out


def score_files(jpegs_for: Callable[[int, int], List[bytes]], n_images: int, members: List[Tuple[object, object]],
                batch_size: int = REF_BATCH, rank: int = 0, world: int = 1, dist=None,
                scorer: Optional[Callable] = None, tta: int = 1, tta_seed: int = 0) -> np.ndarray:
    """Score images [0, n_images) with every member; returns ``[M, n_images]`` fp32 probabilities on every rank.

    ``jpegs_for(lo, hi)`` returns the JPEG byte strings of images lo..hi-1 (read lazily, per batch).
    ``members`` = [(spec, model)] with ``spec.input_hw`` and ``model.predict(x) -> [n, C]``.
    ``tta`` > 1: every image is scored ``tta`` times under ``apply_augment`` draws and the predictions are averaged
    (main.py:92,109-111, ``CFG.agg = 'mean'``); the JPEGs are still decoded once.
    ``scorer(raws, members[, flags]) -> [M, n]`` replaces the GPU path in the CPU (gloo) tests."""
    lo, hi = shard_bounds(n_images, rank, world)
    counts = [shard_bounds(n_images, r, world)[1] - shard_bounds(n_images, r, world)[0] for r in range(world)]
    flags_all = tta_flags(n_images, tta, tta_seed) if tta > 1 else None
    chunks = []
    starts = list(range(lo, hi, batch_size))

    def host_stage(b0):
        """file read + Huffman decode of one batch (C++ threads, the GIL is released inside the ctypes call)"""
        raws = jpegs_for(b0, min(b0 + batch_size, hi))
        if scorer is not None:
            return raws
        from . import pipeline
        return pipeline.entropy_decode(raws, pinned=True)

    # one batch of read-ahead: the host stage of batch i+1 runs while the GPU scores batch i (the reference gets the
    # same overlap from tf.data's prefetch, dataset/dataset.py:101)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(host_stage, starts[0]) if starts else None
        for i, b0 in enumerate(starts):
            b1 = min(b0 + batch_size, hi)
            staged = nxt.result()
            nxt = pool.submit(host_stage, starts[i + 1]) if i + 1 < len(starts) else None
            fl = None if flags_all is None else flags_all[:, b0:b1]
            if scorer is not None:
                chunks.append(scorer(staged, members) if fl is None else scorer(staged, members, fl))
            else:
                chunks.append(_score_batch(staged, members, fl))
    M = len(members)
    if chunks:
        local = torch.cat(chunks, dim=1)
    else:
        dev = "cuda" if (scorer is None and torch.cuda.is_available()) else "cpu"
        local = torch.zeros((M, 0), dtype=torch.float32, device=dev)
    full = all_gather_scores(local, counts, dist)
    return full.detach().float().cpu().numpy()


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
        self.assign: Optional[List[List[int]]] = None

    def _calibrate(self, members, inputs):
        """one serial, timed pass; its predictions ARE the first batch's result (nothing is computed twice)"""
        cost, out = [], []
        for i, (spec, model) in enumerate(members):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out.append(model.predict(inputs[spec.input_hw]))
            e1.record()
            e1.synchronize()
            cost.append(e0.elapsed_time(e1))
        order = sorted(range(len(members)), key=lambda i: -cost[i])
        load = [0.0] * self.n
        assign: List[List[int]] = [[] for _ in range(self.n)]
        for i in order:                                   # longest-processing-time-first packing
            j = min(range(self.n), key=lambda k: load[k])
            assign[j].append(i)
            load[j] += cost[i]
        self.assign = assign
        return out

    def predict_all(self, members, inputs) -> list:
        """inputs: {input_hw: tensor} produced on the current stream.  Returns member.predict() per member."""
        if self.n <= 1 or len(members) <= 1:
            return [model.predict(inputs[spec.input_hw]) for spec, model in members]
        if self.assign is None or sum(len(a) for a in self.assign) != len(members):
            return self._calibrate(members, inputs)
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        out = [None] * len(members)
        for st, idxs in zip(self.streams, self.assign):
            if not idxs:
                continue
            st.wait_event(ready)
            with torch.cuda.stream(st):
                for i in idxs:
                    spec, model = members[i]
                    out[i] = model.predict(inputs[spec.input_hw])
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        return out


def default_streams() -> int:
    import os
    return int(os.environ.get("VIP_STREAMS", "3"))


_MEMBER_STREAMS: Optional[MemberStreams] = None


def _score_batch(staged, members, flags: Optional[np.ndarray] = None) -> torch.Tensor:
    """``staged`` = ``pipeline.entropy_decode(raws)`` (or the raw JPEG byte strings).  Decode once -> per member: resize
    to its resolution, predict, multi->binary.  Returns [M, n] (device).
    ``flags`` bool [tta, n, 3] (hflip, vflip, gray): one pass per row over augmented copies of the resized batch, mean
    over passes (the mean commutes with the multi->binary map 1 - p0)."""
    from . import pipeline
    batch = pipeline.decode_entropy(staged) if isinstance(staged, tuple) else pipeline.decode_jpegs(staged)
    global _MEMBER_STREAMS
    if _MEMBER_STREAMS is None:
        _MEMBER_STREAMS = MemberStreams(default_streams())
    cache: Dict[int, torch.Tensor] = {}
    for spec, _ in members:
        hw = spec.input_hw
        if hw not in cache:
            cache[hw] = batch.resized(hw, hw)
    def one_pass(inputs):
        rows = []
        for p in _MEMBER_STREAMS.predict_all(members, inputs):   # [n, C] fp32 each
            p = (1.0 - p[:, 0]) if p.shape[1] > 1 else p[:, 0]   # main.py:113-114
            rows.append(p.float())
        return torch.stack(rows, 0)

    if flags is None:
        return one_pass(cache)
    acc = None
    for fl in flags:                                             # augment AFTER decode+resize, as dataset.py:88-99 maps it
        aug = {hw: pipeline.apply_augment(x, fl[:, 0], fl[:, 1], fl[:, 2]) for hw, x in cache.items()}
        s = one_pass(aug)
        acc = s if acc is None else acc + s
    return acc / float(len(flags))
