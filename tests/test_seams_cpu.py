"""CPU: the Dataset / Model seams of SURVEY.md section 8(b) carry the reference's call shapes - the loop of main.py:89-121 is
re-typed here (not imported) and run against ``pipeline.build_dataset`` / ``model.predict(dataset, steps)``."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image


class _CFG:
    pass


def _write_images(tmp_path, n):
    paths = []
    rng = np.random.default_rng(0)
    for i in range(n):
        p = tmp_path / f"im_{i:03d}.jpg"
        Image.fromarray(rng.integers(0, 256, (24, 24, 3), dtype=np.uint8)).save(p, format="JPEG", quality=90)
        paths.append(str(p))
    return paths


def _pil_decode(path):
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0


def _fake_model(classes):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline

    @pipeline.keras_predict
    class Fake:
        def predict(self, x):                       # [B, H, W, 8] -> [B, classes]: a deterministic function of each image
            m = x[..., :3].float().mean((1, 2, 3))
            return torch.stack([m * (c + 1) for c in range(classes)], 1)
    return Fake()


@pytest.mark.parametrize("n,bs,tta,classes", [(37, 16, 1, 1), (5, 16, 1, 1), (37, 16, 2, 1), (16, 8, 1, 2), (1, 128, 3, 1)])
def test_reference_predict_loop_runs_on_the_seams(tmp_path, n, bs, tta, classes):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    build_dataset = pipeline.build_dataset
    test_paths = _write_images(tmp_path, n)
    model = _fake_model(classes)
    CFG = _CFG()
    CFG.tta, CFG.batch_size, CFG.img_size, CFG.agg, CFG.seed, CFG.num_classes = tta, bs, [24, 24], "mean", 42, 1
    # ---- main.py:89-114, re-typed; only decode_fn / augment_fn / device are extra (this container has no GPU) ----
    dtest = build_dataset(
        test_paths,
        labels=None,
        augment=CFG.tta > 1,
        repeat=True,
        cache=False,
        shuffle=False,
        batch_size=CFG.batch_size,
        drop_remainder=False,
        CFG=CFG, decode_fn=_pil_decode, augment_fn=lambda b: b, device="cpu")
    pred = model.predict(dtest, steps=max(CFG.tta * len(test_paths) / CFG.batch_size, 1), verbose=1)
    assert isinstance(pred, np.ndarray) and pred.ndim == 2
    steps = int(np.ceil(max(CFG.tta * n / bs, 1)))
    assert pred.shape == (steps * bs, classes)              # repeat() pads the last batch; Keras returns every row
    pred = pred[:CFG.tta * len(test_paths), :]
    pred = getattr(np, CFG.agg)(pred.reshape((CFG.tta, len(test_paths), -1)), axis=0)
    if pred.shape[1] > 1:
        pred = 1 - pred[:, 0:1]
    # -------------------------------------------------------------------------------------------------------
    want = np.array([_pil_decode(p).astype(np.float16).astype(np.float32).mean() for p in test_paths])
    want = want[:, None] if classes == 1 else 1 - want[:, None]
    assert pred.shape == (n, 1)
    assert np.allclose(pred, want, atol=2e-3)
    assert CFG.is_train is False                            # side effect of dataset.py:73


def test_dataset_stream_semantics(tmp_path):
    import vipcup_amd  # noqa: F401
    from vipcup_amd import pipeline
    paths = _write_images(tmp_path, 5)
    CFG = _CFG()
    CFG.img_size, CFG.seed, CFG.num_classes = [24, 24], 7, 3
    mk = lambda **kw: pipeline.build_dataset(paths, CFG=CFG, decode_fn=_pil_decode, device="cpu", **kw)  # noqa: E731
    ids = lambda ds, k: [[int(round(float(x[..., 0].float().mean()) * 1e4)) for x in b] for b, _ in zip(ds, range(k))]  # noqa: E731
    key = [int(round(float(_pil_decode(p).astype(np.float16)[..., 0].astype(np.float32).mean()) * 1e4)) for p in paths]
    # no repeat: one pass, short last batch unless drop_remainder
    one = ids(mk(batch_size=2, repeat=False, shuffle=False, augment=False), 99)
    assert [len(b) for b in one] == [2, 2, 1] and sum(one, []) == key
    assert [len(b) for b in ids(mk(batch_size=2, repeat=False, shuffle=False, augment=False, drop_remainder=True), 99)] == [2, 2]
    assert len(mk(batch_size=2, repeat=False, shuffle=False)) == 3
    # repeat before batch: batches run across the epoch boundary
    rep = ids(mk(batch_size=4, repeat=True, shuffle=False, augment=False), 3)
    assert sum(rep, []) == (key * 3)[:12]
    # shuffle: a seeded permutation-like stream, every image once per pass when the buffer covers the set
    sh = sum(ids(mk(batch_size=5, repeat=False, shuffle=1024, augment=False), 9), [])
    assert sorted(sh) == sorted(key) and sh == sum(ids(mk(batch_size=5, repeat=False, shuffle=1024, augment=False), 9), [])
    # labels: (batch, one-hot float labels) like decode_with_labels
    ds = pipeline.build_dataset(paths, labels=[0, 1, 2, 1, 0], batch_size=5, repeat=False, shuffle=False, augment=False, CFG=CFG,
                                decode_fn=_pil_decode, device="cpu")
    x, y = next(iter(ds))
    assert x.shape == (5, 24, 24, 8) and y.shape == (5, 3) and y.dtype == torch.float32 and y.argmax(1).tolist() == [0, 1, 2, 1, 0]
    assert CFG.is_train is True
    # cache_dir is created like dataset.py:70-71
    pipeline.build_dataset(paths, cache=True, cache_dir=str(tmp_path / "cache"), CFG=CFG, decode_fn=_pil_decode, device="cpu")
    assert os.path.isdir(tmp_path / "cache")


def test_tta_flag_cache_survives_batches_that_span_many_passes(tmp_path, monkeypatch):
    """10 images in batches of 128 with tta > 1 - one batch covers 13 passes; the per-pass flag cache used to evict passes the
    batch still needed (KeyError 0).  Also a second walk that restarts at pass 0 with a full cache."""
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble, pipeline
    paths = _write_images(tmp_path, 10)
    CFG = _CFG()
    CFG.img_size, CFG.seed, CFG.num_classes = [24, 24], 42, 1
    seen = []

    def fake_augment(batch, flip_h, flip_v, gray):
        seen.append(np.stack([np.asarray(flip_h), np.asarray(flip_v), np.asarray(gray)], 1))
        return batch
    monkeypatch.setattr(pipeline, "apply_augment", fake_augment)
    ds = pipeline.build_dataset(paths, batch_size=128, repeat=True, shuffle=False, augment=True, CFG=CFG, decode_fn=_pil_decode, device="cpu")
    for _ in range(2):                                     # the second predict() restarts the stream at pass 0
        it = iter(ds)
        for _ in range(3):
            next(it)
    assert len(seen) >= 6 and all(s.shape == (128, 3) for s in seen[:6])
    want = np.stack([ensemble.tta_flags_pass(10, t, 42)[i] for t in range(13) for i in range(10)])[:128]
    assert np.array_equal(seen[0], want) and np.array_equal(seen[3], want)
    assert len(ds._flags) <= 14


def test_workspace_query_and_kernel_name_are_exported():
    import ctypes as C
    import vipcup_amd  # noqa: F401
    from vipcup_amd import _abi
    lib = _abi.lib()
    assert lib.vip_workspace_bytes(4, (C.c_int64 * 1)(129792), 1) == 129792       # JPEG planes: one byte per coefficient
    assert lib.vip_workspace_bytes(0, None, 0) == 0
    d = _abi.ConvDesc(B=256, H=25, W=25, Cin=1024, Cout=1024, kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Ho=25, Wo=25, groups=1, ldx=1024,
                      cin_off=0, ldy=1024, cout_off=0, ldr=0, res_off=0, ldw=1024, act_pre=1, act_post=0)
    buf = C.create_string_buffer(64)
    assert lib.vip_conv2d_kernel_name(C.byref(d), 0, 0, 0, buf, 64) == 0 and buf.value == b"gemm8p_kernel"   # K 1024, N 1024: the LDS-DMA kernel
    d.Cout, d.ldy = 320, 320
    assert lib.vip_conv2d_kernel_name(C.byref(d), 0, 0, 0, buf, 64) == 0 and buf.value.startswith(b"pwk_")      # N % 256 != 0
    d.B = 4
    d.H = d.W = d.Ho = d.Wo = 1
    assert lib.vip_conv2d_kernel_name(C.byref(d), 0, 0, 0, buf, 64) == 0 and buf.value == b"rows_gemm_kernel"
    d.Cin = 7
    assert lib.vip_conv2d_kernel_name(C.byref(d), 0, 0, 0, buf, 64) != 0          # argument checks run in the dry run too
