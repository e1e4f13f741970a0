"""Input pipeline — the MI355X counterpart of dataset/dataset.py (``build_decoder.decode`` :22-39,
``build_dataset`` :64-102) and of the TTA ops of dataset/augment.py (:115-120, :142-182).

Differences from the reference, by design: every JPEG is entropy-decoded ONCE on the
host (C++ threads inside libvipcup_hip.so) and turned into RGB once on the GPU; each member resolution
(200, 224, ...) is then one resize launch on the resident uint8 pixels — the reference re-reads and
re-decodes the files for every model.
"""
import ctypes as C
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _abi
from .ops import _p, _stream

_TABLE_DEV: Dict[int, torch.Tensor] = {}


def bicubic_table(device) -> torch.Tensor:
    key = torch.device(device).index or 0
    if key not in _TABLE_DEV:
        host = np.zeros((1025 * 2,), dtype=np.float32)
        _abi.check(_abi.lib().vip_bicubic_table_f32(host.ctypes.data_as(C.c_void_p)), "vip_bicubic_table_f32")
        _TABLE_DEV[key] = torch.from_numpy(host).to(device)
    return _TABLE_DEV[key]


MAX_JPEG_PIXELS = int(os.environ.get("VIP_MAX_JPEG_PIXELS", str(64 << 20)))   # per image; the task's images are 200 x 200


def entropy_decode(jpegs: Sequence[bytes], threads: int = 0, pinned: bool = False):
    """Host stage: list of JPEG byte strings -> (desc array (ctypes), coef int16 numpy array); with ``pinned`` the
    coefficients come back as a page-locked torch tensor instead (torch caches such buffers), so that the H2D copy in
    ``decode_entropy`` is asynchronous and runs at PCIe rate.
    Raises VipError for streams outside the supported Huffman subset (SOF0/1/2, 8-bit, 1 or 3 components) (the reference raises too: TF)."""
    lib = _abi.lib()
    n = len(jpegs)
    if threads <= 0:
        threads = min(16, os.cpu_count() or 1)
    bufs = [np.frombuffer(b, dtype=np.uint8) for b in jpegs]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[len(b) for b in jpegs])
    desc = (_abi.JpegDesc * n)()
    total = 0
    tmp = _abi.JpegDesc()
    need = C.c_size_t(0)
    for i in range(n):
        _abi.check(lib.vip_jpeg_probe_h(ptrs[i], lens[i], C.byref(tmp), C.byref(need)), "vip_jpeg_probe_h")
        if tmp.width * tmp.height > MAX_JPEG_PIXELS:      # a corrupt header can claim 65535 x 65535: do not allocate for it
            raise _abi.VipError(f"jpeg {i}: {tmp.width}x{tmp.height} exceeds VIP_MAX_JPEG_PIXELS={MAX_JPEG_PIXELS}")
        total += need.value
    coef_t = torch.empty((max(total, 1),), dtype=torch.int16, pin_memory=True) if pinned else None
    coef = coef_t.numpy() if pinned else np.empty((max(total, 1),), dtype=np.int16)
    used = C.c_size_t(0)
    st = lib.vip_jpeg_entropy_decode_h(ptrs, lens, n, desc, coef.ctypes.data_as(C.c_void_p), coef.size, C.byref(used),
                                       threads)
    _abi.check(st, "vip_jpeg_entropy_decode_h")
    return desc, (coef_t[:used.value] if pinned else coef[:used.value])


class DecodedBatch:
    """uint8 RGB pixels of a batch, resident on the GPU: ``rgb [n,maxH,maxW,3]``, ``sizes [n,2]`` (h,w)."""

    def __init__(self, rgb: torch.Tensor, sizes: torch.Tensor, sizes_host: List[Tuple[int, int]]):
        self.rgb, self.sizes, self.sizes_host = rgb, sizes, sizes_host

    def __len__(self):
        return self.rgb.shape[0]

    def resized(self, out_h: int, out_w: int, c_out: int = 8) -> torch.Tensor:
        """cast -> tf.image.resize(bicubic) -> /255 (dataset/dataset.py:31-38) -> fp16 NHWC, channels padded."""
        n, maxH, maxW, _ = self.rgb.shape
        out = torch.empty((n, out_h, out_w, c_out), dtype=torch.float16, device=self.rgb.device)
        st = _abi.lib().vip_resize_bicubic_norm_f16(_p(self.rgb), _p(self.sizes), _p(bicubic_table(self.rgb.device)), n,
                                                    maxH, maxW, _p(out), out_h, out_w, c_out, _stream())
        _abi.check(st, "vip_resize_bicubic_norm_f16")
        return out


def decode_jpegs(jpegs: Sequence[bytes], device="cuda", threads: int = 0) -> DecodedBatch:
    """``tf.image.decode_jpeg(channels=3)`` for a batch (dataset/dataset.py:24-28)."""
    return decode_entropy(entropy_decode(jpegs, threads), device)


def decode_entropy(host_stage, device="cuda") -> DecodedBatch:
    """Device half of ``decode_jpegs``: ``host_stage`` = ``entropy_decode(...)`` (which may have run on another thread
    while the GPU was busy with the previous batch)."""
    desc, coef = host_stage
    n = len(desc)
    sizes_host = [(int(d.height), int(d.width)) for d in desc]
    maxH = max(h for h, _ in sizes_host)
    maxW = max(w for _, w in sizes_host)
    max_blocks = max(sum(d.blocks_w[c] * d.blocks_h[c] for c in range(d.ncomp)) for d in desc)
    if isinstance(coef, torch.Tensor):     # page-locked tensor from entropy_decode(pinned=True): asynchronous copy, and the
        coef_d = coef.to(device, non_blocking=True)   # caching host allocator keeps the buffer until the copy has run
    else:
        coef_d = torch.from_numpy(coef).to(device)
    desc_bytes = np.frombuffer(bytes(desc), dtype=np.uint8)
    desc_d = torch.from_numpy(desc_bytes.copy()).to(device)
    planes = torch.empty((coef_d.numel(),), dtype=torch.uint8, device=device)
    rgb = torch.zeros((n, maxH, maxW, 3), dtype=torch.uint8, device=device)
    st = _abi.lib().vip_jpeg_idct_rgb_u8(_p(coef_d), _p(desc_d), n, max_blocks, _p(planes), _p(rgb), maxH, maxW,
                                         _stream())
    _abi.check(st, "vip_jpeg_idct_rgb_u8")
    sizes = torch.tensor(sizes_host, dtype=torch.int32, device=device)
    return DecodedBatch(rgb, sizes, sizes_host)


def apply_augment(x: torch.Tensor, hflip, vflip, gray) -> torch.Tensor:
    """Deterministic form of dataset/augment.py ``apply_augment`` (:153-182): per-image flags instead of the
    reference's TF RNG draws (p=0.8 gate, hflip .5, vflip .5, gray .3) — the caller owns the randomness."""
    B, H, W, Cc = x.shape
    flags = (torch.as_tensor(hflip, dtype=torch.int32) | (torch.as_tensor(vflip, dtype=torch.int32) << 1) |
             (torch.as_tensor(gray, dtype=torch.int32) << 2)).to(x.device)
    assert flags.numel() == B
    out = torch.empty_like(x)
    st = _abi.lib().vip_tta_augment_f16(_p(x), _p(out), _p(flags), B, H, W, Cc, _stream())
    _abi.check(st, "vip_tta_augment_f16")
    return out


def build_dataset(paths: Sequence[str], batch_size: int, img_size: Tuple[int, int], device="cuda", threads: int = 0):
    """Generator form of ``build_dataset(paths, labels=None, augment=False, repeat=False, shuffle=False)``
    (dataset/dataset.py:64-102): yields fp16 NHWC batches ``[bs, H, W, 8]`` in path order; the last batch is
    short (``drop_remainder=False``)."""
    for i in range(0, len(paths), batch_size):
        chunk = paths[i:i + batch_size]
        raw = []
        for p in chunk:
            with open(p, "rb") as f:  # tf.io.read_file (:24)
                raw.append(f.read())
        yield decode_jpegs(raw, device, threads).resized(img_size[0], img_size[1])


def calibration_batch(n: int = 16, seed: int = 20221, device="cuda") -> DecodedBatch:
    """A fixed, seeded batch of synthetic 200x200 RGB images (smooth colour field + pixel noise: the statistics of the
    synthetic test set, SURVEY.md §8d, from a different seed) for ops.calibration().  With real checkpoints pass a
    few real images through ``decode_jpegs`` instead - the correction only needs typical per-channel means."""
    g = torch.Generator().manual_seed(seed)
    low = torch.randn((n, 3, 8, 8), generator=g) * 48.0 + 128.0
    field = torch.nn.functional.interpolate(low.clamp(0, 255), size=(200, 200), mode="bicubic", align_corners=False)
    img = (field + torch.randn((n, 3, 200, 200), generator=g) * 12.0).clamp(0, 255).to(torch.uint8)
    rgb = img.permute(0, 2, 3, 1).contiguous().to(device)
    sizes = torch.tensor([[200, 200]] * n, dtype=torch.int32, device=device)
    return DecodedBatch(rgb, sizes, [(200, 200)] * n)
