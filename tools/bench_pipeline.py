"""Preprocess path in isolation (SURVEY.md §8d 'Preprocess kernel'): host Huffman decode images/s per core, then the
GPU half - dequant + IDCT + upsample + colour (vip_jpeg_idct_rgb_u8) and bicubic resize + /255 (vip_resize_bicubic_norm_f16)
- with their algorithmic bytes.    python tools/bench_pipeline.py [n_images]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import vipcup_amd  # noqa
from vipcup_amd import pipeline
from tools.make_synth import synth_jpeg, synth_pixels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = [synth_jpeg(i) for i in range(n)]
prog = []
for i in range(0, n, 8):
    b = io.BytesIO(); Image.fromarray(synth_pixels(i)).save(b, format="JPEG", quality=80, subsampling=2, progressive=True)
    prog.append(b.getvalue())

for label, raws in (("baseline 200x200 (SURVEY mix of qualities / subsamplings)", base), ("progressive 200x200 q80 4:2:0", prog)):
    for th in (1, 16):
        pipeline.entropy_decode(raws[:8], threads=th)
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps):
            desc, coef = pipeline.entropy_decode(raws, threads=th)
        dt = (time.perf_counter() - t0) / reps
        print(f"host Huffman, {label}: {th:2d} thread(s) {len(raws)/dt:9.0f} img/s  ({len(raws)/dt/th:7.0f} img/s/thread, "
              f"{sum(len(r) for r in raws)/len(raws)/1e3:.1f} KB/file, {coef.size*2/len(raws)/1e3:.0f} KB coef/img)")

raws = base[:256]
batch = pipeline.decode_jpegs(raws)
torch.cuda.synchronize()
desc, coef = pipeline.entropy_decode(raws)
coef_d = torch.from_numpy(coef).cuda()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
# H2D + GPU half (what decode_jpegs does after the host stage)
t0 = time.perf_counter()
for _ in range(reps):
    b = pipeline.decode_jpegs(raws)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / reps
print(f"decode_jpegs (host Huffman 16 threads + H2D + GPU), 256 images: {full*1e3:.2f} ms = {256/full:.0f} img/s")
out_bytes = sum(h * w * 3 for h, w in batch.sizes_host)
for hw in (200, 224):
    x = batch.resized(hw, hw)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        batch.resized(hw, hw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = out_bytes + 256 * hw * hw * 8 * 2
    print(f"resize_norm -> {hw}x{hw}x8 fp16: {ms*1e3:.1f} us, {by/ms/1e6:.0f} GB/s (u8 RGB in + fp16 NHWC8 out)")
