"""Is the JPEG -> scores path bit-reproducible from call to call on a mixed-size batch?  (diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import pipeline, zoo, workloads
from tools.make_synth import synth_jpeg
POISON = "--poison" in sys.argv


def poison():
    """hand NaN-filled blocks back to the caching allocator: the next torch.empty of a similar size gets them, and a kernel that reads
    memory it never wrote shows up as NaN / a changed result instead of depending on what a long process left behind"""
    if not POISON:
        return
    junk = [torch.full((n,), float("nan"), dtype=torch.float16, device="cuda") for n in
            [1 << k for k in range(8, 29)] + [3 * (1 << k) for k in range(8, 27)] + [5 * (1 << k) for k in range(8, 26)]]
    junk += [torch.full((n,), -1, dtype=torch.int32, device="cuda") for n in (64, 1024, 1 << 16)]
    torch.cuda.synchronize()
    del junk


ids = list(range(100, 115)) + [149]
raws = [synth_jpeg(i) for i in ids]
b1 = pipeline.decode_jpegs(raws)
poison()
b2 = pipeline.decode_jpegs(raws)
print("sizes", b1.sizes_host[-1], "rgb equal:", torch.equal(b1.rgb, b2.rgb))
for hw in (200, 224):
    r1 = b1.resized(hw, hw)
    poison()
    r2 = b2.resized(hw, hw)
    print("resized", hw, "equal:", torch.equal(r1, r2), "finite:", bool(torch.isfinite(r1.float()).all()))
for key in ([] if "--fast" in sys.argv else ["vit_tiny_patch16_224", "resnet_rs50"]):
    spec, model = zoo.build_member(key)
    x = b1.resized(spec.input_hw, spec.input_hw)
    z = []
    for _ in range(3):
        poison()
        z.append(model.logits(x).float().clone())
    print(key, "logits equal across calls:", torch.equal(z[0], z[1]) and torch.equal(z[1], z[2]), "max diff", (z[0] - z[2]).abs().max().item(),
          "per-image diff", [(round(v, 7)) for v in (z[0] - z[1]).abs().flatten().tolist() if v > 0])
wl = workloads.build("ensemble4", batch=16, jpegs=raws)
s = []
for _ in range(4):
    poison()
    s.append(wl.step().clone())
print("joined steps equal:", [torch.equal(s[0], t) for t in s[1:]], [(s[0] - t).abs().max().item() for t in s[1:]], "finite:", bool(torch.isfinite(torch.stack(s)).all()))
poison()
wl.step(pipelined=True)
poison()
p1 = wl.step(pipelined=True).clone()
poison()
p2 = wl.flush().clone()
print("pipelined vs joined:", (p1 - s[1]).abs().max().item(), (p2 - s[1]).abs().max().item(), "which image:", (p1 - s[1]).abs().argmax().item())
