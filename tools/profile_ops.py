"""Per-operator time / bytes breakdown of a workload step (HIP events around every op launch).
    python tools/profile_ops.py [workload] [batch] [top]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "ensemble"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 60
wl = workloads.build(name, B)
for _ in range(2):
    wl.step(serial=True)
torch.cuda.synchronize()
rec = []
NAMES = ["conv2d", "dense", "mlp", "se_gate", "dwconv2d", "layernorm", "pool2d", "global_avgpool", "gap_dense_f32", "scale_add_act",
         "window_attention", "mhsa", "vit_tokens", "cls_dense_f32", "radix_combine"]


def tensors(objs):
    for o in objs:
        if isinstance(o, torch.Tensor):
            yield o
        elif hasattr(o, "w") and isinstance(getattr(o, "w"), torch.Tensor):
            yield o.w


def wrap(nm, fn):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = fn(*a, **k)
        e1.record()
        ins = list(tensors(list(a) + list(k.values())))
        outs = [y] if isinstance(y, torch.Tensor) else list(tensors(y))
        by = sum(t.numel() * t.element_size() for t in ins + outs if t.is_cuda)
        extra = " ".join(f"{kk}={vv}" for kk, vv in k.items() if not isinstance(vv, torch.Tensor) and kk in ("stride", "act", "k", "mode"))
        key = f"{nm} " + " ".join("x".join(map(str, t.shape)) for t in ins[:2]) + " -> " + "x".join(map(str, outs[0].shape)) + " " + extra
        rec.append((nm, key, by, e0, e1))
        return y
    return f


for nm in NAMES:
    setattr(ops, nm, wrap(nm, getattr(ops, nm)))
wl.step(serial=True)
torch.cuda.synchronize()
per_op = collections.OrderedDict()
agg = collections.OrderedDict()
for nm, key, by, e0, e1 in rec:
    ms = e0.elapsed_time(e1)
    d = per_op.setdefault(nm, [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += by
    d = agg.setdefault(key, [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += by
tot = sum(v[1] for v in per_op.values())
print(f"total op time {tot:.2f} ms over {len(rec)} launches")
for k, v in sorted(per_op.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:8.3f} ms {100*v[1]/tot:5.1f}% n={v[0]:4d} {v[2]/v[1]/1e6:7.0f} GB/s  {k}")
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:TOP]:
    if k.startswith("conv2d") or k.startswith("dense"):
        continue
    print(f"{v[1]:7.3f} ms n={v[0]:3d} {v[2]/v[1]/1e6:7.0f} GB/s  {k}")
