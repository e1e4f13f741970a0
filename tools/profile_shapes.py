"""Per-shape time breakdown of the conv/GEMM launches of a workload (HIP events around each launch).
    python tools/profile_shapes.py [workload] [batch]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "ensemble"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
wl = workloads.build(name, B)
for _ in range(2):
    wl.step(serial=True)
torch.cuda.synchronize()

rec = []
orig_conv, orig_dense = ops.conv2d, ops.dense

def conv2d(x, cw, stride=1, pad=(0, 0, 0, 0), act=None, act_post=None, residual=None, out=None, cin_off=0, cout_off=0, gate=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig_conv(x, cw, stride, pad, act, act_post, residual, out, cin_off, cout_off, gate)
    e1.record()
    M = y.shape[0] * y.shape[1] * y.shape[2]
    rec.append((f"conv M={M} N={cw.cout} K={cw.kh*cw.kw*cw.cin_g} k{cw.kh} s{stride} g{cw.groups} Cin={cw.cin}", 2.0*M*cw.cout*cw.kh*cw.kw*cw.alg_cin_g,
                2.0*(x.numel()+y.numel()*(2 if residual is not None else 1)), e0, e1))
    return y

def dense(x, cw, act=None, act_post=None, residual=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig_dense(x, cw, act, act_post, residual)
    e1.record()
    M = x.numel() // x.shape[-1]
    rec.append((f"dense M={M} N={cw.cout} K={x.shape[-1]}", 2.0*M*cw.cout*x.shape[-1], 2.0*(x.numel()+y.numel()*(2 if residual is not None else 1)), e0, e1))
    return y

ops.conv2d, ops.dense = conv2d, dense
wl.step(serial=True)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for k, fl, by, e0, e1 in rec:
    d = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl; d[3] += by
tot = sum(v[1] for v in agg.values())
print(f"total conv/dense time {tot:.2f} ms over {len(rec)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1]:7.3f} ms {100*v[1]/tot:5.1f}% n={v[0]:3d} {v[2]/v[1]/1e9:7.1f} TF {v[3]/v[1]/1e6:7.0f} GB/s  {k}")
