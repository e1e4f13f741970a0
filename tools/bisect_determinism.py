"""Which operator of a member is not bit-reproducible?  tools/stress_determinism.py found gcvit_tiny's scores differing between
pipelined and joined bench steps; this hooks every vipcup_amd.ops call of ONE member, records each output, and compares the records
of repeated runs (serial, on a side stream, and on a side stream while another member runs concurrently on a second stream)
with a reference run - the first operator whose output differs is the culprit.

    python tools/bisect_determinism.py [--member gcvit_tiny] [--batch 16] [--iters 20] [--noise convnext_tiny_in22k]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from tools.make_synth import synth_jpeg  # noqa: E402
from vipcup_amd import ops, pipeline, zoo  # noqa: E402

HOOKED = ["conv2d", "dense", "mlp", "se_gate", "dense_split", "dwconv2d", "layernorm", "pool2d", "global_avgpool", "gap_dense_f32",
          "gap_ln_dense_f32", "scale_add_act", "window_attention", "mhsa", "vit_tokens", "mul", "radix_combine", "head_prob"]


class Recorder:
    def __init__(self):
        self.rec = []
        self.depth = 0
        self.orig = {n: getattr(ops, n) for n in HOOKED}

    def __enter__(self):
        for n, fn in self.orig.items():
            setattr(ops, n, self._wrap(n, fn))
        return self

    def __exit__(self, *exc):
        for n, fn in self.orig.items():
            setattr(ops, n, fn)

    def _wrap(self, name, fn):
        def w(*a, **k):
            self.depth += 1
            try:
                out = fn(*a, **k)
            finally:
                self.depth -= 1
            outs = out if isinstance(out, tuple) else (out,)
            shp = tuple(a[0].shape) if a and isinstance(a[0], torch.Tensor) else ()
            for o in outs:
                self.rec.append((f"{'  ' * self.depth}{name}{shp}", o.clone()))
            return out
        return w


def run(model, x):
    with Recorder() as r:
        model.logits(x)
    return r.rec


def first_diff(ref, got):
    for i, ((n0, t0), (n1, t1)) in enumerate(zip(ref, got)):
        if not torch.equal(t0, t1):
            d = (t0.float() - t1.float()).abs()
            return i, n0, float(d.max()), int((d > 0).sum()), t0.numel()
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--member", default="gcvit_tiny")
    ap.add_argument("--noise", default="convnext_tiny_in22k")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    spec, model = zoo.build_member(a.member)
    nspec, noise = zoo.build_member(a.noise)
    raws = [synth_jpeg(100 + i) for i in range(a.batch)]
    batch = pipeline.decode_jpegs(raws)
    x = batch.resized(spec.input_hw, spec.input_hw)
    xn = batch.resized(nspec.input_hw, nspec.input_hw)
    torch.cuda.synchronize()
    ref = run(model, x)
    torch.cuda.synchronize()
    print(f"{a.member}: {len(ref)} recorded operator outputs per forward pass, batch {a.batch}")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    modes = {"serial, default stream": (None, False), "side stream alone": (sa, False), "side stream + a second member on another stream": (sa, True)}
    for label, (st, with_noise) in modes.items():
        hits = {}
        for it in range(a.iters):
            torch.cuda.synchronize()
            if with_noise:
                with torch.cuda.stream(sb):
                    for _ in range(2):
                        noise.logits(xn)
            if st is None:
                got = run(model, x)
            else:
                with torch.cuda.stream(st):
                    got = run(model, x)
            torch.cuda.synchronize()
            fd = first_diff(ref, got)
            if fd is not None:
                hits.setdefault((fd[0], fd[1]), []).append(fd[2:])
        if not hits:
            print(f"  [{label}] {a.iters} runs: every operator output bit-identical to the reference run")
        for (idx, name), v in sorted(hits.items()):
            print(f"  [{label}] first differing output #{idx} {name.strip()}: in {len(v)} of {a.iters} runs; max |d| {max(t[0] for t in v):.3e}, "
                  f"up to {max(t[1] for t in v)} of {v[0][2]} elements")
            lo = max(0, idx - 3)
            print("     context: " + " | ".join(n.strip() for n, _ in ref[lo:idx + 1]))


if __name__ == "__main__":
    main()
