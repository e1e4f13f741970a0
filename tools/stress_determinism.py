"""Are the scores of the bench step bit-reproducible under the schedule bench.py times (3 member streams, steps forked before the
previous one is joined, the next batch's decode enqueued between fork and join)?  DESIGN.md section 8.6 (round 2) left one observation
open: pipelined vs joined steps once differed by 1.5e-4 on a mixed-size batch in a long process.  This runs the same batch through
`--iters` pipelined steps and compares every member's scores bit for bit with a joined reference step; with --flood a fourth, unrelated
stream keeps the chip busy with copies (the "many live streams" condition of the long process); --no-record-stream switches the
allocator's cross-stream bookkeeping (ensemble.MemberStreams.predict_all) off for an A/B.

    python tools/stress_determinism.py [--workload ensemble8] [--batch 16|256] [--iters 300] [--flood] [--no-record-stream]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from tools.make_synth import synth_jpeg  # noqa: E402
from vipcup_amd import ensemble, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ensemble8")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--flood", action="store_true")
    ap.add_argument("--no-record-stream", action="store_true")
    ap.add_argument("--precision", default="fast")
    a = ap.parse_args()
    if a.no_record_stream:
        ensemble._record_stream = lambda t, s: None
    ids = [100 + i for i in range(a.batch - 1)] + [149]          # one 256x192 image: the resize branch, a mixed-size decode
    raws = [synth_jpeg(i) for i in ids]
    wl = workloads.build(a.workload, batch=a.batch, jpegs=raws, precision=a.precision)
    wl.step()
    wl.step()
    ref = wl.member_scores.clone()
    torch.cuda.synchronize()
    junk = None
    flood_stream = torch.cuda.Stream() if a.flood else None
    if a.flood:
        junk = torch.empty((1 << 28,), dtype=torch.uint8, device="cuda")
    bad = {}
    worst = 0.0
    got = []
    for it in range(a.iters):
        if a.flood:
            with torch.cuda.stream(flood_stream):
                junk[: 1 << 27].copy_(junk[1 << 27:])
        wl.step(pipelined=True)
        if wl.scores is not None and it > 0:
            got.append(wl.member_scores.clone())
    wl.flush()
    got.append(wl.member_scores.clone())
    torch.cuda.synchronize()
    for k, g in enumerate(got):
        d = (g - ref).abs()
        if float(d.max()) != 0.0:
            worst = max(worst, float(d.max()))
            for m in torch.nonzero(d.max(dim=1).values).flatten().tolist():
                bad.setdefault(wl.members[m], []).append((k, float(d[m].max())))
    print(f"[determinism] {a.workload} batch {a.batch} precision {a.precision} flood={a.flood} record_stream={not a.no_record_stream}: "
          f"{len(got)} pipelined steps vs the joined reference: {'ALL BIT-IDENTICAL' if not bad else 'MISMATCHES'}; worst |dp| {worst:.3e}")
    for m, v in bad.items():
        print(f"   {m}: {len(v)} steps differ, first at step {v[0][0]}, max {max(x[1] for x in v):.3e}")
    wl.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
