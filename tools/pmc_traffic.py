"""HBM traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Units/corrections per MI355X_MICROARCH.md (HBM section): both counters are KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B/lane) streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, sys

FAMILIES = {"conv_igemm_kernel": ("conv_igemm_kernel",), "pw_gemm_kernel": ("pw_gemm_kernel",),
            "pwk_gemm_kernel": ("pwk_gemm_kernel", "pwk_direct_kernel"), "gemm8p_kernel": ("gemm8p_kernel",), "rows_gemm_kernel": ("rows_gemm_kernel",), "mlp_fused_kernel": ("mlp_fused_kernel",),
            "mlp_stream_kernel": ("mlp_stream_kernel",), "window_attn_kernel": ("window_attn_kernel", "window_attn_pipe_kernel"),
            "dwconv": ("dwconv_tile_kernel", "dwconv_kernel"), "layernorm": ("layernorm_kernel",),
            "scale_add_act": ("scale_add_act_kernel",), "gap": ("gap_kernel",), "se_gate": ("se_gate_kernel",),
            # kernels of the packed strict step (the GEMM names above are shared: its summary is a separate file)
            "mlp_h2_kernel": ("mlp_h2_kernel",), "dwconv_lds_h2_kernel": ("dwconv_lds_h2_kernel",), "attn_h2_kernel": ("attn_h2_kernel",),
            "h2_layernorm_kernel": ("h2_layernorm_kernel",)}


def load(d, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for fam, keys in FAMILIES.items():
                if any(k in r["Kernel_Name"] for k in keys):
                    tot[fam] += float(r["Counter_Value"])
                    n[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in FAMILIES:
    if fam not in nf:
        continue
    rd = fetch[fam] * 1024 * 2 / nf[fam]
    wr = write[fam] * 1024 / max(nw.get(fam, 1), 1)
    out[fam] = {"launches_profiled": nf[fam], "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                "hbm_bytes_per_launch": rd + wr}
out["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --steps 2 --warmup 1`, " \
               "VIP_STREAMS=1; FETCH_SIZE x2 (gfx950 correction), KiB -> bytes"
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa: E402,F401
from vipcup_amd import workloads  # noqa: E402
out["_source_digest"] = workloads.source_digest()      # bench.py reports `traffic` only from a summary of the SAME kernel sources
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
