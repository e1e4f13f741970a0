"""Build libvipcup_hip.so (gfx950) in-tree with hipcc.

Cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
Usage: python vip-cup-2022_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libvipcup_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# Experiments - kernels that measured slower than what they would replace - are only compiled on request:
#   VIP_BUILD_EXPERIMENTS=1 python vip-cup-2022_amd/build.py --force
EXPERIMENTS = os.environ.get("VIP_BUILD_EXPERIMENTS", "0") == "1"
EXPERIMENT_SOURCES = {"dwconv_mfma.hip", "mbconv_fused.hip", "gcvit_block14.hip"}
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{CSRC}",
            "-Wno-unused-result", "-ffp-contract=fast", f"-DVIP_BUILD_EXPERIMENTS={int(EXPERIMENTS)}"] \
    + os.environ.get("VIP_EXTRA_CXXFLAGS", "").split()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".cpp")) and (EXPERIMENTS or f not in EXPERIMENT_SOURCES))


# Per-file flags.  -fno-slp-vectorize: in these two files hipcc's SLP vectoriser pairs scalar fp32 operations into packed VALU ops with
# half-SWAPPED sources (v_pk_*_f32 ... op_sel:[1,0] ...) - the instruction form behind the round-2 window-attention failure next to
# MFMA-heavy co-runners (tools/repro/); nothing in them is VALU-bound, and tools/isa_lint.py (a CPU test) keeps every
# shipped kernel free of that form.
EXTRA_FLAGS = {"se_gate.hip": ["-fno-slp-vectorize"], "jpeg_pipeline.hip": ["-fno-slp-vectorize"]}

# sources that #include another SOURCE file (one kernel family, two arithmetic modes)
INCLUDES_SOURCE = {"conv_h2.hip": ["conv_igemm.hip"]}


def _deps_mtime(src=None):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(INCLUDE, "vipcup_hip.h"))
    hdrs += [os.path.join(CSRC, f) for f in INCLUDES_SOURCE.get(os.path.basename(src or ""), [])]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + (".exp.o" if EXPERIMENTS else ".o"))
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _deps_mtime(src)):
        return obj
    cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if (force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
