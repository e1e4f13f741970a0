"""hipcc -Rpass-analysis=kernel-resource-usage of one csrc file, one line per kernel: VGPRs, spills, scratch, LDS.
usage: python tools/kernel_resources.py vip-cup-2022_amd/csrc/conv_h2.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Iinclude", "-Ivip-cup-2022_amd/csrc",
       "-Wno-unused-result", "-ffp-contract=fast", "-DVIP_BUILD_EXPERIMENTS=0", "-x", "hip", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:]]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{name:60s} vgpr {c.get('VGPRs', '?'):>4s} agpr {c.get('AGPRs', '?'):>3s} spill {c.get('VGPRs Spill', '?'):>3s} scratch {c.get('ScratchSize [bytes/lane]', '?'):>4s} "
          f"lds {c.get('LDS Size [bytes/block]', '?'):>6s} occ {c.get('Occupancy [waves/SIMD]', '?')}")
