"""CPU: the host JPEG decoder (csrc/jpeg_host.cpp, plain C++) under AddressSanitizer + UBSan on mutated streams -
truncations, bit flips, spurious 0xFF markers, corrupted headers, cut spans - of baseline, restart-marker, 4:2:2,
grayscale and progressive files.  Any out-of-bounds access aborts the harness."""
import os
import shutil
import subprocess

import pytest

from tests.test_oracle_jpeg import _progressive, _rgb_coded, _variants
from tools.make_synth import synth_jpeg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_decoder_survives_mutated_streams(tmp_path):
    exe = tmp_path / "jpeg_fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           f"-I{ROOT}/include", os.path.join(ROOT, "tests", "fuzz", "jpeg_fuzz.cpp"),
           os.path.join(ROOT, "vip-cup-2022_amd", "csrc", "jpeg_host.cpp"), "-o", str(exe), "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    files = []
    corpus = {"synth0": synth_jpeg(0), "synth49": synth_jpeg(49), **_variants(), **_progressive(), **_rgb_coded()}
    for name, raw in corpus.items():
        p = tmp_path / f"{name}.jpg"
        p.write_bytes(raw)
        files.append(str(p))
    r = subprocess.run([str(exe), "400", *files], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0"})
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    assert "fuzzed" in r.stdout
    n_total, n_ok = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert n_total > 6000 and 0 < n_ok < n_total          # some mutations still decode, most are rejected or cut short
