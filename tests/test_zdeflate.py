"""zdeflate.h (the zlib stream a GPU lane writes for one object) and zwave.h (the same stream as a GPU workgroup works
it out, run here by its plain C++ emulation) against the zlib library on the host: thousands of inputs of every kind and
size, byte for byte, once optimised and once under ASan/UBSan, and once the workgroup formulation alone on many more
inputs of the sizes it takes (around its size classes and tiles too)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags,rounds,wave_only", [(["-O2"], 4000, False),
                                                    (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                                      "-static-libasan", "-static-libubsan"], 400, False),
                                                    (["-O2"], 5000, True)])
def test_restated_deflate_writes_zlibs_bytes(tmp_path, flags, rounds, wave_only):
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    exe = str(tmp_path / "zd")
    cmd = ["g++", "-std=c++17"] + flags + ["-I" + os.path.join(ROOT, "vit_amd", "csrc"),
                                            os.path.join(ROOT, "tests", "zdeflate_driver.cpp"), "-lz", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, str(rounds), "2024"] + (["wave-only"] if wave_only else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, (r.stdout + r.stderr)[-2000:]
