"""The host-only parts of the library (arena packer with its streaming-store path, host SHA-1
lane) built with -fsanitize=address,undefined and run on random batches."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packer_and_host_lane_are_clean_under_sanitizers(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    exe = str(tmp_path / "drv")
    csrc = os.path.join(ROOT, "vit_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-static-libasan", "-static-libubsan", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           os.path.join(ROOT, "tests", "host_sanitize_driver.cpp"), os.path.join(csrc, "packer.cpp"),
           os.path.join(csrc, "sha1_host.cpp"), "-lpthread", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_shared_worker_pool_under_thread_sanitizer(tmp_path):
    """work_pool.h: every task once, high-priority (packing) groups overtake the low-priority (host lane) group,
    and the pool never has more threads in tasks than it was sized to, however many queues submit."""
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    exe = str(tmp_path / "pool_drv")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "vit_amd", "csrc"),
           os.path.join(ROOT, "tests", "host_pool_driver.cpp"), "-lpthread", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pool clean" in r.stdout, (r.stdout + r.stderr)[-3000:]
