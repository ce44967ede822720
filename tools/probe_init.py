import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t0 = time.perf_counter()
import vit_amd
t1 = time.perf_counter()
from vit_amd import _abi
lib = _abi.load()
t2 = time.perf_counter()
import ctypes as C
n = C.c_int(0)
h = vit_amd.Hasher((0,))
t3 = time.perf_counter()
d = h.hash_blobs([b"hello\n"])
t4 = time.perf_counter()
d = h.hash_blobs([b"hello\n"])
t5 = time.perf_counter()
d = h.hash_blobs([b"x" * 4096] * 400)
t6 = time.perf_counter()
d = h.hash_blobs([b"x" * 4096] * 400)
t7 = time.perf_counter()
h.close()
t8 = time.perf_counter()
print("import %.1f  load %.1f  create %.1f  first hash %.1f  second %.2f  400x4K first %.2f  again %.2f  close %.1f ms" % tuple(
    1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7)))
t0 = time.perf_counter()
h2 = vit_amd.Hasher((0,))                  # HIP is warm now: this is what the hasher itself costs
t1 = time.perf_counter()
h2.hash_blobs([b"hello\n"])
t2 = time.perf_counter()
h2.close()
print("second hasher in the same process: create %.2f ms, first hash %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
