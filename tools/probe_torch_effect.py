"""Does initialising torch's CUDA context in the same process change the kernel's speed?  Development probe."""
import statistics
import sys

import vit_amd

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode in ("init", "tensor"):
    import torch
    torch.cuda.init()
    torch.cuda.synchronize()
    if mode == "tensor":
        x = torch.zeros(1 << 20, device="cuda")
        torch.cuda.synchronize()
h = vit_amd.Hasher((0,))
b = h.synth_arena("blob", 0x5EED0002, 0, 1_000_000, 4096)
h.time_hash_device(b, warmup=3, iters=0)
ms = []
for _ in range(5):
    ms += h.time_hash_device(b, warmup=0, iters=10)
print(mode, "median %.4f ms  min %.4f" % (statistics.median(ms), min(ms)), flush=True)
