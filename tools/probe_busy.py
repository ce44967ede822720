#!/usr/bin/env python3
"""Run the hashing kernel and the load-free compute-ceiling kernel a few times each, for a
counter pass that compares how busy the vector ALUs are in the two (run under rocprofv3 --pmc)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd  # noqa: E402

h = vit_amd.Hasher((0,))
b = h.synth_arena("blob", 0x5EED0002, 0, 983040, 4096, slot_align=128)     # 15 objects per lane of the chip, exactly
ms = h.time_hash_device(b, warmup=2, iters=5)
peak = h.valu_peak(983040, 65, iters=5)
print(json.dumps({"sha1_ms": [round(x, 4) for x in ms], "valu_peak_ms": round(peak, 4)}))
