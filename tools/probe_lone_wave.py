"""Compute-only rate of the SHA-1 instruction stream with one, two and five waves per SIMD (vit_valu_peak):
what a wave that has its SIMD to itself loses to its own dependent instructions."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vit_amd
h = vit_amd.Hasher((0,))
for lanes, label in ((64, "one wave on the chip"), (65536, "one wave per SIMD"), (2*65536, "two per SIMD"), (5*65536, "five per SIMD")):
    ms = h.valu_peak(lanes, 2048, iters=3)
    waves_per_simd = max(1, lanes // 65536)
    print("%-22s %.3f ms for 2048 blocks per lane: %.3f us per block per wave, %.3f us per block of SIMD time" % (label, ms, ms * 1e3 / 2048, ms * 1e3 / 2048 / waves_per_simd))
