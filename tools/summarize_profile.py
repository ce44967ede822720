#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into the
small text/JSON summaries kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01

Writes <out>_kernel_stats.csv (verbatim copy of the --stats table),
<out>_pmc.txt (per-kernel counter averages over the timed launches) and, when a
FETCH_SIZE/WRITE_SIZE pass is present, profiles/hbm_traffic.json with the HBM
bytes per launch of the dominant kernel (FETCH_SIZE doubled, as the gfx950
counter tallies 128-byte requests at 64 bytes; both are in KiB).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def load_pmc(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    for f in glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, dst + "_kernel_stats.csv")
    lines = []
    traffic = {}
    for sub in sorted(os.listdir(src)):
        p = os.path.join(src, sub)
        if not os.path.isdir(p) or not sub.startswith("pmc"):
            continue
        pmc = load_pmc(p)
        lines.append("== pass %s" % sub)
        for kern, ctrs in sorted(pmc.items()):
            if "sha1" not in kern and "valu_peak" not in kern:
                continue
            lines.append("  kernel %s" % kern)
            for c, vals in sorted(ctrs.items()):
                # the first dispatches are warm-up launches (code load, cold caches): report the median
                v = sorted(vals)[len(vals) // 2]
                lines.append("    %-24s median %.6g over %d dispatches" % (c, v, len(vals)))
                if c in ("FETCH_SIZE", "WRITE_SIZE") and "sha1" in kern:
                    traffic.setdefault(kern, {})[c] = v
    open(dst + "_pmc.txt", "w").write("\n".join(lines) + "\n")
    for kern, t in traffic.items():
        if "FETCH_SIZE" in t:
            fetch = t["FETCH_SIZE"] * 1024 * 2          # KiB -> bytes, x2 gfx950 correction for wide streaming reads
            write = t.get("WRITE_SIZE", 0.0) * 1024
            import subprocess
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            try:
                commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True,
                                                 cwd=os.path.dirname(os.path.abspath(__file__))).strip()
            except Exception:
                commit = None                                # no history on the GPU box: filled in when the summary is copied
            json.dump({"kernel": kern, "summary": dst + "_pmc.txt", "commit": commit,
                       "kernel_source_sha256": bench.kernel_source_sha256(),
                       "objects_per_launch": bench.N_BLOBS, "object_bytes": bench.BLOB_LEN,
                       "bytes_per_launch": int(fetch + write), "fetch_bytes_corrected": int(fetch),
                       "write_bytes": int(write), "raw_FETCH_SIZE_KiB": t["FETCH_SIZE"],
                       "raw_WRITE_SIZE_KiB": t.get("WRITE_SIZE"),
                       "note": "FETCH_SIZE doubled per the gfx950 calibration (128-B requests tallied at 64 B)"},
                      open(os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json"), "w"), indent=1)
            break
    print("\n".join(lines))


if __name__ == "__main__":
    main()
