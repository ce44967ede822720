#!/usr/bin/env python3
"""Condense what tools/collect_deflate_wave_pmc.sh collected into the text summary kept under profiles/.

    python tools/summarize_deflate_pmc.py gpurun_out/zwpmc profiles/r03_deflate_pmc.txt

Per content kind (random: 262144 resident 4 KiB blobs; text: 131072 text-like 4 KiB blobs through vit_deflate_objects)
and per kernel the counters are SUMMED over the run's launches (every probe runs its batch twice), and the fabric traffic
is put against the input bytes those launches deflated.  FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is given as
reported and doubled (on gfx950 the counter tallies a wide streaming read's 128-byte requests at 64 bytes; the scattered
reads of these kernels are not calibrated, so the truth lies between the two)."""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
INPUT = {"random": 2 * 262144 * 4106, "text": 2 * 131072 * 4106}     # framed bytes: "blob 4096\0" + 4096
lines = ["Deflate kernels, one workgroup per object with its tables in LDS (deflate_wave_kernel<bytes, bucket bits, position bits>),",
         "collected by tools/collect_deflate_wave_pmc.sh (one rocprofv3 pass per counter group), condensed by tools/summarize_deflate_pmc.py.",
         "Sums over the launches of a run; every probe deflates its batch twice.", ""]
for kind in ("random", "text"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, kind + "_*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "deflate" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append("== %s content: %s" % (kind, "tools/probe_deflate.py 262144 4096 (resident arena)" if kind == "random"
                                        else "tools/probe_deflate_text.py 131072 4096 (vit_deflate_objects, 64 MiB chunks)"))
    for f in glob.glob(os.path.join(src, kind + "_kt", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "deflate" in r["Name"]:
                lines.append("  %-48s %3s launches, %10.3f ms in all, %8.3f ms each" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"],
                                                                                          float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
    fetch = sum(sum(v.get("FETCH_SIZE", [])) for v in agg.values()) * 1024
    write = sum(sum(v.get("WRITE_SIZE", [])) for v in agg.values()) * 1024
    inp = INPUT[kind]
    lines.append("  input %.3f GB; FETCH_SIZE %.3f GB (doubled %.3f GB), WRITE_SIZE %.3f GB" % (inp / 1e9, fetch / 1e9, 2 * fetch / 1e9, write / 1e9))
    lines.append("  -> fabric bytes per input byte: %.1f as reported (%.1f read + %.1f written), %.1f with FETCH_SIZE doubled"
                 % ((fetch + write) / inp, fetch / inp, write / inp, (2 * fetch + write) / inp))
    for kern, ctrs in sorted(agg.items()):
        lines.append("  kernel %s" % kern)
        for c, vals in sorted(ctrs.items()):
            lines.append("    %-24s %16.0f  over %d launches" % (c, sum(vals), len(vals)))
        if "SQ_THREAD_CYCLES_VALU" in ctrs and sum(ctrs.get("SQ_ACTIVE_INST_VALU", [0])):
            lines.append("    lanes active per VALU instruction: %.2f of 64" % (sum(ctrs["SQ_THREAD_CYCLES_VALU"]) / sum(ctrs["SQ_ACTIVE_INST_VALU"])))
        if sum(ctrs.get("SQ_WAVE_CYCLES", [0])):
            wc = sum(ctrs["SQ_WAVE_CYCLES"])
            lines.append("    of the wave-cycles: %.1f %% issuing, %.1f %% waiting for an instruction's operands"
                         % (100 * sum(ctrs.get("SQ_ACTIVE_INST_ANY", [0])) / wc, 100 * sum(ctrs.get("SQ_WAIT_INST_ANY", [0])) / wc))
    for f in sorted(glob.glob(os.path.join(src, kind + "_kt.log"))):
        for ln in open(f):
            if ln.startswith("{"):
                lines.append("  probe: " + ln.strip())
    lines.append("")
open(dst, "w").write("\n".join(lines))
print("\n".join(lines))
