#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/: one kernel-trace pass of
# the default bench command and four counter passes, each a run of its own
# (counters and traces are never combined; each pass stays within the per-block
# counter limits).  Run on the GPU box from the repository root:
#
#     bash tools/collect_profiles.sh gpurun_out/prof_r01 && \
#     python3 tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01
#
# Every step has its own time limit and the script stops at the first failure.
set -e
out=${1:-gpurun_out/prof}
export TMPDIR=/tmp
mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --no-cpu-baseline --no-end-to-end > "$out/kt.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --no-cpu-baseline --no-end-to-end --steps 13 --warmup 3 > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --no-cpu-baseline --no-end-to-end --steps 13 --warmup 3 > "$out/pmc_write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq" -- python3 bench.py --no-cpu-baseline --no-end-to-end --steps 5 --warmup 3 > "$out/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_lds" -- python3 bench.py --no-cpu-baseline --no-end-to-end --steps 5 --warmup 3 > "$out/pmc_lds.log" 2>&1
grep -h '^{' "$out"/*.log | python3 -c '
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print(d["steps"], "steps:", d["value"], d["unit"], "kernel_ms_avg", d["roofline"]["kernel_ms_avg"])
'
