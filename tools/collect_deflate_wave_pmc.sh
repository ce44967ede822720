#!/bin/bash
# Counters of the wave-per-object deflate kernels on 65536 text-like 4 KiB blobs (tools/probe_deflate_phases.py), one
# pass per run: instruction mix and LDS behaviour, then fabric traffic.
#   bash tools/collect_deflate_wave_pmc.sh gpurun_out/zwpmc [random|text]
set -e
out=${1:-gpurun_out/zwpmc}
kind=${2:-text}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$out/sq1" -- python3 tools/probe_deflate_phases.py 65536 4096 $kind > "$out/sq1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/sq2" -- python3 tools/probe_deflate_phases.py 65536 4096 $kind > "$out/sq2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 tools/probe_deflate_phases.py 65536 4096 $kind > "$out/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 tools/probe_deflate_phases.py 65536 4096 $kind > "$out/write.log" 2>&1
echo done
