#!/bin/bash
# Counters of the deflate kernel on 262144 resident 4 KiB blobs (four objects per lane at 65536 lanes), one pass per run.
#   bash tools/collect_deflate_pmc.sh gpurun_out/zpmc
set -e
out=${1:-gpurun_out/zpmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 tools/probe_deflate.py 262144 4096 200 > "$out/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 tools/probe_deflate.py 262144 4096 200 > "$out/write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/tcc" -- python3 tools/probe_deflate.py 262144 4096 200 > "$out/tcc.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -- python3 tools/probe_deflate.py 262144 4096 200 > "$out/sq.log" 2>&1
echo done
