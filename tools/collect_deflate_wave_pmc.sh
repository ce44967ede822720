#!/bin/bash
# Counters of the deflate kernels (one workgroup per object, tables in LDS), one rocprofv3 pass per run:
# random content on 262144 resident 4 KiB blobs (tools/probe_deflate.py) and text-like content through
# vit_deflate_objects (tools/probe_deflate_text.py 131072 4096).  Fabric traffic, then instruction mix and LDS
# behaviour, then the kernels' durations.
#   bash tools/collect_deflate_wave_pmc.sh gpurun_out/zwpmc
set -e
out=${1:-gpurun_out/zwpmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # name, counters..., then the probe after --
    local name=$1; shift
    local counters=()
    while [ "$1" != "--" ]; do counters+=("$1"); shift; done
    shift
    timeout -s KILL 120 rocprofv3 --pmc "${counters[@]}" --output-format csv -d "$out/$name" -- python3 "$@" > "$out/$name.log" 2>&1
}
for kind in random text; do
    if [ $kind = random ]; then probe="tools/probe_deflate.py 262144 4096 200"; else probe="tools/probe_deflate_text.py 131072 4096"; fi
    run ${kind}_fetch FETCH_SIZE -- $probe
    run ${kind}_write WRITE_SIZE -- $probe
    run ${kind}_tcc TCC_HIT_sum TCC_MISS_sum -- $probe
    run ${kind}_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- $probe
    run ${kind}_sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $probe
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${kind}_kt" -- python3 $probe > "$out/${kind}_kt.log" 2>&1
done
echo done
