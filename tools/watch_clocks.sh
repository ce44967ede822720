#!/bin/bash
# Sample the GPU's shader clock and power while the benchmark kernel runs back to back
# (kernel variant $1: lds or lane; $2 = steps).  Reading only; changes no setting.
variant=${1:-lds}
steps=${2:-5000}
python3 bench.py --no-cpu-baseline --kernel "$variant" --steps "$steps" > gpurun_out/clk_bench_$variant.log 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4; do
    timeout 10 rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | tr -s ' ' | head -8
    echo "--"
    sleep 0.7
done
wait $pid
grep '^{' gpurun_out/clk_bench_$variant.log | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); print(d["config"]["kernel"], d["value"], d["roofline"]["kernel_ms_avg"])'
