#!/bin/bash
# on the GPU box: package power, shader / memory clocks (rocm-smi, 4 samples per second) while (a) nothing runs, (b) the default
# bench workload runs for ~12 s (16 views per launch, back to back), (c) the traffic-only model of the same launch (tools/membench
# scope: no arithmetic).  What "the kernel runs at the package power cap" means in numbers: DESIGN.md section 10.
cd "$(dirname "$0")/.."
sample() { for i in $(seq 1 $1); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level|fclk clock level" | sed -e 's/^GPU\[0\][ \t]*: //' | tr '\n' ';'; echo; sleep 0.25; done; }
echo "== idle"; sample 4
python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 32000 --warmup 300 > /tmp/power_bench.json 2>/dev/null &
BP=$!
sleep 5
echo "== fused kernel, 16 views per launch, steady"; sample 12
wait $BP
python3 -c "import json; d=json.load(open('/tmp/power_bench.json')); print('bench:', d['value'], 'Mpx/s  frac', d['roofline']['frac'], ' ms/step', d['ms_per_step'])"
( for i in 1 2 3 4 5 6; do ./tools/membench scope > /dev/null 2>&1; done ) &
MP=$!
sleep 3
echo "== traffic-only model (membench scope)"; sample 10
wait $MP
