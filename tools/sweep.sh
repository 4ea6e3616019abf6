#!/bin/bash
# Workload sweep of the committed kernel (steady-state defaults scaled so that every line runs ~1 s of GPU time).
run() { python3 bench.py --no-cpu-baseline --no-side --no-clouds "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %8.1f Gpx/s  frac %.3f  %.4f ms' % (' '.join(sys.argv[1:]), d['value']/1e3, d['roofline']['frac'], d['ms_per_step']))" "$@"; }
run --views 1 --steps 20000 --warmup 3000
run --views 4 --steps 8000 --warmup 1000
run --views 16
run --views 32 --steps 1000 --warmup 150
run --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200
run --width 8192 --height 768 --ngray 12 --views 8 --steps 1200 --warmup 200
run --width 1600 --height 1200 --views 16
run --ngray 8 --fringe-width 8 --views 16
run --ngray 9 --fringe-width 4 --views 16
run --rig distorted --views 16
run --rig general --views 16
