#!/bin/bash
# tools/run_vpt.sh LIB... : views-per-lane sweep (SL3D_VPT) of the given libraries, then tools/membench
for lib in "$@"; do for v in 1 2 4 8 16; do
r=$(SL3D_VPT=$v SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 1000 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo "$(basename $lib) vpt=$v $r"; done; done
tools/membench 33.1776 | grep -E "flags= (0|1)|mode 0"
