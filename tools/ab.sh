#!/bin/bash
# Kernel A/B experiments.
#   here (no GPU):  tools/ab.sh build NAME "-DSL3D_PAIR_UNROLL=2" [NAME2 "flags2" ...]   -> ab/libsl3d_NAME.so (+ VGPR/scratch report)
#                   tools/ab.sh buildrev NAME REV ["flags"]   -> ab/libsl3d_NAME.so built from git revision REV
#   on the GPU box: tools/ab.sh run [bench args]        -> one line per variant: value frac ms  (steady-state defaults)
# ab/ is git-ignored but travels with gpurun.  The default library (3dscan_amd/libsl3d.so) is always measured as "base".
set -u
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
if [ "$1" = build ]; then
  shift; mkdir -p ab
  while [ $# -ge 2 ]; do
    n=$1; f=$2; shift 2
    hipcc $FLAGS $f -shared -o ab/libsl3d_$n.so 3dscan_amd/csrc/sl3d_kernels.hip 3dscan_amd/csrc/sl3d_capi.cpp 3dscan_amd/csrc/sl3d_group.cpp -Iinclude -ldl \
      -Rpass-analysis=kernel-resource-usage 2> ab/$n.res || { echo "build $n failed"; tail -5 ab/$n.res; continue; }
    echo "$n [$f]:"; python3 tools/kres.py ab/$n.res 10 | grep -E "KEEP=0 .*RIG=1" 
  done
elif [ "$1" = buildrev ]; then
  # tools/ab.sh buildrev NAME REV [flags] : the library as of git revision REV (baseline for the working tree)
  n=$2; rev=$3; f=${4:-}; mkdir -p ab /tmp/ab_$n/3dscan_amd/csrc /tmp/ab_$n/include
  for x in sl3d_kernels.hip sl3d_capi.cpp sl3d_group.cpp sl3d_ctx.h sl3d_internal.h sl3d_atan_coeffs.h; do git show $rev:3dscan_amd/csrc/$x > /tmp/ab_$n/3dscan_amd/csrc/$x 2>/dev/null || rm -f /tmp/ab_$n/3dscan_amd/csrc/$x; done
  git show $rev:include/sl3d.h > /tmp/ab_$n/include/sl3d.h
  hipcc $FLAGS $f -shared -o ab/libsl3d_$n.so /tmp/ab_$n/3dscan_amd/csrc/*.hip /tmp/ab_$n/3dscan_amd/csrc/*.cpp -ldl && echo "built ab/libsl3d_$n.so from $rev"
elif [ "$1" = run ]; then
  shift
  for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
    [ -f "$lib" ] || continue
    r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side "$@" 2>>gpurun_out/ab_stderr.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('to_compacted_clouds',{}); print(d['value'], d['roofline']['frac'], d['ms_per_step'], '| clouds', c.get('value'), (c.get('kernel_only') or {}).get('value'))")
    echo "$(basename $lib) $r"
  done
fi
