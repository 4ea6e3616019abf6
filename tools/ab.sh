#!/bin/bash
# Kernel A/B experiments.
#   here (no GPU):  tools/ab.sh build NAME "-DSL3D_MEASURE -DSL3D_ABLATE=4" [NAME2 "flags2" ...]   -> ab/libsl3d_NAME.so (+ VGPR/scratch report)
#                   tools/ab.sh buildrev NAME REV ["flags"]   -> ab/libsl3d_NAME.so built from git revision REV
#   on the GPU box: tools/ab.sh run [bench args]        -> one line per variant: value frac ms  (steady-state defaults)
#                   tools/ab.sh alt REPS [bench args]   -> the same, REPS times over, alternating the libraries (boxes drift by several
#                                                          per cent within minutes: only alternating runs compare); CLOUDS=1 adds the
#                                                          compacting launch, ONEVIEW=1 measures side.one_view_cold (one view per launch
#                                                          from HBM) instead of the batch, PERSCAN=1 side.per_scan_device (new mask + one view)
# ab/ is git-ignored but travels with gpurun.  The default library (3dscan_amd/libsl3d.so) is always measured as "base".
set -u
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
if [ "$1" = build ]; then
  shift; mkdir -p ab
  while [ $# -ge 2 ]; do
    n=$1; f=$2; shift 2
    # every translation unit of the library, compiled in parallel into its own object directory, with this variant's flags
    mkdir -p /tmp/ab_obj_$n; : > ab/$n.res; pids=""
    for src in 3dscan_amd/csrc/sl3d_fused_*.hip 3dscan_amd/csrc/sl3d_kernels.hip 3dscan_amd/csrc/sl3d_capi_*.cpp 3dscan_amd/csrc/sl3d_group.cpp; do
      ( hipcc $FLAGS $f -Iinclude -Rpass-analysis=kernel-resource-usage -c -x hip $src -o /tmp/ab_obj_$n/$(basename $src).o 2> /tmp/ab_obj_$n/$(basename $src).res ) & pids="$pids $!"
    done
    ok=1; for p in $pids; do wait $p || ok=0; done
    cat /tmp/ab_obj_$n/*.res > ab/$n.res
    [ $ok = 1 ] && hipcc $FLAGS -shared -o ab/libsl3d_$n.so /tmp/ab_obj_$n/*.o -ldl || { echo "build $n failed"; grep -m5 error ab/$n.res; continue; }
    echo "$n [$f]:"; python3 tools/kres.py ab/$n.res 10 | grep -E "KEEP=0 .*RIG=1" 
  done
elif [ "$1" = buildrev ]; then
  # tools/ab.sh buildrev NAME REV [flags] : the library as of git revision REV (baseline for the working tree)
  n=$2; rev=$3; f=${4:-}; mkdir -p ab /tmp/ab_$n/3dscan_amd/csrc /tmp/ab_$n/include
  for x in $(git ls-tree --name-only $rev 3dscan_amd/csrc/ | xargs -n1 basename | grep -E '\.(hip|cpp|h)$' | grep -v shim); do git show $rev:3dscan_amd/csrc/$x > /tmp/ab_$n/3dscan_amd/csrc/$x; done
  git show $rev:include/sl3d.h > /tmp/ab_$n/include/sl3d.h
  hipcc $FLAGS $f -shared -o ab/libsl3d_$n.so /tmp/ab_$n/3dscan_amd/csrc/*.hip /tmp/ab_$n/3dscan_amd/csrc/*.cpp -ldl && echo "built ab/libsl3d_$n.so from $rev"
elif [ "$1" = run ]; then
  shift
  for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
    [ -f "$lib" ] || continue
    r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side "$@" 2>>gpurun_out/ab_stderr.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('to_compacted_clouds',{}); print(d['value'], d['roofline']['frac'], d['ms_per_step'], '| clouds', c.get('value'), (c.get('kernel_only') or {}).get('value'))")
    echo "$(basename $lib) $r"
  done
elif [ "$1" = alt ]; then
  shift; reps=$1; shift
  mkdir -p gpurun_out
  for rep in $(seq 1 $reps); do
    for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
      [ -f "$lib" ] || continue
      if [ "${PERSCAN:-0}" = 1 ]; then
        r=$(SL3D_LIB=$PWD/$lib python3 tools/mask_timing.py 1920 1080 2>>gpurun_out/ab_stderr.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['per_scan_device']; t=d.get('two_kernel_route') or {}; print(d['scan_us'], 'us per scan,', d['kernel'].split('<')[1], '| two-kernel route', t.get('scan_us'), 'us, mask', t.get('mask_us'))")
      elif [ "${ONEVIEW:-0}" = 1 ]; then
        r=$(SL3D_LIB=$PWD/$lib python3 bench.py --one-view-cold-only --steps 4000 "$@" 2>>gpurun_out/ab_stderr.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['one_view_cold']; print(d['launch_us'], 'us  frac', d['frac'], ' on moved bytes', d['frac_on_moved_bytes'])")
      else
        extra="--no-clouds"; [ "${CLOUDS:-0}" = 1 ] && extra=""
        r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side $extra --steps 1500 --warmup 300 "$@" 2>>gpurun_out/ab_stderr.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d.get('to_compacted_clouds') or {}; k=(c.get('kernel_only') or {}).get('value')
print(d['value'], d['roofline']['frac'], d['ms_per_step'], '| clouds', c.get('value'), k)")
      fi
      echo "rep$rep $(basename $lib): $r"
    done
  done
fi
