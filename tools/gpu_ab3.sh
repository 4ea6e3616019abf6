# on the GPU box: alternate default / ab libraries 3 times on the default bench (steady state) to see through run-to-run noise
#   CLOUDS=1: report the compacting launch too (dense value | clouds end to end, kernel only, kernel-only / dense)
set -u
cd $GRAFT_REPO_ROOT
EXTRA="--no-clouds"; [ "${CLOUDS:-0}" = 1 ] && EXTRA=""
for rep in 1 2 3; do
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
  r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side $EXTRA --steps 1500 --warmup 300 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d.get('to_compacted_clouds') or {}; k=(c.get('kernel_only') or {}).get('value')
print(d['value'], d['roofline']['frac'], '|', c.get('value'), k, round(k/d['value'],4) if k else '')")
  echo "rep$rep $(basename $lib): $r"
done; done 2>&1 | tee gpurun_out/ab3.log
