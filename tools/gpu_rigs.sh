# on the GPU box: every library (default + ab/) on the three rig classes, 16 views, steady state (shortened)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
  [ -f "$lib" ] || continue
  for rig in reference distorted general; do
    r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --steps 600 --warmup 100 --rig $rig "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('to_compacted_clouds',{}); print(d['value'], d['roofline']['frac'], '| clouds', c.get('value'), (c.get('kernel_only') or {}).get('value'))")
    echo "$(basename $lib) rig=$rig: $r"
  done
done 2>&1 | tee gpurun_out/rigs.log
