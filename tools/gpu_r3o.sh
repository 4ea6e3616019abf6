# round 3 (on the GPU box): the final build against the revision before the camera-frame solve took a skew term (ab/libsl3d_preskew.so):
# reference rig, 16 views, alternating 4 times -- the two extra FMAs per pixel must not show
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds") or {}; print(d["value"], d["roofline"]["frac"], "| clouds", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3 4; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_preskew.so; do
  echo "rep$rep $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/r3o/final_vs_preskew.log
