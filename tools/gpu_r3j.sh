# round 3 (on the GPU box): the LDS reciprocal table against rcp + Newton (ab/libsl3d_rcp0.so) per launch size, alternating
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds") or {}; print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2), "| clouds", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_rcp0.so; do for v in 1 2 4 16; do
  extra="--no-clouds"; [ $v = 16 ] && extra=""
  echo "rep$rep $(basename $lib) views=$v: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side $extra --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee gpurun_out/r3j/rcp_ab.log
