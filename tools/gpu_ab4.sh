# on the GPU box: A/B of everything under ab/ against the shipped build, 16 views per launch, dense + segmented clouds, three alternations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab4
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds",{}); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2), "| clouds kernel-only", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
  echo "rep$rep $(basename $lib) views=16: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --steps 1500 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/ab4/ab.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
  echo "$(basename $lib) views=1: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 --views 1 2>/dev/null | python3 -c "$q")"
done 2>&1 | tee -a gpurun_out/ab4/ab.log
