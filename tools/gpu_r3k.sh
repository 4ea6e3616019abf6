# round 3 (on the GPU box): 16 views per launch, LDS reciprocal table (default) against rcp + Newton (ab/libsl3d_rcp0.so), 6 alternations;
# then rocprofv3 kernel durations of back-to-back ONE-view launches (the launch-to-launch figure of the bench includes the gap between kernels)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds") or {}; print(d["value"], d["roofline"]["frac"], "| clouds", c.get("value"), (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3 4 5 6; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_rcp0.so; do
  echo "rep$rep $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/r3k/rcp_ab16.log
export TMPDIR=/tmp
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_rcp0.so; do
  n=$(basename $lib .so)
  SL3D_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3k/stats_$n -o s -- python3 bench.py --no-cpu-baseline --no-side --no-clouds --views 1 --steps 3000 --warmup 300 > gpurun_out/r3k/one_view_$n.json 2> /dev/null
  echo "$n one view: $(python3 -c "import json; d=json.load(open('gpurun_out/r3k/one_view_$n.json')); print('launch to launch', round(d['ms_per_step']*1e3,2), 'us')")"; head -2 gpurun_out/r3k/stats_$n/*kernel_stats.csv | cut -c1-160
done 2>&1 | tee gpurun_out/r3k/one_view_rocprof.log
