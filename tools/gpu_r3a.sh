# round 3, first kernel session (the camera-polynomial variant it A/B-ed -- SL3D_CAM_POLY -- was measured and removed: profiles/README.md)
# round 3, first kernel session (on the GPU box): the new tests, then A/B of the camera polynomial (SL3D_CAM_POLY=0/1), of the views
# per lane with it (measurement build ab/libsl3d_m.so: SL3D_VPT), of the one-view launch, and of the segmented clouds against the
# look-back (bench side figure) and against 12-byte stores (ab/libsl3d_seg0.so)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -q -m gpu -x -k "not config" > gpurun_out/r3a/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3a/pytest.log
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds") or {}; s=d.get("side") or {}; print(d["value"], d["roofline"]["frac"], "| clouds", c.get("value"), (c.get("kernel_only") or {}).get("value"), "| lookback", (s.get("clouds_lookback_kernel_only") or {}).get("value"), "| 1view", (s.get("one_view_latency") or {}).get("launch_us"))'
for rep in 1 2; do
for poly in 1 0; do
  echo "rep$rep default lib poly=$poly: $(SL3D_CAM_POLY=$poly python3 bench.py --no-cpu-baseline --steps 1500 --warmup 300 2>/dev/null | python3 -c "$q")"
done
[ -f ab/libsl3d_seg0.so ] && echo "rep$rep seg0 lib: $(SL3D_LIB=$PWD/ab/libsl3d_seg0.so python3 bench.py --no-cpu-baseline --no-side --steps 1500 --warmup 300 2>/dev/null | python3 -c "$q")"
done 2>&1 | tee gpurun_out/r3a/ab.log
for poly in 1 0; do for vpt in 1 2 4 8; do
  r=$(SL3D_VPT=$vpt SL3D_CAM_POLY=$poly SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1000 --warmup 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
  echo "poly=$poly vpt=$vpt: $r"
done; done 2>&1 | tee gpurun_out/r3a/vpt.log
for poly in 1 0; do for v in 1 2 4; do
  r=$(SL3D_CAM_POLY=$poly python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")
  echo "poly=$poly views=$v: $r"
done; done 2>&1 | tee gpurun_out/r3a/lat.log
