# round 3 (on the GPU box): A/B of SL3D_MASK_EARLY (no s_waitcnt vmcnt(0) between a view's stores and the next view's plane loads)
# against the previous schedule (ab/libsl3d_me0.so): parity first, then 1 / 2 / 16 views per launch, alternating three times
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round2.py -q -m gpu -x > gpurun_out/r3i/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3i/pytest.log
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds") or {}; print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2), "| clouds", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_me0.so; do for v in 1 2 16; do
  extra="--no-clouds"; [ $v = 16 ] && extra=""
  echo "rep$rep $(basename $lib) views=$v: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side $extra --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee gpurun_out/r3i/mask_early_ab.log
for rig in distorted general; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_me0.so; do
  echo "$rig $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1500 --warmup 300 --rig $rig 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee -a gpurun_out/r3i/mask_early_ab.log
