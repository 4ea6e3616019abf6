# round 3 (on the GPU box): A/B of the kernel prologue (mask / camera-table loads requested before the reciprocal-table fill, camera
# entries consumed behind the first view's plane loads) against the previous revision (ab/libsl3d_prev.so): parity first, then
# 1 / 2 / 16 views per launch, alternating three times; then the phase trace of the new prologue
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -q -m gpu -x > gpurun_out/r3e/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3e/pytest.log
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds") or {}; print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2), "| clouds", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_prev.so; do for v in 1 2 16; do
  extra="--no-clouds"; [ $v = 16 ] && extra=""
  echo "rep$rep $(basename $lib) views=$v: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side $extra --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee gpurun_out/r3e/prologue_ab.log
for v in 1 16; do
  SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/phase_trace.py $v > gpurun_out/r3e/phase_trace_$v.txt 2> gpurun_out/r3e/phase_trace_$v.err
  echo "phase trace $v rc=$?"; head -12 gpurun_out/r3e/phase_trace_$v.txt | cut -c1-200
done
