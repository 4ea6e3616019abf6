# on the GPU box: coalesced whole-line stores with the non-temporal hint (shipped since) against the previous store path
# (ab/libsl3d_old.so) and with the hint on the segment stores of the segmented clouds as well (ab/libsl3d_ntseg.so), alternating;
# parity first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ntc
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > gpurun_out/ntc/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/ntc/pytest.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_ntseg.so; do SL3D_LIB=$PWD/$lib timeout 900 python3 tests/fuzz_parity.py 60 11 2>&1 | grep "cases\|FAIL" | head -5; done
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds",{}); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2), "| clouds e2e", c.get("value"), "kernel-only", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_old.so ab/libsl3d_ntseg.so; do
  echo "rep$rep $(basename $lib) views=16: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/ntc/ntc_ab.log
for rep in 1 2; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_old.so; do for v in 1 2 4; do
  echo "rep$rep $(basename $lib) views=$v: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee -a gpurun_out/ntc/ntc_ab.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_old.so; do for rig in distorted general; do
  echo "$(basename $lib) rig=$rig: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1000 --warmup 300 --rig $rig 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee -a gpurun_out/ntc/ntc_ab.log
