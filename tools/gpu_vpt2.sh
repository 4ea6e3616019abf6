# on the GPU box: at most 4 views per lane (shipped) against 8 (ab/libsl3d_vpt8.so), and the 8-part segment scan against the one-block scan
# (ab/libsl3d_scan1.so = the commit before, 8 views per lane), dense and segmented clouds, alternating; parity of the batch paths first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/vpt2
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > gpurun_out/vpt2/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/vpt2/pytest.log
timeout 900 python3 tests/fuzz_parity.py 40 13 2>&1 | grep "cases\|FAIL" | head -5
q='import json,sys; d=json.loads(sys.stdin.read()); c=d.get("to_compacted_clouds",{}); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2), "| clouds e2e", c.get("value"), "kernel-only", (c.get("kernel_only") or {}).get("value"))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_vpt8.so ab/libsl3d_scan1.so; do
  echo "rep$rep $(basename $lib) views=16: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/vpt2/vpt_ab.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_vpt8.so 3dscan_amd/libsl3d.so ab/libsl3d_vpt8.so; do
  echo "$(basename $lib) stripe 135 rows x 128 views: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1000 --warmup 200 --height 135 --views 128 2>/dev/null | python3 -c "$q")"
  echo "$(basename $lib) 8 views: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1500 --warmup 300 --views 8 2>/dev/null | python3 -c "$q")"
  echo "$(basename $lib) distorted: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1000 --warmup 300 --rig distorted 2>/dev/null | python3 -c "$q")"
done 2>&1 | tee -a gpurun_out/vpt2/vpt_ab.log
