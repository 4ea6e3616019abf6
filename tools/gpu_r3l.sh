# round 3 (on the GPU box): table rigs with the projector-table entries gathered (and parked in LDS) before the next view's plane
# loads, against the previous order (ab/libsl3d_gf0.so): parity first, then the distorted / general / reference rigs, alternating
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > gpurun_out/r3l/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3l/pytest.log
timeout 600 python3 tests/fuzz_parity.py 60 77 > gpurun_out/r3l/fuzz.log 2>&1; grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r3l/fuzz.log | tail -2
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
for rep in 1 2 3; do for rig in distorted reference; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_gf0.so; do
  echo "rep$rep $rig $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1500 --warmup 300 --rig $rig 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee gpurun_out/r3l/gather_first_ab.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_gf0.so; do
  echo "distorted 1 view $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 --rig distorted --views 1 2>/dev/null | python3 -c "$q")"
done 2>&1 | tee -a gpurun_out/r3l/gather_first_ab.log
