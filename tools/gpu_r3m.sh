# round 3 (on the GPU box): the camera-frame solve generalised to a skewed (upper-triangular, affine) camera matrix -- the "general"
# rig of the bench now takes the pipelined table kernel (RIG 2) instead of the un-pipelined general one (RIG 0; ab/libsl3d_noskew.so)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > gpurun_out/r3m/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3m/pytest.log
timeout 900 python3 tests/fuzz_parity.py 90 101 > gpurun_out/r3m/fuzz.log 2>&1; grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r3m/fuzz.log | tail -2
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
for rep in 1 2 3; do for rig in general distorted reference; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_noskew.so; do
  echo "rep$rep $rig $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1500 --warmup 300 --rig $rig 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee gpurun_out/r3m/skew_ab.log
