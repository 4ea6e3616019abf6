# round 3 (on the GPU box): small launches with the first view's planes requested before the mask is known (EARLY) against the
# previous prologue (ab/libsl3d_ep0.so): parity first, then 1 / 2 / 4 views per launch (reference and distorted rigs), alternating;
# then the phase trace of the new one-view launch
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > gpurun_out/r3s/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3s/pytest.log
timeout 600 python3 tests/fuzz_parity.py 60 5 > gpurun_out/r3s/fuzz.log 2>&1; grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r3s/fuzz.log | tail -2
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
for rep in 1 2 3; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_ep0.so; do for v in 1 2 4; do
  echo "rep$rep $(basename $lib) views=$v: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views $v 2>/dev/null | python3 -c "$q")"
done; done; done 2>&1 | tee gpurun_out/r3s/early_planes_ab.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_ep0.so; do
  echo "distorted 1 view $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views 1 --rig distorted 2>/dev/null | python3 -c "$q")"
done 2>&1 | tee -a gpurun_out/r3s/early_planes_ab.log
SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/phase_trace.py 1 > gpurun_out/r3s/phase_trace_1.txt 2> gpurun_out/r3s/phase_trace_1.err; head -12 gpurun_out/r3s/phase_trace_1.txt | cut -c1-160
