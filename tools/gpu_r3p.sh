# round 3 (on the GPU box): dense xyz stores coalesced across the wave (ab/libsl3d_cst.so) against three 16-byte stores per lane, 5 alternations
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
SL3D_LIB=$PWD/ab/libsl3d_cst.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -m gpu -x -k "not shim" > gpurun_out/r3p/pytest_cst.log 2>&1
echo "pytest (coalesced-store build) rc=$?"; tail -3 gpurun_out/r3p/pytest_cst.log
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
for rep in 1 2 3 4 5; do for lib in 3dscan_amd/libsl3d.so ab/libsl3d_cst.so; do
  echo "rep$rep $(basename $lib): $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/r3p/coalesced_stores_ab.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_cst.so; do for v in 1 4; do
  echo "$(basename $lib) views=$v: $(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee -a gpurun_out/r3p/coalesced_stores_ab.log
