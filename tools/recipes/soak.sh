set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
timeout 1500 python3 tests/fuzz_parity.py 250 303 > gpurun_out/soak/fuzz_small.log 2>&1; grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/soak/fuzz_small.log | tail -3
FUZZ_MAXW=2100 FUZZ_MAXH=1300 timeout 1500 python3 tests/fuzz_parity.py 40 404 > gpurun_out/soak/fuzz_large.log 2>&1; grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/soak/fuzz_large.log | tail -3
# launch time against the number of Gray planes (bytes per pixel 20 + 4N), same box, alternating: what part of a launch scales with the bytes
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "ms", d["ms_per_step"])'
for rep in 1 2; do for cfg in "6 32" "8 8" "10 2" "12 2" "9 4" "11 2"; do set -- $cfg
  echo "rep$rep N=$1 fw=$2: $(python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1500 --warmup 300 --ngray $1 --fringe-width $2 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/soak/time_vs_ngray.log
