# round 3 (on the GPU box): views per lane 2 / 4 / 8 of the final dense kernel at 16 views per launch, alternating 4 times (measurement build)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"])'
for rep in 1 2 3 4; do for vpt in 8 4 2; do
  echo "rep$rep vpt=$vpt: $(SL3D_VPT=$vpt SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/r3n/vpt_b.log
rocm-smi --showclocks 2>/dev/null | head -20
