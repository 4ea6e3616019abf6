# on the GPU box: alternate default / ab libraries 3 times on the default bench (steady state) to see through run-to-run noise
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
  r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 1500 --warmup 300 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
  echo "rep$rep $(basename $lib): $r"
done; done 2>&1 | tee gpurun_out/ab3.log
