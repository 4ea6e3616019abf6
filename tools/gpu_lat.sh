set -u
cd $GRAFT_REPO_ROOT
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
for v in 1 2 16; do
  r=$(SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")
  echo "$(basename $lib) views=$v: $r"
done; done 2>&1 | tee gpurun_out/lat_sweep.log
