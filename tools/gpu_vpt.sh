# on the GPU box: views-per-lane x camera-table sweep of the dense kernel (measurement build ab/libsl3d_m.so)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rig in reference distorted general; do
for tab in 0 1; do for vpt in 1 2 4 8; do
  r=$(SL3D_VPT=$vpt SL3D_CAMTAB=$tab SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --steps 600 --warmup 100 --rig $rig 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
  echo "rig=$rig camtab=$tab vpt=$vpt: $r"
done; done; done 2>&1 | tee gpurun_out/vpt_sweep.log
for v in 1 2 4; do
  r=$(SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")
  echo "views=$v default: $r"
  r=$(SL3D_CAMTAB=0 SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")
  echo "views=$v camtab=0: $r"
done 2>&1 | tee -a gpurun_out/vpt_sweep.log
