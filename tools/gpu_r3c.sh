# round 3 (on the GPU box): what a BANDED one-view launch could reach, measured without writing it: K views of 1920 x (1080/K)
# with K views per lane (measurement build, SL3D_VPT=K) move the same bytes as one 1080p view through 2025/K blocks that pipeline
# their K steps, against the shipped one-view launch (2025 blocks, 2 rounds on 1024 slots, nothing to pipeline)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
for rep in 1 2; do
  echo "rep$rep 1 view 1080 rows (shipped): $(SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views 1 2>/dev/null | python3 -c "$q")"
  for k in 2 3 4 6 8; do
    h=$((1080 / k))
    echo "rep$rep $k bands of $h rows, $k per lane: $(SL3D_VPT=$k SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views $k --height $h 2>/dev/null | python3 -c "$q")"
  done
  echo "rep$rep 2 views 1080 rows, 2 per lane: $(SL3D_VPT=2 SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views 2 2>/dev/null | python3 -c "$q")"
  echo "rep$rep 2 views 1080 rows, 1 per lane: $(SL3D_VPT=1 SL3D_LIB=$PWD/ab/libsl3d_m.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views 2 2>/dev/null | python3 -c "$q")"
done 2>&1 | tee gpurun_out/r3c/bands.log
