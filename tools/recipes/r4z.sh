# on the GPU box (round 4, session z): views per lane once more on the final kernel (measurement build, SL3D_VPT)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "ms", d["ms_per_step"])'
for rep in 1 2; do
for cfg in "4 1" "4 2" "4 4" "8 2" "8 4" "8 8" "16 2" "16 4" "16 8" "3 1" "3 2" "3 3"; do set -- $cfg
  echo "rep$rep views=$1 vpt=$2: $(SL3D_VPT=$2 SL3D_LIB=$PWD/ab/libsl3d_meas.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps $((24000 / $1)) --warmup $((4000 / $1)) --views $1 2>/dev/null | python3 -c "$q")"
done; done
