# on the GPU box (round 4, session q): two tiles per block with the stores of a unit issued right behind its stage 7 (nodefer)
# instead of behind the next unit's decode
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4q
mkdir -p $T
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
for f in ab_oneview ab_views2; do echo "== $f"; cat $T/$f.txt; done
