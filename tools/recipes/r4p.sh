# on the GPU box (round 4, session p): is it the second tile or the new unit loop?  meas = this build with -DSL3D_MEASURE, run with
# SL3D_TPB=1 (one tile per block through the new loop); libsl3d.so = two tiles per block; base = the build before.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4p
mkdir -p $T
SL3D_TPB=1 ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_tpb1.txt 2>&1
SL3D_TPB=1 bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2_tpb1.txt 2>&1
SL3D_TPB=1 bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4_tpb1.txt 2>&1
for f in ab_oneview_tpb1 ab_views2_tpb1 ab_views4_tpb1; do echo "== $f"; cat $T/$f.txt; done
