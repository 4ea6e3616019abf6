# on the GPU box (round 4, session k): 32-bit tile division (libsl3d.so) and, on top, the next views' mask dwords requested
# unconditionally so that no copy of a loaded value sits behind the plane loads (maskuncond: stage 7 starts without waiting for the
# next view's planes), against the build before (base)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4k
mkdir -p $T
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted > $T/ab_rig_distorted.txt 2>&1
for f in ab_dense_clouds ab_oneview_cold ab_views2 ab_views4 ab_rig_distorted; do echo "== $f"; cat $T/$f.txt; done
