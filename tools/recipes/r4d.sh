# on the GPU box (round 4, session d): the camera-side T1 of a radial model from a 4-KB table over r0^2 (libsl3d.so) against the
# 8 B/px HBM table of round 3 (ab/libsl3d_base.so): one view per launch from HBM, 2 / 4 / 16 views, clouds; then the GPU suite
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4d
mkdir -p $T
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200 > $T/ab_c2.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x --durations=8 > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; echo "smoke rc=$?" >> $T/summary.txt
cat $T/summary.txt; tail -15 $T/pytest_gpu.log; tail -4 $T/smoke.log; for f in ab_oneview_cold ab_dense_clouds ab_views2 ab_views4 ab_c2; do echo "== $f"; cat $T/$f.txt; done
