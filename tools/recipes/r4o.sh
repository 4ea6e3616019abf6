# on the GPU box (round 4, session o): small launches with two tiles per block (small_launch) -- parity first, then against the
# build before it (base) for 1 / 2 / 4 views per launch, the three pipelined rigs, configs[2]; dense 16-view launches must not move.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4o
mkdir -p $T
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 --rig radial > $T/ab_oneview_radial.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 --rig distorted > $T/ab_oneview_distorted.txt 2>&1
bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200 > $T/ab_c2.txt 2>&1
CLOUDS=1 bash tools/ab.sh alt 2 > $T/ab_dense_clouds.txt 2>&1
cat $T/summary.txt; tail -3 $T/pytest_gpu.log; for f in ab_oneview_cold ab_views2 ab_views4 ab_oneview_radial ab_oneview_distorted ab_c2 ab_dense_clouds; do echo "== $f"; cat $T/$f.txt; done
