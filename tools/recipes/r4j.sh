# on the GPU box (round 4, session j): PinnedRows through scalar loads behind the item's first memory requests (libsl3d.so) against
# six serialized vector loads at the start of every wave (base)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4j
mkdir -p $T
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted > $T/ab_rig_distorted.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
cat $T/summary.txt; tail -2 $T/pytest_gpu.log; for f in ab_oneview_cold ab_views2 ab_views4 ab_dense_clouds ab_rig_distorted; do echo "== $f"; cat $T/$f.txt; done
