# on the GPU box (round 4, session b): the restructured kernel translation units (sl3d_fused.h + sl3d_fused_*.hip, wavefront fences
# at the cross-lane LDS hand-offs) against round 3's final build and against the same build without the fences.
#   1 the whole -m gpu suite + smoke    2 alternating A/B: 16 views dense + clouds, one view per launch from HBM
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4b
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; echo "smoke rc=$?" >> $T/summary.txt
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted > $T/ab_rig_distorted.txt 2>&1
cat $T/summary.txt; tail -25 $T/pytest_gpu.log; cat $T/ab_dense_clouds.txt $T/ab_oneview_cold.txt $T/ab_rig_distorted.txt
