# on the GPU box (round 4, session l): libsl3d.so = PinnedRows scalar + 32-bit division + unconditional mask prefetch + no false
# dependency at the top of stage 7; defer = the same + stores of view v issued behind the decode of view v + 1; prev = PinnedRows
# scalar only; base = the build before all of it.  Then the parity suites on the shipped build.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4l
mkdir -p $T
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted > $T/ab_rig_distorted.txt 2>&1
bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200 > $T/ab_c2.txt 2>&1
SL3D_LIB=$PWD/ab/libsl3d_defer.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > $T/pytest_defer.log 2>&1; echo "pytest(defer) rc=$?" > $T/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $T/summary.txt
cat $T/summary.txt; tail -2 $T/pytest_defer.log; tail -2 $T/pytest_gpu.log; for f in ab_dense_clouds ab_oneview_cold ab_views2 ab_views4 ab_rig_distorted ab_c2; do echo "== $f"; cat $T/$f.txt; done
