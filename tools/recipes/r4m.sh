# on the GPU box (round 4, session m): rig class 3 (radial projector table in LDS) against the build before it (base: the same
# calibration through rig class 2's global table), reference rig unchanged?, then the parity suites.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4m
mkdir -p $T
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
CLOUDS=1 bash tools/ab.sh alt 3 --rig radial > $T/ab_rig_radial.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 --rig radial > $T/ab_oneview_radial.txt 2>&1
bash tools/ab.sh alt 2 --rig radial --views 4 --steps 4000 --warmup 600 > $T/ab_radial_views4.txt 2>&1
bash tools/ab.sh alt 2 > $T/ab_reference.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 > $T/ab_oneview_reference.txt 2>&1
cat $T/summary.txt; tail -3 $T/pytest_gpu.log; for f in ab_rig_radial ab_oneview_radial ab_radial_views4 ab_reference ab_oneview_reference; do echo "== $f"; cat $T/$f.txt; done
