# on the GPU box (round 4, session x): rig class 2's projector-table gathers as SGPR base + 32-bit offset (base = the build before)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4x
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x -k "distort or rig or table or projector or small_launches" > $T/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $T/pytest.log
CLOUDS=1 bash tools/ab.sh alt 3 --rig distorted > $T/ab_rig_distorted.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 --rig distorted > $T/ab_oneview_distorted.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted --views 4 --steps 4000 --warmup 600 > $T/ab_distorted_views4.txt 2>&1
bash tools/ab.sh alt 2 --rig general > $T/ab_rig_general.txt 2>&1
for f in ab_rig_distorted ab_oneview_distorted ab_distorted_views4 ab_rig_general; do echo "== $f"; cat $T/$f.txt; done
