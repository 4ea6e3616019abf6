# on the GPU box (round 4, session ab): no wait for store acknowledgements inside a view -- stage 5's LDS pair offset without the
# v_mad_u64_u32 false dependency, the next view's valid bits taken right behind the decode (base = commit 4544834)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4ab
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x > $T/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $T/pytest.log
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 2 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
bash tools/ab.sh alt 2 --views 8 --steps 3000 --warmup 400 > $T/ab_views8.txt 2>&1
bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200 > $T/ab_c2.txt 2>&1
bash tools/ab.sh alt 2 --rig radial > $T/ab_rig_radial.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted > $T/ab_rig_distorted.txt 2>&1
for f in ab_dense_clouds ab_oneview_cold ab_views4 ab_views8 ab_c2 ab_rig_radial ab_rig_distorted; do echo "== $f"; cat $T/$f.txt; done
