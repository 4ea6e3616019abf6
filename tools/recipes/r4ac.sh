# on the GPU box (round 4, sessions ac / ad): early plane requests in the large-launch kernels (ac); their LDS tables filled under those requests (ad)
# keep the gated form) -- base = the build before (commit bc594e1)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4ad
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x > $T/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $T/pytest.log
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
bash tools/ab.sh alt 2 --views 8 --steps 3000 --warmup 400 > $T/ab_views8.txt 2>&1
bash tools/ab.sh alt 2 --views 32 --steps 800 --warmup 150 > $T/ab_views32.txt 2>&1
bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 8 --steps 600 --warmup 100 > $T/ab_c2_8views.txt 2>&1
bash tools/ab.sh alt 2 --rig radial > $T/ab_rig_radial.txt 2>&1
bash tools/ab.sh alt 2 --rig distorted > $T/ab_rig_distorted.txt 2>&1
bash tools/ab.sh alt 2 --ngray 9 --fringe-width 4 > $T/ab_n9.txt 2>&1
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_base.so; do echo "== $lib"; SL3D_LIB=$PWD/$lib python3 tools/sparse_mask.py 2>/dev/null | grep "8 view"; done > $T/sparse.txt
for f in ab_dense_clouds ab_views8 ab_views32 ab_c2_8views ab_rig_radial ab_rig_distorted ab_n9 sparse; do echo "== $f"; cat $T/$f.txt; done
