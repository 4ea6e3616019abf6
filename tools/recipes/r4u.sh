# on the GPU box (round 4, session u): the straight-line kernels take any pair of axes (shorter axis padded in front) -- parity, the
# launch times of unequal axes (tools/nvnh.py), and the equal-axes workloads against the build before (base): must not move.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4u
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
python3 tools/nvnh.py 16 > $T/nvnh_16.txt 2>/dev/null
python3 tools/nvnh.py 1 > $T/nvnh_1.txt 2>/dev/null
SL3D_LIB=$PWD/ab/libsl3d_base.so python3 tools/nvnh.py 16 > $T/nvnh_16_base.txt 2>/dev/null
SL3D_LIB=$PWD/ab/libsl3d_base.so python3 tools/nvnh.py 1 > $T/nvnh_1_base.txt 2>/dev/null
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200 > $T/ab_c2.txt 2>&1
bash tools/ab.sh alt 2 --rig radial > $T/ab_rig_radial.txt 2>&1
cat $T/summary.txt; tail -3 $T/pytest_gpu.log; for f in nvnh_16 nvnh_16_base nvnh_1 nvnh_1_base ab_dense_clouds ab_oneview_cold ab_views4 ab_c2 ab_rig_radial; do echo "== $f"; cat $T/$f.txt; done
