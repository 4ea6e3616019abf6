# on the GPU box (round 4, session y): k_seg_scan with one memory round trip (base = the build before)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4y
mkdir -p $T
timeout 1500 python -m pytest tests -q -m gpu -x -k "cloud or segment or register or compact or shim or padded or small_launches or config2 or 4gib" > $T/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $T/pytest.log
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_base.so; do
  SL3D_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $T/ovc_$(basename $lib .so) -o stats -- python3 tools/oneview_clouds.py 2>/dev/null | grep "one view" | sed "s|^|$(basename $lib): |"
  grep -h "k_seg_scan\|2, false>" $T/ovc_$(basename $lib .so)/*kernel_stats.csv | cut -d, -f1-4 | sed "s|^|$(basename $lib): |"
done > $T/oneview_clouds.txt 2>&1
CLOUDS=1 bash tools/ab.sh alt 3 > $T/ab_dense_clouds.txt 2>&1
CLOUDS=1 bash tools/ab.sh alt 2 --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 1200 --warmup 200 > $T/ab_c2_clouds.txt 2>&1
cat $T/oneview_clouds.txt; for f in ab_dense_clouds ab_c2_clouds; do echo "== $f"; cat $T/$f.txt; done
