# one-view clouds without the scan launch: tests, then A/B against ab/libsl3d_*.so (tools/ab.sh buildrev r4 <rev> beforehand), one box, alternating
timeout 1500 python -m pytest tests -q -m gpu -x -k "cloud or compaction or segment or register" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest.log | quiet
for rep in 1 2 3; do
  for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
    [ -f "$lib" ] || continue
    echo "== rep$rep $(basename $lib)"; SL3D_LIB=$PWD/$lib python3 tools/oneview_clouds.py 2>/dev/null | quiet
  done
done | tee $OUT/oneview_clouds_ab.txt
CLOUDS=1 bash tools/ab.sh alt 2 2>/dev/null | tee $OUT/batch_clouds_ab.txt
stats oneview python3 tools/oneview_clouds.py
