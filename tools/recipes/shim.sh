# the drop-in shim: its tests (stage by stage and deferred), then the per-scan host wall time of every route (tools/shim_bench.cpp)
timeout 1500 python -m pytest tests/test_gpu_shim.py -q -m gpu -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest.log
SL3D_SHIM_TIMING=1 timeout 900 python3 tools/shim_timing.py ${1:-7} > $OUT/shim_scan_ms.json 2> $OUT/shim_timing.err
echo "shim_timing rc=$?"; cat $OUT/shim_scan_ms.json; grep "sl3d shim" $OUT/shim_timing.err | tail -4
