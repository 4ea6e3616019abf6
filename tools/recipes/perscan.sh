# The per-scan path (new mask + ONE view): the fused-mask tests, then side.per_scan_device through both routes (one MASKIN launch /
# k_mask_prepare + the fused kernel), this build against ab/libsl3d_*.so alternating, and the per-kernel averages under rocprofv3.
#   perscan [reps] [notests]
REPS=${1:-3}
if [ "${2:-}" != notests ]; then
  timeout 1500 python -m pytest tests/test_gpu_mask_fused.py -q -m gpu -x --durations=5 > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; tail -12 $OUT/pytest.log | quiet
fi
PERSCAN=1 bash tools/ab.sh alt $REPS 2>&1 | quiet | tee $OUT/perscan_ab.txt
stats perscan python3 tools/mask_timing.py 1920 1080
