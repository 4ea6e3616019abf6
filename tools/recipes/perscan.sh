# The per-scan path (new mask + ONE view): the fused-mask tests, then side.per_scan_device through both routes (one MASKIN launch /
# k_mask_prepare + the fused kernel) and the one-view launch from HBM, this build against ab/libsl3d_*.so alternating.
#   perscan [reps]
REPS=${1:-3}
timeout 1500 python -m pytest tests/test_gpu_mask_fused.py -q -m gpu -x --durations=5 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest.log | quiet
for r in $(seq 1 $REPS); do
  python3 tools/mask_timing.py 1920 1080 2>>$OUT/stderr.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['per_scan_device']; t=d.get('two_kernel_route',{})
print('per_scan rep$r: one launch', d['scan_us'], 'us frac', d['frac'], d['kernel'], '| two kernels', t.get('scan_us'), 'us mask', t.get('mask_us'))"
done
ONEVIEW=1 bash tools/ab.sh alt $REPS 2>&1 | quiet
stats perscan python3 tools/mask_timing.py 1920 1080
