# The gated MASKIN kernels (views known to be sparsely selected: the reference's 19 % selections): the mask tests, then new mask + one
# view of 1080p with a 18.7 % selection through one launch and through k_mask_prepare + the gated fused kernel, alternating; then a fuzz run.
#   gated [reps] [fuzz cases]
REPS=${1:-3}
timeout 1500 python -m pytest tests/test_gpu_mask_fused.py tests/test_gpu_mask.py -q -m gpu -x --durations=5 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest.log | quiet
for rep in $(seq $REPS); do
  LASSO=1 python3 tools/mask_timing.py 1920 1080 2>>$OUT/stderr.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())['per_scan_device']; t=d['two_kernel_route']
print('rep$rep 19 % selection:', d['scan_us'], 'us per scan,', d['kernel'].split('<')[1], '| two-kernel route', t['scan_us'], 'us,', t['kernel'].split('<')[1], 'mask', t['mask_us'])" | tee -a $OUT/lasso_ab.txt
  python3 tools/mask_timing.py 1920 1080 2>>$OUT/stderr.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())['per_scan_device']; t=d['two_kernel_route']
print('rep$rep dense selection:', d['scan_us'], 'us per scan,', d['kernel'].split('<')[1], '| two-kernel route', t['scan_us'], 'us,', t['kernel'].split('<')[1], 'mask', t['mask_us'])" | tee -a $OUT/lasso_ab.txt
done
timeout 1200 python3 tests/fuzz_parity.py ${2:-150} 631 > $OUT/fuzz_small.log 2>&1; quiet < $OUT/fuzz_small.log | tail -4
