# launch lanes: the tests, then one view per launch from HBM and the per-scan loop (dense and 19 % selections) with the lanes and
# (SL3D_NO_LAUNCH_LANES=1) without, alternating.   lanes [reps]
REPS=${1:-3}
timeout 1500 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_mask_fused.py tests/test_gpu_mask.py -q -m gpu -x --durations=5 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest.log | quiet
one() { python3 bench.py --one-view-cold-only --steps 4000 2>>$OUT/stderr.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['one_view_cold']; print(d['launch_us'], 'us  frac', d['frac'])"; }
scan() { python3 tools/mask_timing.py 1920 1080 2>>$OUT/stderr.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())['per_scan_device']; t=d['two_kernel_route']
print(d['scan_us'], 'us per scan,', d['kernel'].split('<')[1], '| two-kernel route', t['scan_us'])"; }
for rep in $(seq $REPS); do
  echo "rep$rep one view per launch, lanes:  $(one)" | tee -a $OUT/lanes_ab.txt
  echo "rep$rep one view per launch, serial: $(SL3D_NO_LAUNCH_LANES=1 one)" | tee -a $OUT/lanes_ab.txt
  echo "rep$rep per scan, lanes:  $(scan)" | tee -a $OUT/lanes_ab.txt
  echo "rep$rep per scan, serial: $(SL3D_NO_LAUNCH_LANES=1 scan)" | tee -a $OUT/lanes_ab.txt
  echo "rep$rep per scan 19 %, lanes:  $(LASSO=1 scan)" | tee -a $OUT/lanes_ab.txt
  echo "rep$rep per scan 19 %, serial: $(SL3D_NO_LAUNCH_LANES=1 LASSO=1 scan)" | tee -a $OUT/lanes_ab.txt
done
