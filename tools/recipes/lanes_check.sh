export TMPDIR=/tmp; mkdir -p gpurun_out/lanes2
timeout 900 python -m pytest tests/test_gpu_lanes.py -q -m gpu -x 2>&1 | tail -2
for mode in lanes serial; do
  if [ $mode = serial ]; then export SL3D_NO_LAUNCH_LANES=1; else unset SL3D_NO_LAUNCH_LANES; fi
  python3 bench.py --no-cpu-baseline --idle-samples 0 2>gpurun_out/lanes2/$mode.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['side']
print('$mode', 'headline', d['value'], '| same view repeated', s['one_view_cache_resident']['launch_us'], '| one_view_cold', s['one_view_cold']['launch_us'], '| per_scan', s['per_scan_device']['scan_us'], s['per_scan_device_clouds']['scan_us'], s['per_scan_device_19pct_selection']['scan_us'], '| config2', s['config2_12mp']['ms_per_launch'], s['config2_12mp']['frac'], '| clouds', d['to_compacted_clouds']['value'])"
done
