# final <tag>: the round's evidence on the build that ships -- the rocprofv3 passes behind profiles/<tag>_* and profiles/c2_<tag>_*,
# the per-scan kernels, the one-view launches (dense and clouds) under rocprofv3, the shim's per-scan wall times, the bench line
# (defaults and the driver's arguments), the whole -m gpu suite + smoke.  Everything lands in gpurun_out/final/ under the names it
# is committed with in profiles/.
TAG=${1:-r05}
T=$OUT
bash tools/profile.sh $TAG > $T/profile_$TAG.log 2>&1
python3 tools/summarize_profile.py $TAG > $T/profile_${TAG}_summary.log 2>&1
cp gpurun_out/profile_$TAG/stats_bench.json $T/${TAG}_bench_under_rocprof.json 2>/dev/null
bash tools/profile.sh c2_$TAG --width 4096 --height 3000 --fringe-width 4 --views 3 > $T/profile_c2_$TAG.log 2>&1
python3 tools/summarize_profile.py c2_$TAG > $T/profile_c2_${TAG}_summary.log 2>&1
cp gpurun_out/profile_c2_$TAG/stats_bench.json $T/c2_${TAG}_bench.json 2>/dev/null
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/${TAG}_traffic.json profiles/${TAG}_traffic_clouds.json profiles/c2_${TAG}_kernel_stats.csv profiles/c2_${TAG}_pmc.json profiles/c2_${TAG}_traffic.json profiles/c2_${TAG}_traffic_clouds.json $T/ 2>/dev/null
# every kernel on the per-scan path: new device-resident mask + one-view launch (dense, then clouds), cold views; 1080p and 12 Mpx.
# The rocprofv3 passes of the one-view paths run with SL3D_NO_LAUNCH_LANES=1: a kernel's duration is then that of a LONE launch (with the
# lanes two launches share the GPU and each one's own duration says little); the overlapped series is what bench.py's HIP events time,
# and one pass with the lanes on (<tag>_oneview_cold_lanes) shows the overlap in the trace.
export SL3D_NO_LAUNCH_LANES=1
STATS_LINES=6 stats ${TAG}_per_scan python3 tools/mask_timing.py 1920 1080
STATS_LINES=6 stats c2_${TAG}_per_scan python3 tools/mask_timing.py 4096 3000
cp $T/${TAG}_per_scan.out $T/${TAG}_per_scan.json; cp $T/c2_${TAG}_per_scan.out $T/c2_${TAG}_per_scan.json
# ... and with the reference's own kind of selection (a 19 % lasso, a new one every scan): the gated kernels
export LASSO=1
STATS_LINES=6 stats ${TAG}_per_scan_19pct python3 tools/mask_timing.py 1920 1080
unset LASSO
cp $T/${TAG}_per_scan_19pct.out $T/${TAG}_per_scan_19pct.json
STATS_LINES=6 stats ${TAG}_oneview python3 tools/oneview_clouds.py
cp $T/${TAG}_oneview.out $T/${TAG}_oneview_dense_clouds_host.txt
STATS_LINES=4 stats ${TAG}_oneview_cold python3 bench.py --one-view-cold-only --steps 4000
cp $T/${TAG}_oneview_cold.out $T/${TAG}_oneview_cold_under_rocprof.json
unset SL3D_NO_LAUNCH_LANES
STATS_LINES=4 stats ${TAG}_oneview_cold_lanes python3 bench.py --one-view-cold-only --steps 4000
cp $T/${TAG}_oneview_cold_lanes.out $T/${TAG}_oneview_cold_lanes_under_rocprof.json
# the 16-view launches of the other kernel families (rig classes 2 / 3 / 0, 9 and 14 Gray planes): one rocprofv3 row per side.* figure
STATS_LINES=8 stats ${TAG}_families python3 bench.py --families-only
cp $T/${TAG}_families.out $T/${TAG}_families.json
STATS_LINES=4 stats ${TAG}_rig3 python3 bench.py --rig radial --no-cpu-baseline --no-side
cp $T/${TAG}_rig3.out $T/${TAG}_rig3_bench_under_rocprof.json
SL3D_SHIM_TIMING=1 timeout 900 python3 tools/shim_timing.py 7 > $T/${TAG}_shim_scan_ms.json 2> $T/shim_timing.err
timeout 900 python3 bench.py > $T/${TAG}_bench.json 2> $T/bench.err
timeout 600 python3 bench.py --steps 20 --warmup 5 > $T/${TAG}_bench_driver_args.json 2> $T/bench_driver.err
python3 -c "
import json
for f in ('${TAG}_bench','${TAG}_bench_driver_args','${TAG}_bench_under_rocprof','c2_${TAG}_bench'):
    try:
        d=json.load(open('$T/%s.json'%f)); print(f, d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic'))
    except Exception as e: print(f, 'ERR', e)"
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 > $T/${TAG}_final_gpu_suite.txt 2>&1; echo "pytest rc=$?"; tail -3 $T/${TAG}_final_gpu_suite.txt | quiet
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $T/${TAG}_final_gpu_suite.txt 2>&1; echo "smoke rc=$?"
cat $T/${TAG}_traffic.json | head -12
