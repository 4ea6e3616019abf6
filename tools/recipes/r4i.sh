# on the GPU box (round 4, session i): the rocprofv3 passes behind profiles/r04_* and profiles/c2_r04_* on the round's final build,
# the cold one-view launch under rocprofv3, the bench line with the driver's arguments
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4i
mkdir -p $T
bash tools/profile.sh r04 > $T/profile_r04.log 2>&1
python3 tools/summarize_profile.py r04 > $T/profile_r04_summary.log 2>&1
cp gpurun_out/profile_r04/stats_bench.json $T/r04_bench_under_rocprof.json 2>/dev/null
bash tools/profile.sh c2_r04 --width 4096 --height 3000 --fringe-width 4 --views 3 > $T/profile_c2_r04.log 2>&1
python3 tools/summarize_profile.py c2_r04 > $T/profile_c2_r04_summary.log 2>&1
cp gpurun_out/profile_c2_r04/stats_bench.json $T/c2_r04_bench.json 2>/dev/null
cp profiles/r04_kernel_stats.csv profiles/r04_pmc.json profiles/r04_traffic.json profiles/r04_traffic_clouds.json profiles/c2_r04_kernel_stats.csv profiles/c2_r04_pmc.json profiles/c2_r04_traffic.json profiles/c2_r04_traffic_clouds.json $T/ 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $T/oneview_cold_stats -o stats -- python3 bench.py --one-view-cold-only --steps 4000 > $T/r04_oneview_cold_under_rocprof.json 2> $T/oneview_cold_rocprof.err
cp $T/oneview_cold_stats/*kernel_stats.csv $T/r04_oneview_cold_kernel_stats.csv 2>/dev/null
timeout 900 python3 bench.py > $T/r04_bench.json 2> $T/bench.err
timeout 600 python3 bench.py --steps 20 --warmup 5 > $T/r04_bench_driver_args.json 2> $T/bench_driver.err
head -4 $T/r04_kernel_stats.csv; cat $T/r04_traffic.json | head -8; head -3 $T/c2_r04_kernel_stats.csv; head -3 $T/r04_oneview_cold_kernel_stats.csv
python3 -c "
import json
for f in ('r04_bench','r04_bench_driver_args','r04_bench_under_rocprof','c2_r04_bench'):
    d=json.load(open('$T/%s.json'%f)); print(f, d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic_source'))"
