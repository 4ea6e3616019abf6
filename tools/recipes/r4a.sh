# on the GPU box (round 4, session a): where the round starts from.
#   1 the default bench line (now with side.one_view_cold / cold_20_steps)   2 the driver's arguments
#   3 the data path of a one-view launch from HBM (tools/membench oneview_cold)
#   4 one view per launch from HBM: shipped build, measurement build with / without the camera table
#   5 rocprofv3 --kernel-trace --stats of the cold one-view command
#   6 configs[2] (4096x3000, 3 views) under rocprofv3 + PMC: profiles/c2_r04_*; views / views-per-lane sweep of that shape
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4a
mkdir -p $T
q='import json,sys; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(d.get("value"), r.get("frac"), d.get("ms_per_step"), json.dumps(d.get("one_view_cold") or ""))'
timeout 900 python3 bench.py > $T/bench.json 2> $T/bench.err; echo "bench rc=$?" > $T/summary.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > $T/bench_driver_args.json 2> $T/bench_driver_args.err; echo "bench driver args rc=$?" >> $T/summary.txt
timeout 300 ./tools/membench oneview_cold > $T/membench_oneview_cold.txt 2>&1
{
for rep in 1 2; do
  echo "shipped:            $(python3 bench.py --one-view-cold-only --steps 4000 2>/dev/null | python3 -c "$q")"
  echo "measure, table:     $(SL3D_LIB=$PWD/ab/libsl3d_measure.so python3 bench.py --one-view-cold-only --steps 4000 2>/dev/null | python3 -c "$q")"
  echo "measure, no table:  $(SL3D_CAMTAB=0 SL3D_LIB=$PWD/ab/libsl3d_measure.so python3 bench.py --one-view-cold-only --steps 4000 2>/dev/null | python3 -c "$q")"
  echo "shipped, 2 resident views (cache): $(python3 bench.py --one-view-cold-only --cold-views 2 --steps 4000 2>/dev/null | python3 -c "$q")"
  echo "shipped, 4 resident views:         $(python3 bench.py --one-view-cold-only --cold-views 4 --steps 4000 2>/dev/null | python3 -c "$q")"
  echo "shipped, 16 resident views:        $(python3 bench.py --one-view-cold-only --cold-views 16 --steps 4000 2>/dev/null | python3 -c "$q")"
done
} > $T/oneview_cold_ab.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $T/oneview_cold_stats -o stats -- python3 bench.py --one-view-cold-only --steps 4000 > $T/oneview_cold_under_rocprof.json 2> $T/oneview_cold_rocprof.err
cp $T/oneview_cold_stats/*kernel_stats.csv $T/r04_oneview_cold_kernel_stats.csv 2>/dev/null
# configs[2]
bash tools/profile.sh c2_r04 --width 4096 --height 3000 --fringe-width 4 --views 3 > $T/profile_c2_r04.log 2>&1
python3 tools/summarize_profile.py c2_r04 > $T/profile_c2_r04_summary.log 2>&1
cp profiles/c2_r04_* $T/ 2>/dev/null; cp gpurun_out/profile_c2_r04/stats_bench.json $T/c2_r04_bench.json 2>/dev/null
{
run() { python3 bench.py --no-cpu-baseline --no-side --no-clouds --width 4096 --height 3000 --fringe-width 4 --steps 1200 --warmup 200 "$@" 2>/dev/null | python3 -c "$q"; }
for rep in 1 2; do
for v in 3 4 6 8; do echo "c2 views=$v shipped: $(run --views $v)"; done
for vpt in 1 2 3 4; do echo "c2 views=3 vpt=$vpt (measure): $(SL3D_VPT=$vpt SL3D_LIB=$PWD/ab/libsl3d_measure.so run --views 3)"; done
for vpt in 1 2 4; do echo "c2 views=4 vpt=$vpt (measure): $(SL3D_VPT=$vpt SL3D_LIB=$PWD/ab/libsl3d_measure.so run --views 4)"; done
done
} > $T/c2_sweep.txt 2>&1
cat $T/summary.txt; cat $T/oneview_cold_ab.txt; cat $T/membench_oneview_cold.txt | tail -20; cat $T/c2_sweep.txt; head -5 $T/r04_oneview_cold_kernel_stats.csv
