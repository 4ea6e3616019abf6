# round 3 (on the GPU box): final checks of the round's build -- the whole suite, smoke, the default bench line, the N > 1 branch on
# one GPU (gloo) with its assembly legs printed, then the rocprofv3 passes behind profiles/r03_*
set -u
cd $GRAFT_REPO_ROOT
bash tools/gpu_full.sh r3f
python3 bench.py --gpus 2 --backend gloo --devices 0,0 --views 2 --steps 20 --warmup 5 --no-cpu-baseline --no-side --check > gpurun_out/r3f/bench_2ranks_gloo.json 2> gpurun_out/r3f/bench_2ranks_gloo.err
echo "2-rank gloo bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3f/bench_2ranks_gloo.json')); print(json.dumps(d.get('with_assembly'))[:1500]); print(d.get('ranks'))"
tail -4 gpurun_out/r3f/bench_2ranks_gloo.err
export TMPDIR=/tmp
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1
python3 tools/summarize_profile.py r03 > gpurun_out/profile_r03_summary.log 2>&1
tail -5 gpurun_out/profile_r03_summary.log; cp profiles/r03_* gpurun_out/ 2>/dev/null; head -8 gpurun_out/r03_kernel_stats.csv
cp gpurun_out/profile_r03/stats_bench.json gpurun_out/r03_bench_under_rocprof.json 2>/dev/null
