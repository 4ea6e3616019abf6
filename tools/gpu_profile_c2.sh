set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile.sh c2_r02 --width 4096 --height 3000 --fringe-width 4 --views 3 > gpurun_out/profile_c2_r02.log 2>&1
python3 tools/summarize_profile.py c2_r02 > gpurun_out/profile_c2_r02_summary.log 2>&1
cp profiles/c2_r02_* gpurun_out/ 2>/dev/null; cp gpurun_out/profile_c2_r02/stats_bench.json gpurun_out/c2_r02_bench.json
head -4 gpurun_out/c2_r02_kernel_stats.csv; cat gpurun_out/c2_r02_traffic.json | head -8; python3 -c "
import json; d=json.load(open('gpurun_out/c2_r02_bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('to_compacted_clouds'))"
