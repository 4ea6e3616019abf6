# on the GPU box (round 4, session h): small-launch instantiation for the clouds kernels; full suite; bench line; power trace
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4h
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x --durations=6 > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; echo "smoke rc=$?" >> $T/summary.txt
timeout 900 python3 bench.py > $T/bench.json 2> $T/bench.err; echo "bench rc=$?" >> $T/summary.txt
timeout 300 bash tools/power_trace.sh > $T/power_trace.txt 2>&1
cat $T/summary.txt; tail -4 $T/pytest_gpu.log; python3 -c "
import json; d=json.load(open('$T/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['kernel']); s=d['side']
for k in ('one_view_cold','one_view_cold_clouds','one_view_cache_resident','rig2_distorted_projector','cold_20_steps'): print(k, json.dumps(s.get(k))[:260])
print(json.dumps(d['to_compacted_clouds'])[:300])"; cat $T/power_trace.txt
