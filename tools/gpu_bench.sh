set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_driver_args.json 2>> gpurun_out/r02_bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/r02_bench.json'))
for k in ('value','ms_per_step','roofline','to_compacted_clouds','side','set_mask_us','host_buffers_one_view','host_buffers_pipelined'): print(k, json.dumps(d.get(k))[:900])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'], d['cpu_baseline']['gpu_matches_oracle'])
d=json.load(open('gpurun_out/r02_bench_driver_args.json')); print('driver args:', d['value'], d['roofline']['frac'])
"
