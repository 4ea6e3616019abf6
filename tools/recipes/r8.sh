# on the GPU box: the N = 8 branch of bench.py as the driver starts it (torch.distributed.run, 8 ranks), on ONE GPU over gloo -- the
# row arithmetic of 8 stripes, the rank report, the assembly legs, and stdout = exactly one JSON line; then the same check at N = 1
mkdir -p gpurun_out/r8
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 10 --warmup 3 --backend gloo --devices 0,0,0,0,0,0,0,0 --views 4 --no-cpu-baseline --no-side --check > gpurun_out/r8/bench8.json 2> gpurun_out/r8/bench8.err
echo "rc=$? stdout lines: $(wc -l < gpurun_out/r8/bench8.json)"; python3 -c "
import json; d=json.load(open('gpurun_out/r8/bench8.json')); print(d['value'], d['n_gpus'], d['config']['rows_per_gpu'], d['config']['sharding']); print(json.dumps(d.get('with_assembly'))[:2500]); print(len(d.get('ranks',[])), 'ranks reported')"
timeout 600 python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-side > gpurun_out/r8/bench1.json 2> gpurun_out/r8/bench1.err
echo "N=1 rc=$? stdout lines: $(wc -l < gpurun_out/r8/bench1.json)"; python3 -c "
import json; d=json.load(open('gpurun_out/r8/bench1.json')); print(d['value'], d['roofline']['frac'])"
