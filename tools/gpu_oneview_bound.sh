# on the GPU box: the data path of a one-view launch at 8 / 4 / 3 waves per SIMD, planes at once or in two dependent halves (tools/membench oneview)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
./tools/membench oneview 2>&1 | tee gpurun_out/oneview_bound.txt
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
echo "fused kernel, 1 view per launch: $(python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views 1 2>/dev/null | python3 -c "$q")" | tee -a gpurun_out/oneview_bound.txt
