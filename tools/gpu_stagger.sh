# on the GPU box: artificial start stagger of the first round of a one-view launch (measurement build, env SL3D_STAGGER)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/stagger
q='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], "us/step", round(d["ms_per_step"]*1e3,2))'
for rep in 1 2; do for s in 0 2 4 6 9 258 260 262 265 0; do
  echo "rep$rep stagger=$s views=1: $(SL3D_STAGGER=$s SL3D_LIB=$PWD/ab/libsl3d_stag.so python3 bench.py --no-cpu-baseline --no-side --no-clouds --steps 3000 --warmup 300 --views 1 2>/dev/null | python3 -c "$q")"
done; done 2>&1 | tee gpurun_out/stagger/stagger_ab.log
for s in 0 4 260; do
SL3D_STAGGER=$s SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/phase_trace.py 1 > gpurun_out/stagger/phase_trace_s$s.txt 2> gpurun_out/stagger/phase_trace_s$s.err; head -12 gpurun_out/stagger/phase_trace_s$s.txt | cut -c1-160
done
