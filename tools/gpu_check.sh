# on the GPU box: rig sweep of the default library, small batches, then the -m gpu suite
set -u
cd $GRAFT_REPO_ROOT
bash tools/gpu_rigs.sh
for v in 1 2 4; do
  python3 bench.py --no-cpu-baseline --no-side --steps 2000 --warmup 300 --views $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('views', d['config']['views_per_gpu_per_step'], d['value'], d['roofline']['frac'], d['ms_per_step'])"
done
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -8
