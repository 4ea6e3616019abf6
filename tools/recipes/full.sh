# on the GPU box: the whole -m gpu suite, smoke, the default bench line; logs under gpurun_out/<tag>/
set -u

TAG=full
mkdir -p gpurun_out/$TAG
timeout 3000 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/$TAG/pytest_gpu.log 2>&1
echo "pytest rc=$?" > gpurun_out/$TAG/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/$TAG/summary.txt
timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
echo "bench rc=$?" >> gpurun_out/$TAG/summary.txt
cat gpurun_out/$TAG/summary.txt; tail -30 gpurun_out/$TAG/pytest_gpu.log
