set -u
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "compaction or mask or group" > gpurun_out/r2a/pytest_new.log 2>&1
echo "pytest_new rc=$?" >> gpurun_out/r2a/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r2a/summary.txt
timeout 900 bash tools/ab.sh run > gpurun_out/r2a/ab.log 2>&1
echo "ab rc=$?" >> gpurun_out/r2a/summary.txt
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
echo "bench rc=$?" >> gpurun_out/r2a/summary.txt
cat gpurun_out/r2a/summary.txt; tail -5 gpurun_out/r2a/pytest_new.log; cat gpurun_out/r2a/ab.log
