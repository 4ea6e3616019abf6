set -u
mkdir -p gpurun_out/r2b
cd $GRAFT_REPO_ROOT
timeout 900 bash tools/ab.sh run --steps 600 --warmup 100 > gpurun_out/r2b/ab.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "compaction or mask or group" > gpurun_out/r2b/pytest_new.log 2>&1
echo "pytest_new rc=$?" >> gpurun_out/r2b/summary.txt
cat gpurun_out/r2b/summary.txt; tail -15 gpurun_out/r2b/pytest_new.log; cat gpurun_out/r2b/ab.log
