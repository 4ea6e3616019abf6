# round 3 (on the GPU box): the clouds tests on the new scan kernel, the bench line, and rocprofv3 kernel stats of the same command
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_shim.py -q -m gpu -x > gpurun_out/r3g/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3g/pytest.log
python3 bench.py > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err; echo "bench rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3g/stats -o stats -- python3 bench.py --no-cpu-baseline --no-side > gpurun_out/r3g/stats_bench.json 2> gpurun_out/r3g/stats.err
head -6 gpurun_out/r3g/stats/*kernel_stats.csv | cut -c1-200
