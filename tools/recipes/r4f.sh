# on the GPU box (round 4, session f): block size of the small-launch instantiation (256 = libsl3d.so, 128, 64) with both pixel pairs
# unrolled, against the shipped rolled loops (base); the shim's shared-text formatter + mapped parallel file population
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4f
mkdir -p $T
ONEVIEW=1 bash tools/ab.sh alt 3 > $T/ab_oneview_cold.txt 2>&1
bash tools/ab.sh alt 2 --views 2 --steps 6000 --warmup 1000 > $T/ab_views2.txt 2>&1
bash tools/ab.sh alt 2 --views 4 --steps 4000 --warmup 600 > $T/ab_views4.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_shim.py tests/test_gpu_round3.py -q -m gpu -x > $T/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
timeout 900 python3 tools/shim_timing.py 5 > $T/shim_scan_ms.json 2> $T/shim_timing.err; echo "shim timing rc=$?" >> $T/summary.txt
cat $T/summary.txt; tail -3 $T/pytest_gpu.log; for f in ab_oneview_cold ab_views2 ab_views4; do echo "== $f"; cat $T/$f.txt; done; cat $T/shim_scan_ms.json
