# on the GPU box: the Level-1 drop-in path per scan (tools/shim_bench.cpp), phases of save_point_cloud() on stderr
set -u
cd $GRAFT_REPO_ROOT
T=gpurun_out/r4g
mkdir -p $T
SL3D_SHIM_TIMING=1 timeout 600 python3 tools/shim_timing.py 5 > $T/shim_scan_ms.json 2> $T/shim_scan_ms.err
grep "sl3d shim" $T/shim_scan_ms.err | tail -4; cat $T/shim_scan_ms.json
timeout 900 python -m pytest tests/test_gpu_shim.py -q -m gpu -x 2>&1 | tail -3
