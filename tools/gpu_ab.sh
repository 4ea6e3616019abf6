# on the GPU box: A/B of every library under ab/ against the default one (steady state, shortened); log -> gpurun_out/ab_<tag>.log
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-x}; shift || true
mkdir -p gpurun_out; rm -f gpurun_out/ab_stderr.log
timeout 1500 bash tools/ab.sh run --steps 600 --warmup 100 "$@" > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; grep "look-back" gpurun_out/ab_stderr.log
if [ -f ab/libsl3d_h3p1v16.so ]; then echo "vpt16:"; SL3D_VPT=16 SL3D_LIB=$PWD/ab/libsl3d_h3p1v16.so python3 bench.py --no-cpu-baseline --no-side --steps 600 --warmup 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('to_compacted_clouds',{}); print(d['value'], '| clouds', c.get('value'), (c.get('kernel_only') or {}).get('value'))"; fi
