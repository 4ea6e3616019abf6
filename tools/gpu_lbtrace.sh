set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== look-back trace of one sl3d_run_clouds launch (16 x 1920x1080, final structure: SL3D_SLACK=2, first window requested behind the plane loads)"
SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/lb_trace.py 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
echo "== look-back counters (-DSL3D_CX=64 build; the counters themselves slow the kernel down by ~12 %)"
SL3D_LIB=$PWD/ab/libsl3d_st.so python3 bench.py --no-cpu-baseline --no-side --steps 300 --warmup 50 2>&1 >/dev/null | grep "look-back"
echo "== ablations, same box (bench.py --steps 600: value | to_compacted_clouds end to end, kernel only)"
bash tools/ab.sh run --steps 600 --warmup 100 2>/dev/null | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl"
} > gpurun_out/r02_lookback_trace.txt 2>&1
cat gpurun_out/r02_lookback_trace.txt
