# round 3 (on the GPU box): the whole suite on the current build, then the phase traces of the dense kernel (ab/libsl3d_trace.so)
set -u
cd $GRAFT_REPO_ROOT
bash tools/gpu_full.sh r3d
mkdir -p gpurun_out/r3d
for v in 1 2 16; do
  SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/phase_trace.py $v > gpurun_out/r3d/phase_trace_$v.txt 2> gpurun_out/r3d/phase_trace_$v.err
  echo "phase trace $v views rc=$?"; head -14 gpurun_out/r3d/phase_trace_$v.txt
done
