# on the GPU box (round 4, session t): the cold one-view phase trace and the power trace of the final build
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4t
mkdir -p $T
SL3D_LIB=$PWD/ab/trace_libsl3d.so python3 tools/phase_trace.py 1 cold > $T/r04_phase_trace_1_cold_final.txt 2> $T/phase_trace.err
SL3D_LIB=$PWD/ab/trace_libsl3d.so python3 tools/phase_trace.py 2 cold > $T/r04_phase_trace_2_cold_final.txt 2>> $T/phase_trace.err
bash tools/power_trace.sh > $T/r04_power_trace_final.txt 2>&1
./tools/membench oneview_cold > $T/r04_membench_oneview_cold_final.txt 2>&1
cat $T/r04_phase_trace_1_cold_final.txt; grep -A3 "fused kernel" $T/r04_power_trace_final.txt | head -5; grep bench: $T/r04_power_trace_final.txt; cat $T/r04_membench_oneview_cold_final.txt | tail -6
