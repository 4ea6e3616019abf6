# on the GPU box (round 4, session c): frame layout in the data-path model, early projector-table gathers (RIG 2), phase trace of
# the one-view launch from HBM
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r4c
mkdir -p $T
timeout 300 ./tools/membench layout > $T/membench_layout.txt 2>&1
bash tools/ab.sh alt 3 --rig distorted > $T/ab_rig_distorted.txt 2>&1
bash tools/ab.sh alt 2 --rig general > $T/ab_rig_general.txt 2>&1
SL3D_LIB=$PWD/ab/trace_libsl3d.so timeout 300 python3 tools/phase_trace.py 1 cold > $T/phase_trace_1_cold.txt 2>&1
SL3D_LIB=$PWD/ab/trace_libsl3d.so timeout 300 python3 tools/phase_trace.py 1 > $T/phase_trace_1_cached.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu -x -k "distort or rig or table or projector" > $T/pytest_rig.log 2>&1; echo "pytest rc=$?" > $T/summary.txt
cat $T/summary.txt $T/membench_layout.txt $T/ab_rig_distorted.txt $T/ab_rig_general.txt; head -50 $T/phase_trace_1_cold.txt; tail -3 $T/pytest_rig.log
