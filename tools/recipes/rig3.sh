# rig class 3 (radial projector table) after the fp64 node coordinate: its tests + the fake-RCCL / distinct-sides group tests, then
# the batch rate against ab/libsl3d_*.so, alternating
timeout 1500 python -m pytest tests -q -m gpu -x -k "radial or reference_distorted or fake_rccl or distinct_sides or randomised" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log | quiet
bash tools/ab.sh alt 3 --rig radial 2>/dev/null | tee $OUT/rig3_ab.txt
