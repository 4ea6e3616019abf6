# k_mask_prepare (H0 / S3b / S3d): the tests that reach it, then its time per view under rocprofv3 at 1080p and 12 Mpx
timeout 1500 python -m pytest tests -q -m gpu -x -k "mask or quad or gray_planes or sparse" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest.log
stats mask1080 python3 tools/mask_timing.py 1920 1080
stats mask12m python3 tools/mask_timing.py 4096 3000
cat $OUT/mask1080.out $OUT/mask12m.out | quiet
