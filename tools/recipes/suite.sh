# suite: the whole -m gpu suite + smoke on this build -> $OUT/final_gpu_suite.txt (what profiles/<tag>_final_gpu_suite.txt is a copy of)
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 > $OUT/final_gpu_suite.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/final_gpu_suite.txt | quiet
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/final_gpu_suite.txt 2>&1; echo "smoke rc=$?"
