# tests [-k expression ...]: part of the -m gpu suite
timeout 2400 python -m pytest tests -q -m gpu -x --durations=10 "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log
