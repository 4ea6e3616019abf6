# tests [paths / -k expression ...]: part of the -m gpu suite (default: all of tests/)
ARGS="$*"; [ -z "$ARGS" ] && ARGS=tests
case "$ARGS" in -*) ARGS="tests $ARGS";; esac
timeout 2400 python -m pytest -q -m gpu -x --durations=10 $ARGS > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log | quiet
