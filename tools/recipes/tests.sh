# tests [paths / -k "expression" ...]: part of the -m gpu suite (default: all of tests/)
if [ $# -eq 0 ]; then set -- tests; else case "$1" in -*) set -- tests "$@";; esac; fi
timeout 2400 python -m pytest -q -m gpu -x --durations=10 "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log | quiet
