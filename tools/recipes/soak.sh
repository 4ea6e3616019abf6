# long fuzz runs of tests/fuzz_parity.py (random frame / window / pattern-set / rig / mask configurations against the oracle):
# soak [small cases] [large cases]
timeout 1500 python3 tests/fuzz_parity.py ${1:-250} ${3:-505} > $OUT/fuzz_small.log 2>&1; quiet < $OUT/fuzz_small.log | tail -3
FUZZ_MAXW=2100 FUZZ_MAXH=1300 timeout 1500 python3 tests/fuzz_parity.py ${2:-40} ${4:-606} > $OUT/fuzz_large.log 2>&1; quiet < $OUT/fuzz_large.log | tail -3
