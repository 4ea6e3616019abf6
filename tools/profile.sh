#!/bin/bash
# Runs on the GPU box:  tools/profile.sh <tag> [extra bench.py arguments, e.g. --width 4096 --height 3000 --fringe-width 4 --views 3]
#  1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average duration)
#  2. separate --pmc passes (no tracing domains beside --kernel-trace) for FETCH_SIZE and WRITE_SIZE on the bench
#     and on tools/membench (known byte counts, same access widths) to calibrate the gfx950 counter units
#  3. a few SQ counters (VALU busy, instruction mix)
# Everything lands in gpurun_out/profile_<tag>/; tools/summarize_profile.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 200 --warmup 50 --precondition-ms 0 --no-cpu-baseline --no-side $EXTRA"   # PMC passes: counters do not depend on the clocks
BENCH_STEADY="python3 bench.py --no-cpu-baseline --no-side $EXTRA"                    # stats pass: the steady-state defaults (300 + 2000 launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH_STEADY > $OUT/stats_bench.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o bench_$n -- $BENCH > /dev/null 2> $OUT/pmc_bench_$n.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o membench_$n -- tools/membench 33.1776 > $OUT/membench_$n.log 2> $OUT/pmc_membench_$n.err
done
for c in "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o bench_$n -- $BENCH > /dev/null 2> $OUT/pmc_bench_$n.err
done
ls $OUT $OUT/stats $OUT/pmc | head -60
