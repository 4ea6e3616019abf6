#!/bin/bash
# The one GPU-session runner:   gpurun -- 'bash tools/session.sh <recipe> [args]'
# A recipe is a file tools/recipes/<recipe>.sh, sourced on the GPU box from the root of the repository copy with
#   OUT = gpurun_out/<recipe>  (created), TMPDIR = /tmp, and the helpers below.
# Standing recipes: full (suite + smoke + bench), profile <tag> (the rocprofv3 passes behind profiles/<tag>_*), r8, soak,
# tests <pytest -k expression>, mask (k_mask_prepare: tests + kernel stats), perscan (new mask + one view, both routes), idle, final <tag> ...
set -u
R=${1:?usage: tools/session.sh <recipe> [args]}
shift
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
quiet() { grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl\|amdgpu.ids"; }
# stats <name> <command...>: rocprofv3 --kernel-trace --stats of a command; the per-kernel table goes to $OUT/<name>_kernel_stats.csv
stats() {
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/$name.out 2> $OUT/$name.err
    local f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv && head -${STATS_LINES:-12} $OUT/${name}_kernel_stats.csv
}
[ -f tools/recipes/$R.sh ] || { echo "no recipe tools/recipes/$R.sh"; exit 2; }
source tools/recipes/$R.sh "$@"
