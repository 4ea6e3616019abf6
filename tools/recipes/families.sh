# families <tag>: bench.py --families-only under rocprofv3 -> $OUT/<tag>_families_kernel_stats.csv + <tag>_families.json (one row per side.* figure
# of the other kernel families: rig classes 2 / 3 / 0, 9 and 14 Gray planes, 16 views per launch)
TAG=${1:-r06}
STATS_LINES=8 stats ${TAG}_families python3 bench.py --families-only
cp $OUT/${TAG}_families.out $OUT/${TAG}_families.json
