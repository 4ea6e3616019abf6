# A/B of the small-launch instantiations: one view per launch from HBM (ONEVIEW), new mask + one view (PERSCAN), 2 and 4 views per launch,
# and configs[2] (4096x3000, 3 views per launch) -- this build against ab/libsl3d_*.so alternating.   smallone [reps]
REPS=${1:-2}
echo "== one view per launch from HBM"; ONEVIEW=1 bash tools/ab.sh alt $REPS 2>&1 | quiet
echo "== new mask + one view"; PERSCAN=1 bash tools/ab.sh alt $REPS 2>&1 | quiet
for v in 2 4; do echo "== $v views per launch"; bash tools/ab.sh alt $REPS --views $v 2>&1 | quiet; done
echo "== configs[2]: 4096x3000, 3 views per launch"; bash tools/ab.sh alt $REPS --width 4096 --height 3000 --fringe-width 4 --views 3 --steps 300 --warmup 50 2>&1 | quiet
