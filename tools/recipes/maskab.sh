# k_mask_prepare variants (ab/libsl3d_*.so against the default build), per-view kernel time under rocprofv3 at 1080p and 12 Mpx
for lib in 3dscan_amd/libsl3d.so ab/libsl3d_*.so; do
  n=$(basename $lib .so)
  for size in "1920 1080" "4096 3000"; do
    t=${size% *}
    SL3D_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_${n}_$t -o s -- python3 tools/mask_timing.py $size > $OUT/${n}_$t.out 2> /dev/null
    f=$(find $OUT/p_${n}_$t -name "*kernel_stats.csv" | head -1)
    echo "$n $size: $(grep k_mask_prepare $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('calls', r[1], 'avg_us', round(float(r[3])/1e3,2), 'min_us', round(float(r[5])/1e3,2))") | $(python3 -c "
import json,sys
for l in open('$OUT/${n}_$t.out'):
    if l.startswith('{'):
        d=json.loads(l); p=d['per_scan_device']; print('scan_us', p['scan_us'], 'mask_us', p['mask_us'], 'pinned_ready', d['set_mask_us']['pinned']['until_ready'])")"
  done
done | tee $OUT/mask_variants.txt
