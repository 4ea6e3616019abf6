# one scan from idle clocks (bench.py side.one_scan_from_idle), with and without sl3d_prewarm; [samples] [sleep seconds]
timeout 1200 python3 bench.py --idle-only --idle-samples ${1:-12} --idle-sleep ${2:-1.0} > $OUT/idle.json 2> $OUT/idle.err
echo "rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/idle.json'))['one_scan_from_idle']
for k,v in d.items():
    if isinstance(v,dict): print(k, 'steady', v['steady'], '\n   from_idle', v['from_idle'], 'penalty', v['idle_penalty'])
"
