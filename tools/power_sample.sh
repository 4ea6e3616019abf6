python3 bench.py --no-cpu-baseline --steps 30000 --warmup 300 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -6; sleep 1; done
wait $BP
python3 -c "import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['ms_per_step'])"
