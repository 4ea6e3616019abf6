for lib in 3dscan_amd/libsl3d.so ab/libsl3d_a1.so; do for v in 1 2 4 8 16; do
r=$(SL3D_VPT=$v SL3D_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 1000 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo "$(basename $lib) vpt=$v $r"; done; done
tools/membench 33.1776 | tail -22
