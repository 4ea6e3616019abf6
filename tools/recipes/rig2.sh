# rig class 2 (projector with tangential terms: per-pixel table gathers) against ab/libsl3d_*.so, alternating: 16 views, 4 views, one view cold
bash tools/ab.sh alt 3 --rig distorted 2>/dev/null | tee $OUT/rig2_ab16.txt
bash tools/ab.sh alt 2 --rig distorted --views 4 2>/dev/null | tee $OUT/rig2_ab4.txt
ONEVIEW=1 bash tools/ab.sh alt 2 --rig distorted 2>/dev/null | tee $OUT/rig2_ab1.txt
