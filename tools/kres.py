#!/usr/bin/env python3
"""Per-instantiation register / LDS / occupancy table of k_fused from `hipcc -Rpass-analysis=kernel-resource-usage` output.
    make -C 3dscan_amd/csrc resources TU=sl3d_fused_dense_rig1 2> res.txt; tools/kres.py res.txt [N ...]"""
import re
import sys

txt = open(sys.argv[1]).read()
want = set(sys.argv[2:]) or {"10"}
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    m = re.search(r"k_fusedILb(\d)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)", name)
    if not m:
        continue
    keep, nmax, fgen, exact, rig, comp, rcpt, early = m.groups()
    if fgen == "1" or nmax not in want:
        continue

    def g(k):
        return re.search(k + r": (\d+)", b).group(1)
    print(f"KEEP={keep} N={nmax} EXACT={exact} RIG={rig} CMODE={comp} RCPT={rcpt} EARLY={early}: VGPR {g('VGPRs')} SGPR {g('SGPRs')} "
          f"scratch {g('ScratchSize .bytes/lane.')} occ {g('Occupancy .waves/SIMD.')} LDS {g('LDS Size .bytes/block.')}")
