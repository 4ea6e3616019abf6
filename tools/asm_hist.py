#!/usr/bin/env python3
"""Instruction histogram per basic block of one kernel in the gfx950 assembly `make -C 3dscan_amd/csrc asm` writes.

    python tools/asm_hist.py [kernel-substring] [asm-file]

Classes: f64 (v_*_f64 except conversions), cvt, rcp, i/f32 VALU, mov (v_mov / v_accvgpr), cndmask, salu, smem, vmem, lds, wait, branch.
Used to see what the rolled pixel loop of k_fused really issues (DESIGN.md "Where the time goes")."""
import collections
import re
import sys

sub = sys.argv[1] if len(sys.argv) > 1 else "k_fusedILb0ELi10ELb0ELb1ELi1"
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/sl3d_asm/sl3d_fused_dense_rig1-hip-amdgcn-amd-amdhsa-gfx950.s"
s = open(path).read()
m = re.search(r"^(_ZN4sl3d[^\n]*%s[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(sub), s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
print(m.group(1))


def cls(op):
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_div"):
        return "rcp/div"
    if op.startswith("v_") and op.endswith("_f64") or "_f64_" in op:
        return "f64"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "vmov"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane"
    if op.startswith("v_"):
        return "valu32"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


blocks = []
cur = ["entry", collections.Counter(), collections.Counter()]
blocks.append(cur)
for l in m.group(2).split("\n"):
    l = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = [l.split(":")[0], collections.Counter(), collections.Counter()]
        blocks.append(cur)
    elif l and not l.startswith(";") and not l.startswith("."):
        op = l.split()[0]
        cur[1][cls(op)] += 1
        cur[2][op] += 1
verbose = len(sys.argv) > 3
tot = collections.Counter()
for name, c, ops in blocks:
    n = sum(c.values())
    tot.update(c)
    print(f"{name:12s} {n:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
    if verbose and n > 100:
        print("             " + " ".join(f"{k}:{v}" for k, v in ops.most_common(40)))
print("total", sum(tot.values()), dict(tot))
