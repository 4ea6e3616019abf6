#!/usr/bin/env python3
"""tools/isa_diff.py A.s B.s [substring] -- compares the gfx950 ISA of every kernel two builds have in common.

A.s / B.s are `hipcc -save-temps` device assembly files (`make -C 3dscan_amd/csrc asm` writes /tmp/sl3d_*-gfx950.s; several
files per side may be given separated by commas).  For every kernel symbol present on both sides the instruction stream is
compared after normalising what a refactoring may legitimately change: comments, directives, local label numbers.
Prints one line per kernel that differs (first differing instruction + the instruction-histogram delta) and a summary.
With `--kernarg` the immediate offsets of scalar loads from the kernel-argument segment (s[0:1] / s[4:5] ...) are masked too
(a change of the KParams layout moves them and nothing else)."""
import collections
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def kernels(paths, mask_kernarg):
    out = {}
    for path in paths.split(","):
        cur, body = None, []
        for line in open(path, errors="replace"):
            line = line.split(";", 1)[0].rstrip()
            if not line.strip():
                continue
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur, body = m.group(1), []
                continue
            if cur is None:
                continue
            s = line.strip()
            if s.startswith(".Lfunc_end"):
                out[cur] = body
                cur = None
                continue
            if s.startswith(".") and not s.startswith(".LBB"):
                continue
            body.append(s)
    norm = {}
    for k, body in out.items():
        labels = {}
        res = []
        for s in body:
            def lab(m):
                return labels.setdefault(m.group(0), f".L{len(labels)}")
            s = re.sub(r"\.LBB\d+_\d+", lab, s)
            if mask_kernarg:
                s = re.sub(r"(s_load_\w+ \S+ s\[\d+:\d+\], )0x[0-9a-f]+", r"\1OFF", s)
            res.append(s)
        norm[k] = res
    return norm


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mask = "--kernarg" in sys.argv
    A, B = kernels(args[0], mask), kernels(args[1], mask)
    sub = args[2] if len(args) > 2 else ""
    common = sorted(set(A) & set(B))
    names = demangle(common)
    same = diff = 0
    for k in common:
        if sub and sub not in names[k]:
            continue
        if A[k] == B[k]:
            same += 1
            continue
        diff += 1
        ha = collections.Counter(s.split()[0] for s in A[k] if not s.endswith(":"))
        hb = collections.Counter(s.split()[0] for s in B[k] if not s.endswith(":"))
        delta = {op: hb[op] - ha[op] for op in set(ha) | set(hb) if hb[op] != ha[op]}
        first = next((i for i, (x, y) in enumerate(zip(A[k], B[k])) if x != y), min(len(A[k]), len(B[k])))
        print(f"DIFF {names[k]}: {len(A[k])} -> {len(B[k])} lines, first difference at {first}: "
              f"{A[k][first] if first < len(A[k]) else '<end>'!r} vs {B[k][first] if first < len(B[k]) else '<end>'!r}; histogram delta {delta or 'none (order only)'}")
    onlyA = [names.get(k, k) for k in sorted(set(A) - set(B)) if not sub or sub in k]
    onlyB = [k for k in sorted(set(B) - set(A)) if not sub or sub in k]
    print(f"{same} kernels identical, {diff} differ; {len(set(A) - set(B))} only in A, {len(set(B) - set(A))} only in B")


if __name__ == "__main__":
    main()
