#!/usr/bin/env python3
"""Builds and runs tools/shim_bench.cpp: the reference's six stage calls + save_point_cloud() through the drop-in shim at the
reference's 1600x1200, from BMP files and from memory, with the [col][row] globals produced on the device and (A/B) by host
transposes.  `python tools/shim_timing.py [scans]` prints the JSON; bench.py embeds it as side.shim_scan_ms."""
import importlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(scans=5, timeout=600):
    import numpy as np
    syn = importlib.import_module("3dscan_amd.synth")
    csrc = os.path.join(ROOT, "3dscan_amd", "csrc")
    with tempfile.TemporaryDirectory(prefix="sl3d_shim_") as tmp:
        exe = os.path.join(tmp, "shim_bench")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "shim_bench.cpp"),
                               os.path.join(csrc, "sl3d_shim.cpp"), os.path.join(csrc, "sl3d_shim_globals.cpp"), "-L" + os.path.join(ROOT, "3dscan_amd"),
                               "-lsl3d", "-Wl,-rpath," + os.path.join(ROOT, "3dscan_amd"), "-o", exe])
        cal = syn.cal_tuple(syn.synth_rig(1600, 1200, 1280, 720))
        np.concatenate(cal).astype(np.float64).tofile(os.path.join(tmp, "cal.bin"))
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        r = subprocess.run([exe, os.path.join(tmp, "cal.bin"), data, str(scans)], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError(f"shim_bench rc={r.returncode}: {r.stderr[-400:]}")
        if os.environ.get("SL3D_SHIM_TIMING"):
            sys.stderr.write("".join(l + "\n" for l in r.stderr.splitlines() if "[sl3d shim]" in l))
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 5)))
